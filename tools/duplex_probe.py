"""Why did host-to-device and device-to-host copies not overlap?  Measures, with page-locked and with pageable host
buffers: H2D alone, D2H alone, and both at once on two streams (hipMemcpyAsync through torch's copy_ non_blocking),
GB/s per direction.  If the duplex rates equal the solo rates the link and the two copy engines are full duplex and an
earlier serialisation came from pageable staging or a shared stream; if they halve, the two directions share a resource."""
import sys, time
import torch

dev = torch.device("cuda", 0)
MB = 256
n = MB * (1 << 20) // 4


def bench(pinned: bool):
    h_src = torch.empty(n, dtype=torch.int32, pin_memory=pinned)
    h_dst = torch.empty(n, dtype=torch.int32, pin_memory=pinned)
    h_src.fill_(1)
    d_a = torch.empty(n, dtype=torch.int32, device=dev)
    d_b = torch.ones(n, dtype=torch.int32, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def run(h2d: bool, d2h: bool, reps=6):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if h2d:
                with torch.cuda.stream(s1):
                    d_a.copy_(h_src, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2):
                    h_dst.copy_(d_b, non_blocking=True)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    t_h2d, t_d2h, t_both = run(True, False), run(False, True), run(True, True)
    gb = MB / 1024
    kind = "page-locked" if pinned else "pageable"
    print(f"{kind:11s} {MB} MiB each way: H2D alone {gb / t_h2d:6.1f} GB/s, D2H alone {gb / t_d2h:6.1f} GB/s, both at once "
          f"{t_both * 1e3:6.2f} ms = {gb / t_both:6.1f} GB/s per direction ({2 * gb / t_both:6.1f} GB/s total; "
          f"sum of the solo times {1e3 * (t_h2d + t_d2h):6.2f} ms, longer solo time {1e3 * max(t_h2d, t_d2h):6.2f} ms)", flush=True)


for pinned in (True, False):
    bench(pinned)
# small transfers (the host API moves 64 x 0.6 MB in, 64 x 1.2 MB out per call): rate vs size, page-locked
for kb in (64, 1024, 16384):
    m = kb * 256
    h = torch.empty(m, dtype=torch.int32, pin_memory=True)
    d = torch.empty(m, dtype=torch.int32, device=dev)
    for name, f in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"  {name} {kb:6d} KiB page-locked: {kb / 1048576 / dt:6.1f} GB/s ({dt * 1e6:7.1f} us per copy)", flush=True)
