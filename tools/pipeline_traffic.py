"""Whole-pipeline HBM traffic per frame from the two PMC tables of tools/pipeline_traffic.sh.
usage: pipeline_traffic.py <workload> FETCH_SIZE.txt WRITE_SIZE.txt bench.json   -> the two tables + one JSON line
       pipeline_traffic.py --merge out.json file.txt [file.txt ...]              -> profiles/pipeline_traffic.json"""
import json, re, sys
if sys.argv[1] == "--merge":
    res = {}
    for f in sys.argv[3:]:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        res[j["workload"]] = j
    open(sys.argv[2], "w").write(json.dumps(res, indent=1) + "\n")
    sys.exit(0)
wl, ff, wf, bj = sys.argv[1:5]
def table(path):
    rows = {}
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"(.*\S)\s+(\d+)\s+(\d+)\s*$", line)
        if m:
            rows[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return rows
F, Wt = table(ff), table(wf)
b = json.loads(open(bj).read().strip().splitlines()[-1])
frames = b["config"]["batch_per_gpu"] * (b["steps"] + b["warmup"])
# the plane-fit kernel also runs the bench's stand-alone series (--roofline-iters 1: three warm-up launches + one timed launch of a whole group)
pf_frames = frames + 4 * b["config"]["batch_per_gpu"] * b["config"]["batches_per_launch"]
print(open(ff).read().rstrip()); print(open(wf).read().rstrip())
own = lambda k: k.startswith("dpx::") and "verify_fastdiv" not in k
scale = lambda k: frames / pf_frames if "planefit" in k else 1.0
fetch = sum(n * v * scale(k) for k, (n, v) in F.items() if own(k))
write = sum(n * v * scale(k) for k, (n, v) in Wt.items() if own(k))
m = re.search(r"\((\d+)x(\d+)\)", b["metric"])
N = int(m.group(1)) * int(m.group(2))
per_frame = (2 * fetch + write) * 1024 / frames
per_kernel = {k: round((2 * F.get(k, (0, 0))[0] * F.get(k, (0, 0))[1] + Wt.get(k, (0, 0))[0] * Wt.get(k, (0, 0))[1]) * scale(k) * 1024 / frames) for k in sorted(set(F) | set(Wt)) if own(k)}
print(json.dumps({"workload": wl, "frames_counted": frames, "hbm_bytes_per_frame": int(round(per_frame)), "algorithmic_bytes_per_frame": N * 6,
                  "ratio": round(per_frame / (N * 6), 3), "bytes_per_frame_by_kernel": per_kernel,
                  "correction": "2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 counts 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section), summed over every "
                                "launch of the library's kernels in warm-up + timed steps, divided by the frames those steps processed",
                  "command": f"tools/pipeline_traffic.sh <tag> {wl}"}))
