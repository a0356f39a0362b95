"""Diagnostic: where the waves of the pipeline's kernels sit in steady state (dpx_debug_trace).
Prints time-averaged occupancy per SIMD / CU by kernel kind: resident waves, VGPR share, LDS share, and how often a CU
could have taken another workgroup of each kind.   usage: python tools/wave_trace.py [steps] [DPX_SKIP letters]"""
import os, sys, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import subprocess
from pathlib import Path as _P
_root = _P(__file__).resolve().parents[1]
# the product build carries no trace code: build (in a child process: importing the package here would fix the library
# path) and select the -DDPX_TRACE variant
subprocess.run([sys.executable, "-c", "from deplex_amd import build as b; b.build_variant('trace')"], cwd=str(_root), check=True)
os.environ["DPX_LIB_PATH"] = str(_root / "deplex_amd" / "lib" / "variants" / "trace.so")
if len(sys.argv) > 2: os.environ["DPX_SKIP"] = sys.argv[2]
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import deplex_amd as D
from deplex_amd import _capi
import synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
B, S, G = 64, int(os.environ.get("DPX_BENCH_STREAMS", 4)), 8
K = synth.read_K(synth.TUM_K); intr = tuple(float(v) for v in (K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
dev = torch.device("cuda", 0)
d_in = [torch.from_numpy(synth.tum_stream(B, seed0=r * B).view(np.int16)).to(dev) for r in range(8)]
d_out = torch.empty((S * G, B, 480 * 640), dtype=torch.int32, device=dev)
pipe = D.Pipeline(480, 640, D.Config(), batch=B, slots=S, group=G)
def run(n):
    for i in range(n): pipe.submit_depth(d_in[i % 8].data_ptr(), B, *intr, d_out[i % (S * G)].data_ptr())
    pipe.sync()
run(512)
NS, cap = 65536, 768  # hardware wave slots, entries per slot
buf = torch.zeros((NS * cap, 4), dtype=torch.int64, device=dev)
cnt = (C.c_uint32 * NS)()
_capi.check(_capi.lib().dpx_debug_trace(0, buf.data_ptr(), cap, cnt))
import time
t0 = time.perf_counter(); run(steps); dt = time.perf_counter() - t0
_capi.check(_capi.lib().dpx_debug_trace(0, None, 0, cnt))
cnt = np.minimum(np.array(cnt[:], dtype=np.int64), cap)
used = np.nonzero(cnt)[0]
tr = np.concatenate([buf[s * cap: s * cap + cnt[s]].cpu().numpy().view(np.uint64) for s in used])
n = len(tr)
print(f"{len(used)} wave slots used, fullest holds {cnt.max()} of {cap} entries")
print(f"{steps} steps in {dt * 1e3:.2f} ms ({dt / steps * 1e6:.1f} us/batch with tracing), {n} waves traced")
st, en, hw, kw = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64), tr[:, 2], tr[:, 3]
ok = en > st
st, en, hw, kw = st[ok], en[ok], hw[ok], kw[ok]
kind = (kw & 0xF).astype(int); wid = ((kw >> 4) & 0xF).astype(int)
xcc = ((hw >> 32) & 0xF).astype(int); h32 = (hw & 0xFFFFFFFF).astype(np.int64)
simd = (h32 >> 4) & 3; cu = (h32 >> 8) & 0xF; sh = (h32 >> 12) & 1; se = (h32 >> 13) & 7
cu_id = ((xcc * 8 + se) * 2 + sh) * 16 + cu
_, cu_idx = np.unique(cu_id, return_inverse=True)
ncu = cu_idx.max() + 1
simd_idx = cu_idx * 4 + simd
names = {1: "planefit", 2: "prep", 3: "grow", 4: "tail", 5: "labels"}
vg = {1: 112, 2: 64, 3: 48, 4: 112, 5: 16}  # allocated vector registers per wave (blocks of 8)
lds = {2: 29.6, 3: 32.6, 4: 32.6}
T0, T1 = st.min(), en.max()
w0, w1 = T0 + (T1 - T0) // 4, T1 - (T1 - T0) // 4
print(f"{ncu} CUs seen; window {(w1 - w0) / 100:.0f} us of {(T1 - T0) / 100:.0f} us; wave lifetimes (us): " +
      ", ".join(f"{names[k]} {np.mean((en - st)[kind == k]) / 100:.1f}" for k in names if (kind == k).any()))
# sample the state at random instants of the window
rng = np.random.default_rng(0)
ts = np.sort(rng.integers(w0, w1, size=300))
nS = ncu * 4
acc_w = {k: np.zeros(nS) for k in names}; vg_use = np.zeros(nS); lds_use = np.zeros(ncu)
free_slots_hist = np.zeros(3)
can_pf = can_grow = can_prep = can_tail = 0.0
pf_dist = np.zeros(9)
joint = np.zeros((3, 3))
for t in ts:
    live = (st <= t) & (en > t)
    per_simd_vg = np.zeros(nS); per_simd_w = np.zeros(nS); per_cu_lds = np.zeros(ncu)
    for k in names:
        m = live & (kind == k)
        c = np.bincount(simd_idx[m], minlength=nS)
        acc_w[k] += c; per_simd_vg += c * vg[k]; per_simd_w += c
        if k == 1: pf_dist += np.bincount(np.minimum(c, 8), minlength=9)
        if k in lds:
            m0 = m & (wid == 0)  # the workgroup holds its LDS as long as its longest-living wave (wave 0) runs
            per_cu_lds += np.bincount(cu_idx[m0], minlength=ncu) * lds[k]
    vg_use += per_simd_vg; lds_use += per_cu_lds
    pf_cu = np.bincount(cu_idx[live & (kind == 1)], minlength=ncu)
    reg_cu = np.bincount(cu_idx[live & (kind >= 2) & (kind <= 4) & (wid == 0)], minlength=ncu)
    for a, lo, hi in ((0, 0, 0), (1, 1, 8), (2, 9, 99)):
        m = (pf_cu >= lo) & (pf_cu <= hi)
        joint[a, 0] += m.sum(); joint[a, 1] += reg_cu[m].sum(); joint[a, 2] += per_cu_lds[m].sum()
    vfree = (512 - per_simd_vg).reshape(ncu, 4); wfree = (8 - per_simd_w).reshape(ncu, 4); lfree = 160 - per_cu_lds
    # (workgroups of one wave per SIMD: plane fit 4 x 112, preparation 4 x 64, tail 4 x 112; grow one wave of 48)
    can_pf += np.mean((vfree >= 112).all(1) & (wfree >= 1).all(1))
    can_prep += np.mean((vfree >= 64).all(1) & (wfree >= 1).all(1) & (lfree >= 29.6))
    can_tail += np.mean((vfree >= 112).all(1) & (wfree >= 1).all(1) & (lfree >= 32.6))
    can_grow += np.mean((vfree >= 48).any(1) & (wfree >= 1).any(1) & (lfree >= 32.6))
ns = len(ts)
print("time-averaged resident waves per SIMD: " + ", ".join(f"{names[k]} {acc_w[k].mean() / ns:.2f}" for k in names))
print(f"VGPR file in use {vg_use.mean() / ns / 512 * 100:.0f} %, LDS in use {lds_use.mean() / ns / 160 * 100:.0f} %")
print("SIMDs with n plane-fit waves resident: " + ", ".join(f"{i}: {pf_dist[i] / pf_dist.sum() * 100:.0f} %" for i in range(6)))
for a, nm in enumerate(("no plane-fit wave", "1-8 plane-fit waves", "9-16 plane-fit waves")):
    if joint[a, 0]: print(f"CUs with {nm}: {joint[a, 0] / joint[:, 0].sum() * 100:.0f} % of samples, region workgroups resident there {joint[a, 1] / joint[a, 0]:.2f}, LDS in use {joint[a, 2] / joint[a, 0] / 160 * 100:.0f} %")
print(f"share of (CU, instant) samples where one more workgroup would fit: planefit {can_pf / ns * 100:.0f} %, prep {can_prep / ns * 100:.0f} %, "
      f"grow {can_grow / ns * 100:.0f} %, tail {can_tail / ns * 100:.0f} %")
