"""Summarise a rocprofv3 kernel trace CSV: per-kernel durations and average concurrency in the steady-state window."""
import csv, sys, random
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r['Kernel_Name']
    k = 'pf' if 'planefit' in n else ('reg' + (n[n.find('Li') + 2] if 'regions2' in n and 'ELi' in n else '')) if 'regions' in n else 'lab' if 'labels' in n else None
    if k: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k))
ev.sort()
s = ev[len(ev) // 4][0]; e = ev[3 * len(ev) // 4][1]
kinds = sorted(set(k for *_, k in ev))
random.seed(0); ts = [random.uniform(s, e) for _ in range(300)]
nsteps = sum(1 for a, b, k in ev if k == 'pf' and a >= s and b <= e)
print(f"window {(e-s)/1e3:.0f} us, {nsteps} plane-fit launches -> {(e-s)/1e3/max(nsteps,1):.1f} us/step")
for k in kinds:
    d = [(b - a) / 1e3 for a, b, kk in ev if kk == k and a >= s and b <= e]
    c = [sum(1 for a, b, kk in ev if kk == k and a <= t < b) for t in ts]
    if d: print(f"  {k}: n={len(d)} avg {sum(d)/len(d):.1f} us (min {min(d):.1f}, max {max(d):.1f}); avg concurrent {sum(c)/len(c):.2f}")
