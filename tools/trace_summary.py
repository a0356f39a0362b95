"""Summarise a rocprofv3 kernel trace CSV: per-kernel durations and average concurrency in the steady-state window."""
import csv, re, sys, random
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r['Kernel_Name']
    m = re.search(r'k_regions2<\w+, \w+, (\d)>', n) or re.search(r'k_regions2ILb\dELb\dELi(\d)E', n)
    k = 'pf' if 'planefit' in n else 'reg1' if 'region_prep' in n else ('reg' + (m.group(1) if m else '')) if 'regions' in n else 'lab' if 'labels' in n else 'sort' if 'k_sort' in n else 'graph' if 'cellgraph' in n else None
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
tot = [sum(1 for a, b, kk in ev if a <= t < b) for t in ts]
print(f"  all kernels: avg concurrent {sum(tot)/len(tot):.2f}, max {max(tot)}, idle fraction {sum(1 for c in tot if c == 0)/len(tot):.2f}")
