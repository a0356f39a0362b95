"""Steady-state view of a rocprofv3 kernel trace of bench.py: per kernel kind the duration (avg / min) and, per queue,
the gap between the end of one kernel and the start of the next one of the same queue; average concurrency.
usage: trace_streams.py kernel_trace.csv"""
import csv, re, sys, random
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
def kind(n):
    m = re.search(r'k_regions2<\w+, \w+, (\d)>', n) or re.search(r'k_regions2ILi\dELb\dELi(\d)E', n)
    if 'planefit' in n: return 'pf'
    if 'region_prep' in n: return 'reg1'
    if 'regions' in n: return 'reg' + (m.group(1) if m else '')
    if 'labels' in n: return 'lab'
    return None
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), kind(r['Kernel_Name']), r['Queue_Id']) for r in rows if kind(r['Kernel_Name']))
s = ev[len(ev) // 4][0]; e = ev[3 * len(ev) // 4][1]
win = [x for x in ev if x[0] >= s and x[1] <= e]
byq = defaultdict(list)
for a, b, k, q in win: byq[q].append((a, b, k))
gaps = defaultdict(list)
for q, L in byq.items():
    L.sort()
    for (a0, b0, k0), (a1, b1, k1) in zip(L, L[1:]):
        gaps[f"{k0}->{k1}"].append((a1 - b0) / 1e3)
dur = defaultdict(list)
for a, b, k, q in win: dur[k].append((b - a) / 1e3)
npf = len(dur['pf'])
print(f"window {(e - s) / 1e3:.0f} us, {npf} plane-fit launches, {len(byq)} queues -> {(e - s) / 1e3 / max(npf, 1):.1f} us per group")
for k in sorted(dur):
    d = dur[k]; print(f"  {k}: n={len(d)} avg {sum(d) / len(d):.1f} us  min {min(d):.1f}  max {max(d):.1f}")
for k in sorted(gaps):
    g = gaps[k]; print(f"  gap {k}: n={len(g)} avg {sum(g) / len(g):.1f} us  min {min(g):.1f}  max {max(g):.1f}")
random.seed(0); ts = [random.uniform(s, e) for _ in range(400)]
for k in sorted(dur):
    c = [sum(1 for a, b, kk, q in win if kk == k and a <= t < b) for t in ts]
    print(f"  concurrent {k}: {sum(c) / len(c):.2f}")
tot = [sum(1 for a, b, kk, q in win if a <= t < b) for t in ts]
print(f"  all: avg concurrent {sum(tot) / len(tot):.2f}, max {max(tot)}")
