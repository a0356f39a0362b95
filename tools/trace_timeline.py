"""Prints the tail of a rocprofv3 kernel trace as a timeline: trace_timeline.py kernel_trace.csv [n_last_planefits]
(start relative to the first listed kernel, duration, queue, short name) — used to see how a short bench run overlaps."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
npf = int(sys.argv[2]) if len(sys.argv) > 2 else 20
def short(n):
    m = re.search(r'k_regions2<\w+, \w+, (\d)>', n) or re.search(r'k_regions2ILi\dELb\dELi(\d)E', n)
    if 'planefit' in n: return 'pf'
    if 'region_prep' in n: return 'reg1'
    if 'regions' in n: return 'reg' + (m.group(1) if m else '')
    if 'labels' in n: return 'lab'
    return n[:24]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?')) for r in rows)
pf = [i for i, e in enumerate(ev) if e[2] == 'pf']
first = pf[-npf] if len(pf) >= npf else 0
t0 = ev[first][0]
for a, b, k, q in ev[first:]:
    print(f"{(a - t0) / 1e3:9.1f} us  +{(b - a) / 1e3:7.1f}  q{q:>3}  {k}")
print(f"span {(max(e[1] for e in ev[first:]) - t0) / 1e3:.1f} us")
