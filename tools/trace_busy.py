"""Which launches fill the time?  From a rocprofv3 kernel trace (csv) of a bench run, over the middle half of the run: per
kernel the share of wall time during which at least one launch of it is running (union of its intervals), the average
number of its launches in flight, its average duration; per queue the share of wall time with nothing running.
usage: trace_busy.py kernel_trace.csv [rows]"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 16
def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    return n.replace('dpx::', '')
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r['Queue_Id']) for r in rows)
s = ev[len(ev) // 4][0]; e = ev[3 * len(ev) // 4][1]
win = [(max(a, s), min(b, e), k, q) for a, b, k, q in ev if b > s and a < e]
wall = e - s
def union(iv):
    iv = sorted(iv); tot = 0; ca, cb = iv[0]
    for a, b in iv[1:]:
        if a > cb: tot += cb - ca; ca, cb = a, b
        else: cb = max(cb, b)
    return tot + cb - ca
byk = defaultdict(list); byq = defaultdict(list)
for a, b, k, q in win: byk[k].append((a, b)); byq[q].append((a, b))
print(f"window {wall / 1e6:.2f} ms, {len(win)} launches on {len(byq)} queues; something running {union([(a, b) for a, b, _, _ in win]) / wall:.3f} of the time, "
      f"{sum(b - a for a, b, _, _ in win) / wall:.2f} launches in flight on average")
print(f"{'kernel':58s} {'n':>6s} {'avg us':>9s} {'running':>8s} {'in flight':>9s}")
for k, iv in sorted(byk.items(), key=lambda kv: -sum(b - a for a, b in kv[1]))[:nshow]:
    tot = sum(b - a for a, b in iv)
    print(f"{k[:58]:58s} {len(iv):6d} {tot / len(iv) / 1e3:9.1f} {union(iv) / wall:8.3f} {tot / wall:9.2f}")
idle = [1 - union(iv) / wall for iv in byq.values()]
print(f"queues idle: min {min(idle):.3f} avg {sum(idle) / len(idle):.3f} max {max(idle):.3f}")
