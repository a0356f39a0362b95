"""Timeline of the 20-step run the driver times: the kernels of the timed region (the LAST 20 submissions of a
`bench.py --steps 20 --warmup 5 --no-sustained --no-end-to-end --no-cpu-baseline --no-cloud-roofline` run under
rocprofv3 --kernel-trace), start / end relative to the first of them, per queue.
usage: short_run_timeline.py kernel_trace.csv"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def kind(n):
    m = re.search(r'k_regions2<\w+, \w+, (\d)>', n)
    if 'planefit_depth' in n: return 'pf'
    if 'region_prep' in n: return 'reg1'
    if 'regions' in n: return 'reg' + (m.group(1) if m else '')
    if 'labels' in n: return 'lab'
    return None
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), kind(r['Kernel_Name']), r['Queue_Id'], r.get('Grid_Size', r.get('Grid_Size_X', '')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', ''))) for r in rows if kind(r['Kernel_Name']))
# the pipeline part of the run = pipeline kernels before the stand-alone plane-fit series: take the label launches;
# the timed region's four groups are the last four label launches that follow a region launch on the same queue
labs = [i for i, x in enumerate(ev) if x[2] == 'lab']
if len(labs) < 4:
    sys.exit("no pipeline groups in the trace")
last = labs[-4:]
queues = {ev[i][3] for i in last}
t_end = max(ev[i][1] for i in last)
# walk back: on each of those queues the kernels of the last group
sel = []
for q in queues:
    L = [x for x in ev if x[3] == q and x[1] <= t_end]
    grp = []
    for x in reversed(L):
        grp.append(x)
        if x[2] == 'pf':
            break
    sel += grp
t0 = min(x[0] for x in sel)
for x in sorted(sel):
    print(f"queue {x[3]:>3} {x[2]:5s} start {(x[0] - t0) / 1e3:8.1f} us  end {(x[1] - t0) / 1e3:8.1f} us  dur {(x[1] - x[0]) / 1e3:7.1f} us")
print(f"timed region (first start to last end): {(max(x[1] for x in sel) - t0) / 1e3:.1f} us")
