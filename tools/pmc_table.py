"""Per-kernel table of PMC counters from a rocprofv3 counter_collection.csv: pmc_table.py file.csv
(average per launch of every counter present; kernel names shortened)."""
import csv, re, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); launches = defaultdict(set); counters = []
def short(n):
    n = n.split("(")[0]
    n = re.sub(r"^void ", "", n)
    return n[-70:]
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r["Kernel_Name"]); c = r["Counter_Name"]
    if c not in counters: counters.append(c)
    tot[k][c] += float(r["Counter_Value"])
    launches[k].add(r.get("Dispatch_Id") or r.get("Correlation_Id"))
print(f"{'kernel':70s} {'n':>5s} " + " ".join(f"{c[-16:]:>16s}" for c in counters))
for k in sorted(tot, key=lambda k: -sum(tot[k].values())):
    n = max(len(launches[k]), 1)
    print(f"{k:70s} {n:5d} " + " ".join(f"{tot[k][c] / n:16.0f}" for c in counters))
