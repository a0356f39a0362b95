"""Per-launch average of one PMC counter for the kernels whose name contains a substring (rocprofv3 counter_collection.csv)."""
import csv, sys
from collections import defaultdict
path, counter, sub = sys.argv[1], sys.argv[2], sys.argv[3]
tot = defaultdict(float); launches = defaultdict(set)
for r in csv.DictReader(open(path)):
    if r.get("Counter_Name") != counter or sub not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"].split("(")[0]
    tot[k] += float(r["Counter_Value"])
    launches[k].add(r.get("Dispatch_Id") or r.get("Correlation_Id"))
for k in tot:
    n = max(len(launches[k]), 1)
    print(f"{k}: {counter} total {tot[k]:.0f} over {n} launches -> {tot[k]/n:.1f} per launch")
