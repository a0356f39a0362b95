"""Prints the top rows of a rocprofv3 kernel_stats.csv.  usage: kstats_print.py file.csv [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
print(sys.argv[1])
for r in rows[:n]:
    print(f"  {r['Name'][:62]:62s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs']) / 1e3:10.1f} total_ms {float(r['TotalDurationNs']) / 1e6:9.2f} {r['Percentage']}%")
