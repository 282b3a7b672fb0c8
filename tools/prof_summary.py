"""Prints the per-kernel share table of a rocprofv3 *_kernel_stats.csv (steps = profiled steps incl. warm-up)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time {tot / 1e6:.2f} ms over {steps} steps = {tot / 1e6 / steps:.2f} ms/step")
for r in rows[:30]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']):5d} {float(r['AverageNs']) / 1e3:9.1f} us {float(r['TotalDurationNs']) / tot * 100:6.2f} %")
