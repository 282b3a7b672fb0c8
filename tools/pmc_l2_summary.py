"""Per-kernel L2 hit rate from one rocprofv3 --pmc pass (TCC_HIT_sum, TCC_MISS_sum).  Output: JSON
{kernel: {calls, hits, misses, hit_rate}} (hits / misses: requests per dispatch, averaged)."""
import collections
import csv
import glob
import json
import sys

tot = {"TCC_HIT_sum": collections.defaultdict(float), "TCC_MISS_sum": collections.defaultdict(float)}
cnt = collections.Counter()
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] not in tot:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[r["Counter_Name"]][k] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_HIT_sum":
            cnt[k] += 1
out = {}
for k in sorted(cnt):
    h, m, n = tot["TCC_HIT_sum"][k], tot["TCC_MISS_sum"][k], cnt[k]
    out[k] = {"calls": n, "hits": h / n, "misses": m / n, "hit_rate": h / max(h + m, 1.0)}
print(json.dumps(out, indent=1))
