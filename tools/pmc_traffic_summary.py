"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command.

Units / corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 B, so it is doubled; WRITE_SIZE is
exact for 16-B-per-lane streaming stores.  Output: JSON {kernel: {calls, fetch_bytes, write_bytes}} (averages per
dispatch, bytes) + "_meta" {commit, command} when given as 3rd / 4th argument."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


ft, fc = load(sys.argv[1], "FETCH_SIZE")
wt, wc = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(ft) | set(wt)):
    n = max(fc[k], wc[k], 1)
    out[k] = {"calls": n, "fetch_bytes": 2.0 * 1024.0 * ft[k] / max(fc[k], 1), "write_bytes": 1024.0 * wt[k] / max(wc[k], 1)}
if len(sys.argv) > 3:      # provenance for bench.py's roofline.traffic_source: commit the profile was taken at + command
    out["_meta"] = {"commit": sys.argv[3], "command": sys.argv[4] if len(sys.argv) > 4 else None}
print(json.dumps(out, indent=1))
