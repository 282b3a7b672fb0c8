#!/bin/bash
# usage: tools/pmc2.sh <bench_one name> <tag> "<counters>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc $3 --output-format csv -d gpurun_out/pmc_$2 -- python3 tools/bench_one.py $1 3 > gpurun_out/pmc_$2.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_$2/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
first = None
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if first is None: first = r["Counter_Name"]
        if r["Counter_Name"] == first: cnt[k] += 1
for k, d in agg.items():
    n = max(cnt[k], 1)
    if max(d.values()) / n < 1e5: continue
    print(k, "dispatches", n, {c: f"{v/n:.3e}" for c, v in d.items()})
PY
