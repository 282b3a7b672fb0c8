#!/bin/bash
# usage: tools/pmc.sh <bench_one name> <out tag> : SQ counter pass for one kernel config (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_$2 -- python3 tools/bench_one.py $1 3 > gpurun_out/pmc_$2.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_$2/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, d in agg.items():
    if d.get("SQ_WAVE_CYCLES", 0) < 1e6: continue
    n = cnt[k]
    print(k, "dispatches", n, {c: f"{v/n:.3e}" for c, v in d.items()})
PY
