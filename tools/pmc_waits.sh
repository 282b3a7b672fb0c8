#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_waits.sh <tag>
# Where the waves of every kernel of one bench.py step spend their cycles (8 SQ counters, one pass).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 280 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM \
    --output-format csv -d gpurun_out/pmc_${1}_waits -o vm -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events \
    > gpurun_out/pmc_${1}_waits.log 2>&1 || exit 1
python3 - "gpurun_out/pmc_${1}_waits" > gpurun_out/pmc_${1}_waits.json <<'PY'
import collections, csv, glob, json, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
out = {}
for k, c in tot.items():
    w = c.get("SQ_WAVE_CYCLES", 0.0)
    if w <= 0: continue
    out[k] = {"calls": cnt[k], "wave_cycles": w / max(cnt[k], 1)}
    for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
        out[k][n.replace("SQ_", "").lower() + "_share"] = c.get(n, 0.0) / w
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["wave_cycles"] * kv[1]["calls"])), indent=1))
PY
