#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_traffic.sh <tag> [<git commit of the tree, recorded in the JSON>]
# HBM traffic of every kernel of one bench.py step: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
# (MI355X_MICROARCH.md, HBM section); python3 directly after `--`.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 280 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${1}_$c -o vm -- \
      python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > gpurun_out/pmc_${1}_$c.log 2>&1 || exit 1
done
python3 tools/pmc_traffic_summary.py gpurun_out/pmc_${1}_FETCH_SIZE gpurun_out/pmc_${1}_WRITE_SIZE "${2:-unknown}" \
    "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events" \
    > gpurun_out/pmc_${1}_traffic.json
