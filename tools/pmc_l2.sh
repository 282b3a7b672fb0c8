#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_l2.sh <tag>
# L2 hit rate of every kernel of one bench.py step: TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) (MI355X_MICROARCH.md, L2
# section), one --pmc pass of its own; python3 directly after `--`.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 280 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_${1}_l2 -o vm -- \
    python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > gpurun_out/pmc_${1}_l2.log 2>&1 || exit 1
python3 tools/pmc_l2_summary.py gpurun_out/pmc_${1}_l2 > gpurun_out/pmc_${1}_l2.json
