#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_mfma.sh <tag>
# MFMA / LDS pipe occupancy of every kernel of one bench.py step: one rocprofv3 --pmc pass (3 SQ counters + 1 GRBM);
# python3 directly after `--`.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv \
    -d gpurun_out/pmc_${1}_mfma -o vm -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events \
    > gpurun_out/pmc_${1}_mfma.log 2>&1 || exit 1
python3 tools/pmc_mfma_summary.py gpurun_out/pmc_${1}_mfma > gpurun_out/pmc_${1}_mfma.json
