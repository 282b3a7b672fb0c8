#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_bench.sh <tag>   -> gpurun_out/prof_<tag>/
# rocprofv3 kernel trace + stats of a short bench.py run; python3 directly after `--` (no exec hop).
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$1" -o vm -- \
    python3 "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events \
    > "$GRAFT_REPO_ROOT/gpurun_out/prof_$1.log" 2>&1
