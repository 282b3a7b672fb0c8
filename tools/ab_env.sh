#!/bin/bash
# usage (GPU box, repo root): bash tools/ab_env.sh <rounds> "<ENV=val ...>" "<ENV=val ...>" ...   ("-" = no extra environment)
# Interleaved A/B of environment switches in the training step (bench.py, no events, no CPU baseline): ms/step per run.
R=$1; shift
for i in $(seq $R); do
  for E in "$@"; do
    if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
    env $EE timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events $AB_BENCH_ARGS \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$E]', round(d['ms_per_step'],3))" || exit 1
  done
done
