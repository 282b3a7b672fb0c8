#!/bin/bash
# usage (GPU box, repo root): bash tools/ab_step.sh <rounds> <lib1.so> <lib2.so> ...
# Interleaved A/B of several builds of the library in the training step (bench.py, no events, no CPU baseline): ms/step per run.
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    GETMI_LIB=$PWD/$L timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events $AB_BENCH_ARGS \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],3))" || exit 1
  done
done
