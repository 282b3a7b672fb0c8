#!/bin/bash
# usage (GPU box, repo root): bash tools/ab_step.sh <libA.so> <libB.so> [rounds] [extra bench args]
# Interleaved A/B of two builds of the library in the training step (bench.py, no events, no CPU baseline): ms/step per run.
A=$1; B=$2; R=${3:-3}; shift 3
for i in $(seq $R); do
  for L in $A $B; do
    GETMI_LIB=$PWD/$L timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events "$@" \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],3))" || exit 1
  done
done
