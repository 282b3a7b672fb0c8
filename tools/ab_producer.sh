#!/bin/bash
# usage (GPU box, repo root): bash tools/ab_producer.sh <rounds>  -- the training step with one resident batch vs each batch drawn through RegionBatchProducer
for i in $(seq $1); do
  for A in "" "--producer"; do
    timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events $A \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$A]', d['batch_source'], round(d['ms_per_step'],3))" || exit 1
  done
done
