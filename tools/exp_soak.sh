for K in 20 200 1000 20; do
  timeout -k 10 300 python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-kernel-events | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps $K', round(d['ms_per_step'],3), round(d['ms_per_step_median'],3))" || exit 1
done
