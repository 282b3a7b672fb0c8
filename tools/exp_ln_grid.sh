# usage (GPU box): bash tools/exp_ln_grid.sh  -- LayerNorm backward grid cap (workgroups of 4 rows-in-flight) swept in the bench
for G in 1024 256 384 512 640 768; do
  GETMI_LN_BWD_GRID=$G timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('bwd grid $G', 'step', round(d['ms_per_step'],3), 'ln_bwd', round(k['ln_bwd_kernel']['avg_us'],2), 'ln_fwd', round(k['ln_fwd_kernel']['avg_us'],2))" || exit 1
done
for G in 512 768 1024; do
  GETMI_LN_BWD_GRID=512 GETMI_LN_GRID=$G timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('fwd grid $G', 'step', round(d['ms_per_step'],3), 'ln_bwd', round(k['ln_bwd_kernel']['avg_us'],2), 'ln_fwd', round(k['ln_fwd_kernel']['avg_us'],2))" || exit 1
done
