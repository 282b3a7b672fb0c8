"""Micro-benchmark (GPU) of the GEMM kernels at the C2 shapes (B=256, T=201): TFLOP/s per kernel variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from get_model_amd import kernels as K

M, D, F = 51456, 768, 3072
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(0)
def rb(*s): return (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(dev)
def rf(*s): return (torch.randn(*s, generator=g) * 0.5).to(dev)

def timeit(fn, flops, name, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name:34s} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s")

xD, xF, x3 = rb(M, D), rb(M, F), rb(M, 3 * D)
wqkv, wproj, w1, w2 = rb(3 * D, D), rb(D, D), rb(F, D), rb(D, F)
bD, bF, b3 = rf(D), rf(F), rf(3 * D)
resid = rf(M, D)
oD = torch.empty(M, D, dtype=torch.bfloat16, device=dev); oF = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
oF2 = torch.empty_like(oF); o3 = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
oDf = torch.empty(M, D, device=dev)
timeit(lambda: K.gemm_tn(K.EPI_BF16, xD, wqkv, o3, bias=b3), 2.0 * M * 3 * D * D, "qkv fwd  (bf16 +bias)")
timeit(lambda: K.gemm_tn(K.EPI_BIAS_RESID, xD, wproj, oDf, bias=bD, resid=resid), 2.0 * M * D * D, "proj fwd (bias_resid)")
timeit(lambda: K.gemm_tn(K.EPI_BIAS_GELU, xD, w1, oF, out1=oF2, bias=bF), 2.0 * M * F * D, "fc1 fwd  (bias_gelu)")
timeit(lambda: K.gemm_tn(K.EPI_BIAS_RESID, xF, w2, oDf, bias=bD, resid=resid), 2.0 * M * F * D, "fc2 fwd  (bias_resid)")
timeit(lambda: K.gemm_tn(K.EPI_DGELU, xD, w1, oF, aux=xF), 2.0 * M * F * D, "fc2 dgrad (dgelu)")
dbF = torch.empty(F, device=dev)
timeit(lambda: K.gemm_tn(K.EPI_BIAS_GELUG, xD, w1, oF, out1=oF2, bias=bF), 2.0 * M * F * D, "fc1 fwd  (gelu + gelu', train)")
timeit(lambda: K.gemm_tn(K.EPI_DMUL, xD, w1, oF, aux=xF, dbias=dbF), 2.0 * M * F * D, "fc2 dgrad (dmul + colsum, train)")
timeit(lambda: K.gemm_tn(K.EPI_BF16, xD, w1, oF, bias=bF), 2.0 * M * F * D, "N=3072 K=768 plain bf16 +bias")
timeit(lambda: K.gemm_tn(K.EPI_BF16, xF, w2, oD), 2.0 * M * F * D, "fc1 dgrad (bf16, K=3072)")
wqkvT = rb(D, 3 * D)
timeit(lambda: K.gemm_tn(K.EPI_BF16, x3, wqkvT, oD), 2.0 * M * 3 * D * D, "qkv dgrad (bf16, K=2304)")
timeit(lambda: K.gemm_tn(K.EPI_BF16, xD, wproj, oD), 2.0 * M * D * D, "proj dgrad (bf16, K=768)")
timeit(lambda: K.gemm_wgrad(xD, xD), 2.0 * M * D * D, "wgrad proj 768x768")
timeit(lambda: K.gemm_wgrad(x3, xD), 2.0 * M * 3 * D * D, "wgrad qkv 2304x768")
timeit(lambda: K.gemm_wgrad_pair(xD, xD, x3, xD), 2.0 * M * 4 * D * D, "wgrad proj+qkv (one launch)")
timeit(lambda: K.gemm_wgrad(xF, xD), 2.0 * M * F * D, "wgrad fc1 3072x768")
timeit(lambda: K.gemm_wgrad(xD, xF), 2.0 * M * F * D, "wgrad fc2 768x3072")
B, T, H = 256, 201, 12
out, lse = K.attention_fwd(x3, B, T, H, 0.125)
timeit(lambda: K.attention_fwd(x3, B, T, H, 0.125), 4.0 * B * H * T * T * 64, "attention fwd")
timeit(lambda: K.attention_bwd(x3, out, xD, lse, B, T, H, 0.125), 10.0 * B * H * T * T * 64, "attention bwd")
