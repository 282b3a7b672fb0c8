"""Times the memory-bound kernels of one encoder layer at the bench shape (M = 256 x 201 rows, D = 768):
prints microseconds and effective HBM bandwidth (algorithmic bytes / time)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from get_model_amd import kernels as K  # noqa: E402

dev = torch.device("cuda", 0)
B, T, D, H = 256, 201, 768, 12
M = B * T


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def report(name, us, nbytes):
    print(f"{name:28s} {us:8.1f} us   {nbytes / us * 1e-6:6.2f} TB/s", flush=True)


torch.manual_seed(0)
x = torch.randn(M, D, device=dev)
gam = torch.randn(D, device=dev)
bet = torch.randn(D, device=dev)
dy = torch.randn(M, D, device=dev).bfloat16()
dres = torch.randn(M, D, device=dev)
scale = torch.rand(B, device=dev) + 0.5
y, mean, rstd = K.layernorm_fwd(x, gam, bet, 1e-6)
report("ln_fwd", timeit(lambda: K.layernorm_fwd(x, gam, bet, 1e-6)), M * D * 6)
report("ln_bwd (dres)", timeit(lambda: K.layernorm_bwd(dy, x, mean, rstd, gam, dres=dres)), M * D * (4 + 2 + 4 + 4))
report("ln_bwd (dres,dz,dbias)", timeit(lambda: K.layernorm_bwd(dy, x, mean, rstd, gam, dres=dres, rowscale=scale,
                                                                 rows_per_sample=T, want_dz=True, want_dbias=True)),
       M * D * (4 + 2 + 4 + 4 + 2))
report("scale_cast", timeit(lambda: K.scale_cast(dres, scale, T)), M * D * 6)
dqkv = torch.randn(M, 3 * D, device=dev).bfloat16()
report("colsum_bf16 q third", timeit(lambda: K.colsum_bf16(dqkv[:, :D])), M * D * 2)
report("colsum_bf16 full 2304", timeit(lambda: K.colsum_bf16(dqkv)), M * D * 6)
qkv = torch.randn(M, 3 * D, device=dev).bfloat16()
a, lse = K.attention_fwd(qkv, B, T, H, 0.125)
da = torch.randn(M, D, device=dev).bfloat16()
report("attention_bwd (w/ delta)", timeit(lambda: K.attention_bwd(qkv, a, da, lse, B, T, H, 0.125)), M * D * 2 * 8)
