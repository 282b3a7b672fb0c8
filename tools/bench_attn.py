"""Attention kernels at the bench shape (T = 201) and at long context (T = 2001)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from get_model_amd import kernels as K

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for B, T in ((256, 201), (16, 2001), (64, 129)):
    H = 12
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.randn(B * T, 3 * H * 64, generator=g) * 0.5).bfloat16().cuda()
    do = (torch.randn(B * T, H * 64, generator=g) * 0.5).bfloat16().cuda()
    out, lse = K.attention_fwd(x, B, T, H, 0.125)
    f = timeit(lambda: K.attention_fwd(x, B, T, H, 0.125))
    b = timeit(lambda: K.attention_bwd(x, out, do, lse, B, T, H, 0.125))
    fl = 4.0 * B * H * T * T * 64
    if T == 201:      # as the training step calls it: residual of the output + fused bias sums
        out2, lse2, lo = K.attention_fwd(x, B, T, H, 0.125, want_lo=True)
        b2 = timeit(lambda: K.attention_bwd(x, out2, do, lse2, B, T, H, 0.125, want_bias=True, out_lo=lo))
        print(f"B={B} T={T} (out_lo + bias sums): bwd {b2:8.1f} us {2.5 * fl / b2 / 1e6:7.1f} TF", flush=True)
    print(f"B={B} T={T}: fwd {f:8.1f} us {fl / f / 1e6:7.1f} TF   bwd {b:8.1f} us {2.5 * fl / b / 1e6:7.1f} TF", flush=True)
