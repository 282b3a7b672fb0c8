"""Run ONE kernel configuration a few times (for rocprofv3 --pmc passes).  usage: bench_one.py <name> [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from get_model_amd import kernels as K
name = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
M, D, F = 51456, 768, 3072
g = torch.Generator(device="cpu").manual_seed(0)
def rb(*s): return (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).cuda()
def rf(*s): return (torch.randn(*s, generator=g) * 0.5).cuda()
if name == "wgrad_fc2":
    x, y = rb(M, D), rb(M, F); fn = lambda: K.gemm_wgrad(x, y)
elif name == "tn_k3072":
    x, w = rb(M, F), rb(D, F); o = torch.empty(M, D, dtype=torch.bfloat16, device="cuda"); fn = lambda: K.gemm_tn(K.EPI_BF16, x, w, o)
elif name == "tn_fc1":
    x, w, b = rb(M, D), rb(F, D), rf(F); o = torch.empty(M, F, dtype=torch.bfloat16, device="cuda"); o2 = torch.empty_like(o)
    fn = lambda: K.gemm_tn(K.EPI_BIAS_GELU, x, w, o, out1=o2, bias=b)
elif name == "tn_qkv":
    x, w, b = rb(M, D), rb(3 * D, D), rf(3 * D); o = torch.empty(M, 3 * D, dtype=torch.bfloat16, device="cuda")
    fn = lambda: K.gemm_tn(K.EPI_BF16, x, w, o, bias=b)
elif name == "attn_fwd":
    x = rb(M, 3 * D); fn = lambda: K.attention_fwd(x, 256, 201, 12, 0.125)
elif name == "attn_bwd":
    x = rb(M, 3 * D); out, lse = K.attention_fwd(x, 256, 201, 12, 0.125); do = rb(M, D)
    fn = lambda: K.attention_bwd(x, out, do, lse, 256, 201, 12, 0.125)
for _ in range(iters): fn()
torch.cuda.synchronize()
