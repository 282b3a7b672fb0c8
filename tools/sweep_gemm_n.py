"""Plain bf16 GEMM (+bias) at M = 51456, K = 768 over output widths N: time per launch, per round of tiles, TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from get_model_amd import kernels as K
M, Kd = 51456, int(os.environ.get("SWEEP_K", "768"))
g = torch.Generator(device="cpu").manual_seed(0)
x = (torch.randn(M, Kd, generator=g) * 0.5).to(torch.bfloat16).cuda()
for N in (768, 1024, 1536, 2048, 2304, 2560, 3072, 4096):
    w = (torch.randn(N, Kd, generator=g) * 0.5).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda()
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    fn = lambda: K.gemm_tn(K.EPI_BF16, x, w, o, bias=b)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    t256 = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"N={N:5d}  {us:8.1f} us  {2.0 * M * N * Kd / us / 1e6:7.1f} TF   tiles(256) {t256:5d} = {t256 / 256:.2f} rounds  "
          f"us per 256-tile-round-equivalent {us / (t256 / 256):.1f}")
