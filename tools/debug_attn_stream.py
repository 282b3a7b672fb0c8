"""Where does the streamed attention backward differ from the fp32 reference?  (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from get_model_amd import kernels as K

def ref(qkv, dout, B, T, H, scale):
    qr = qkv.float().requires_grad_(True)
    q, k, v = qr.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q * scale) @ k.transpose(-2, -1)
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B * T, H * 64)
    o.backward(dout.float())
    return qr.grad

for B, T, H, lo in ((1, 201, 1, True), (1, 256, 1, True), (1, 513, 2, False), (1, 2001, 1, True)):
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn(B * T, 3 * H * 64, generator=g) * 1.5).bfloat16().cuda()
    dout = torch.randn(B * T, H * 64, generator=g).bfloat16().cuda()
    if lo:
        out, lse, olo = K.attention_fwd(qkv, B, T, H, 0.125, want_lo=True)
    else:
        (out, lse), olo = K.attention_fwd(qkv, B, T, H, 0.125), None
    d = K.attention_bwd(qkv, out, dout, lse, B, T, H, 0.125, out_lo=olo).float().view(B, T, 3, H, 64)
    r = ref(qkv, dout, B, T, H, 0.125).view(B, T, 3, H, 64)
    print(f"B={B} T={T} H={H} lo={lo}")
    for i, name in enumerate(("dQ", "dK", "dV")):
        e = (d[:, :, i] - r[:, :, i])
        nan = torch.isnan(d[:, :, i])
        rows = nan.any(-1).any(0).any(-1).nonzero().flatten().tolist()
        e = torch.nan_to_num(e, nan=0.0)
        per_slice = [round(e[:, t0:t0 + 224].abs().max().item(), 4) for t0 in range(0, T, 224)]
        print(f"  {name}: nan rows {rows[:12]}{'...' if len(rows) > 12 else ''} n={len(rows)}  max|ref| {r[:, :, i].abs().max().item():.3f}  err per 224 rows {per_slice}")
