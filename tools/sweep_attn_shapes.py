"""Diagnostic (GPU): attention forward + backward against an fp32 reference over edge-case context lengths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from get_model_amd import kernels as K

bad = 0
for T in (1, 2, 7, 31, 33, 63, 64, 65, 191, 192, 193, 223, 224, 225, 255, 256, 257, 447, 448, 449, 671, 673, 1000):
    for B, H, lo in ((1, 1, True), (2, 3, False)):
        g = torch.Generator().manual_seed(T * 7 + H)
        qkv = (torch.randn(B * T, 3 * H * 64, generator=g) * 1.2).bfloat16().cuda()
        dout = torch.randn(B * T, H * 64, generator=g).bfloat16().cuda()
        qr = qkv.float().requires_grad_(True)
        q, k, v = qr.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
        o = (((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B * T, H * 64)
        o.backward(dout.float())
        if lo:
            out, lse, olo = K.attention_fwd(qkv, B, T, H, 0.125, want_lo=True)
        else:
            (out, lse), olo = K.attention_fwd(qkv, B, T, H, 0.125), None
        d, dqb, dvb = K.attention_bwd(qkv, out, dout, lse, B, T, H, 0.125, want_bias=True, out_lo=olo)
        torch.cuda.synchronize()
        gref = qr.grad
        e_o = (out.float() - o.detach()).abs().max().item()
        e_g = (d.float() - gref).abs().max().item() / max(gref.abs().max().item(), 1e-9)
        D = H * 64
        # (T = 1: dQ is exactly 0 and so is the reference sum; scale by the whole gradient then)
        e_qb = (dqb - gref[:, :D].sum(0)).abs().max().item() / max(gref[:, :D].abs().sum(0).max().item(), 1e-3 * gref.abs().max().item())
        e_vb = (dvb - gref[:, 2 * D:].sum(0)).abs().max().item() / max(gref[:, 2 * D:].abs().sum(0).max().item(), 1e-9)
        ok = torch.isfinite(d.float()).all().item() and e_o < 2e-2 and e_g < 1.5e-2 and e_qb < 4e-3 and e_vb < 4e-3
        bad += not ok
        print(f"T={T:5d} B={B} H={H} lo={int(lo)}  out {e_o:.2e}  dqkv {e_g:.2e}  dq_bias {e_qb:.2e}  dv_bias {e_vb:.2e}  {'ok' if ok else 'BAD'}", flush=True)
print("SWEEP", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
