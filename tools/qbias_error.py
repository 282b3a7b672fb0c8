"""Diagnostic (GPU): error of attn.q_bias.grad ALONE (not grouped with v_bias) against the fp32 oracle, per layer, on the
12-layer golden case and a B=16 case:
    GETMI_ATTN_LO=0 python tools/qbias_error.py     # delta = rowsum(dO * bf16(O))
    python tools/qbias_error.py                     # delta = rowsum(dO * (bf16(O) + residual))   (the default)
Prints |g - ref| / |ref| (L2) and the max-abs error relative to max|ref| per layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import get_cpu as O
from get_model_amd.config import build_model, region_model_yaml

torch.set_num_threads(16)
for (L, B, N, seed) in ((12, 4, 200, 3), (4, 16, 200, 8)):
    sd = O.fill_state_dict("finetune", seed=seed, num_layers=L)
    batch = O.make_batch(B, N, seed)
    _, _, og = O.loss_and_grads("finetune", sd, batch, num_layers=L)
    m = build_model(region_model_yaml("finetune", num_layers=L)); m.load_state_dict(sd); m = m.cuda().eval()
    gb = {k: v.cuda() for k, v in batch.items()}
    out = m(**m.get_input(gb)); pred, obs = m.before_loss(out, gb); m.after_loss(m.loss(pred, obs)).backward()
    print(f"L={L} B={B} N={N}  GETMI_ATTN_LO={os.environ.get('GETMI_ATTN_LO', '1')}")
    worst = 0.0
    for i in range(L):
        for nm in ("q_bias", "v_bias"):
            g = getattr(m.encoder.blocks[i].attn, nm).grad.float().cpu(); r = og[f"encoder.blocks.{i}.attn.{nm}"]
            l2 = ((g - r).norm() / r.norm()).item(); mx = ((g - r).abs().max() / r.abs().max()).item()
            if nm == "q_bias":
                worst = max(worst, mx)
            print(f"  layer {i:2d} {nm}: L2 rel {l2:.4f}  max-abs/max {mx:.4f}  |ref| {r.norm().item():.3e}")
    print(f"  worst q_bias max-abs/max: {worst:.4f}")
