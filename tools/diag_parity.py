"""Diagnostic (GPU): per-parameter gradient error of the HIP path vs the fp32 oracle for one golden case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tests.golden_util import load_case
from tests.test_model_parity_gpu import run_hip
from oracle import get_cpu as O

name = sys.argv[1] if len(sys.argv) > 1 else "ft_L12_B4_N200_eval"
c = load_case(name)
loss, pred, grads = run_hip(c, c["train"])
torch.set_num_threads(16)
oloss, opred, ograds = O.loss_and_grads(c["kind"], c["sd"], c["batch"], num_layers=c["L"], keep_masks=c["keep"])
print("loss", loss.item(), oloss.item(), "pred err", ((pred.cpu() - opred).abs().max() / opred.abs().max()).item())
rows = []
for n, g in grads.items():
    g = g.detach().float().cpu().flatten(); r = ograds[n].flatten()
    cos = F.cosine_similarity(g, r, dim=0).item()
    rel = ((g - r).norm() / r.norm()).item()
    mx = ((g - r).abs().max() / r.abs().max()).item()
    rows.append((rel, n, cos, mx, r.norm().item()))
for rel, n, cos, mx, rn in sorted(rows, reverse=True)[:25]:
    print(f"{n:45s} relL2 {rel:.3e} cos {cos:.5f} maxabs/max {mx:.3e} |ref| {rn:.3e}")
