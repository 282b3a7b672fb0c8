"""Shared helpers: load a golden fixture (made by oracle/make_golden.py from the reference itself),
re-create its inputs with the deterministic filler, compare predictions / loss / gradient summaries."""
import glob
import os

import numpy as np
import torch

from oracle import get_cpu as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    L, B, N, train, seed = (int(v) for v in z["meta"])
    kind = str(z["kind"])
    sd = O.fill_state_dict(kind, seed=seed, num_layers=L)
    batch = O.make_batch(B, N, seed, kind=kind)
    keep = None
    if train:
        keep = []
        for i in range(L):
            k1 = (O.hash_tensor((B,), 9000 + seed * 100 + 2 * i, 0.0, 1.0) < 0.7).float()
            k2 = (O.hash_tensor((B,), 9001 + seed * 100 + 2 * i, 0.0, 1.0) < 0.7).float()
            keep.append((k1, k2))
    return dict(z=z, kind=kind, L=L, B=B, N=N, train=bool(train), seed=seed, sd=sd, batch=batch, keep=keep)


def check_against_golden(case, loss, pred, grads, rtol_pred, rtol_loss, rtol_gnorm, atol_gval_rel):
    """grads: {name: tensor}.  Gradient check: L2 norm within rtol_gnorm, sampled elements within
    atol_gval_rel * max|sample| (a normalised max-abs test)."""
    z = case["z"]
    gp = torch.from_numpy(z["pred"])
    scale = gp.abs().max().item()
    err = (pred.detach().float().cpu() - gp).abs().max().item()
    assert err <= rtol_pred * scale, f"pred max-abs err {err:.3e} vs scale {scale:.3e}"
    gl = float(z["loss"])
    assert abs(float(loss) - gl) <= rtol_loss * abs(gl), f"loss {float(loss)} vs golden {gl}"
    worst = 0.0
    for name, g in grads.items():
        s = z["gsum/" + name]
        idx = torch.from_numpy(z["gidx/" + name])
        vals = torch.from_numpy(z["gval/" + name])
        g = g.detach().double().cpu().flatten()
        gn = g.norm().item()
        assert abs(gn - s[1]) <= rtol_gnorm * max(s[1], 1e-12), f"{name}: |g|={gn} golden {s[1]}"
        ref_scale = max(vals.abs().max().item(), s[1] / np.sqrt(g.numel()), 1e-12)
        e = (g[idx].float() - vals).abs().max().item() / ref_scale
        worst = max(worst, e)
        assert e <= atol_gval_rel, f"{name}: sampled grad err {e:.3e} (normalised)"
    return worst
