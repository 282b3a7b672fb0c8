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
    return dict(name=name, z=z, kind=kind, L=L, B=B, N=N, train=bool(train), seed=seed, sd=sd, batch=batch, keep=keep)


def _group_name(name):
    """Every parameter is checked on its own.  (Round 1 checked attn.q_bias only fused with attn.v_bias: its gradient
    is a pure cancellation sum -- each row of the softmax Jacobian sums to zero -- and was off by 5-45 % alone.  The
    cause was delta = rowsum(dO * O) taken from the bf16-ROUNDED attention output, which breaks that zero row sum by
    dO.(O - bf16(O)); with the forward's rounding residual added back (attention `out_lo`) q_bias is within 1.5 % of its
    scale at 12 layers: profiles/r02_qbias_error.txt.)"""
    return name


def check_against_golden(case, loss, pred, grads, rtol_pred, rtol_loss, rtol_gnorm, atol_gval_rel):
    """grads: {name: tensor}.  Gradient check: L2 norm within rtol_gnorm, sampled elements within
    atol_gval_rel * max|sample| (a normalised max-abs test)."""
    z = case["z"]
    gp = torch.from_numpy(z["pred"])
    scale = gp.abs().max().item()
    err = (pred.detach().float().cpu() - gp).abs().max().item()
    err_pred = err / scale
    assert err <= rtol_pred * scale, f"pred max-abs err {err:.3e} vs scale {scale:.3e}"
    gl = float(z["loss"])
    err_loss = abs(float(loss) - gl) / abs(gl)
    assert abs(float(loss) - gl) <= rtol_loss * abs(gl), f"loss {float(loss)} vs golden {gl}"
    groups = {}
    for name, g in grads.items():
        idx = torch.from_numpy(z["gidx/" + name])
        vals = torch.from_numpy(z["gval/" + name])
        g = g.detach().double().cpu().flatten()
        e = groups.setdefault(_group_name(name), dict(n2=0.0, ref_n2=0.0, numel=0, got=[], ref=[]))
        e["n2"] += g.norm().item() ** 2
        e["ref_n2"] += float(z["gsum/" + name][1]) ** 2
        e["numel"] += g.numel()
        e["got"].append(g[idx].float())
        e["ref"].append(vals)
    worst = 0.0
    worst_norm = 0.0
    for name, e in groups.items():
        gn, rn = e["n2"] ** 0.5, e["ref_n2"] ** 0.5
        worst_norm = max(worst_norm, abs(gn - rn) / max(rn, 1e-12))
        assert abs(gn - rn) <= rtol_gnorm * max(rn, 1e-12), f"{name}: |g|={gn} golden {rn}"
        got, ref = torch.cat(e["got"]), torch.cat(e["ref"])
        ref_scale = max(ref.abs().max().item(), rn / np.sqrt(e["numel"]), 1e-12)
        err = (got - ref).abs().max().item() / ref_scale
        worst = max(worst, err)
        assert err <= atol_gval_rel, f"{name}: sampled grad err {err:.3e} (normalised)"
    print(f"MEASURED golden {case.get('name', '')}: pred err {err_pred:.2e} of max, loss err {err_loss:.2e}, worst gradient "
          f"norm dev {worst_norm:.2e}, worst sampled gradient err {worst:.2e}")
    return worst


def grouped(grads):
    """{group name: concatenated flat fp32 cpu tensor} with the q_bias/v_bias pairing of _group_name."""
    out = {}
    for name, g in grads.items():
        out.setdefault(_group_name(name), []).append(g.detach().float().cpu().flatten())
    return {k: torch.cat(v) for k, v in out.items()}
