"""Generate tests/golden/*.npz from the REFERENCE itself (container-only; needs /root/reference).

    python oracle/make_golden.py

Each fixture holds only data: the case parameters (so inputs/weights can be re-created with the
deterministic filler in oracle/get_cpu.py), the reference's prediction and loss, and per-parameter
gradient summaries (sum, L2 norm, and a strided sample of elements).  No reference source is stored.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import get_cpu as O  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = [
    # name, kind, layers, B, N, train(injected drop-path), seed
    ("ft_L2_B2_N200_eval", "finetune", 2, 2, 200, False, 1),
    ("ft_L3_B4_N200_train", "finetune", 3, 4, 200, True, 2),
    ("ft_L12_B4_N200_eval", "finetune", 12, 4, 200, False, 3),      # BASELINE.json configs[0] shape
    ("pre_L2_B2_N200_eval", "pretrain", 2, 2, 200, False, 4),
    ("pre_L3_B3_N64_train", "pretrain", 3, 3, 64, True, 5),
    ("atac_L2_B1_N2000_eval", "atac", 2, 1, 2000, False, 6),        # long-context shape (configs[3])
    ("atac_L2_B2_N37_eval", "atac", 2, 2, 37, False, 7),            # ragged small context
]

GRAD_SAMPLES = 97


def grad_summary(g):
    g = g.detach().double().flatten()
    n = g.numel()
    idx = np.unique(np.linspace(0, n - 1, min(GRAD_SAMPLES, n)).astype(np.int64))
    return np.array([g.sum().item(), g.norm().item()], dtype=np.float64), idx, g[idx].numpy().astype(np.float32)


def keep_masks_for(num_layers, B, seed):
    """Injected per-block (attn-branch, mlp-branch) keep masks; block 0 has rate 0 (Identity)."""
    out = []
    for i in range(num_layers):
        k1 = (O.hash_tensor((B,), 9000 + seed * 100 + 2 * i, 0.0, 1.0) < 0.7).float()
        k2 = (O.hash_tensor((B,), 9001 + seed * 100 + 2 * i, 0.0, 1.0) < 0.7).float()
        out.append((k1, k2))
    return out


def run_case(name, kind, L, B, N, train, seed):
    torch.manual_seed(0)
    model = build_reference_model(kind, num_layers=L)
    sd = O.fill_state_dict(kind, seed=seed, num_layers=L)
    missing, unexpected = model.load_state_dict(sd, strict=True)
    batch = O.make_batch(B, N, seed, kind=kind)
    model.train(train)
    if train:
        km = keep_masks_for(L, B, seed)
        for i, blk in enumerate(model.encoder.blocks):
            if hasattr(blk.drop_path, "injected"):
                blk.drop_path.injected = [km[i][0], km[i][1]]
    # the reference's training-step order: run.py:210-224
    inp = model.get_input(batch)
    out = model(**inp)
    pred, obs = model.before_loss(out, batch)
    loss = model.after_loss(model.loss(pred, obs))
    loss.backward()
    key = list(pred.keys())[0]
    data = {"pred": pred[key].detach().numpy().astype(np.float32),
            "loss": np.array(loss.item(), dtype=np.float64),
            "meta": np.array([L, B, N, int(train), seed], dtype=np.int64),
            "kind": np.array(kind)}
    for pname, p in model.named_parameters():
        s, idx, vals = grad_summary(p.grad)
        data["gsum/" + pname] = s
        data["gidx/" + pname] = idx
        data["gval/" + pname] = vals
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(f"{name}: loss={loss.item():.6f} pred{tuple(pred[key].shape)} mean={pred[key].mean().item():.5f}")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for c in CASES:
        run_case(*c)
