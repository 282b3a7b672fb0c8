"""Generate the round-2 fixtures under tests/golden_extra/ from the REFERENCE itself (container-only).

    python oracle/make_golden_extra.py

  init_<kind>_L2_seed<k>.npz    SURVEY 8 a13: per-tensor checksums (sum, sum of squares, first / last element) of the
                                reference model constructed right after torch.manual_seed(k): `_init_weights`
                                (model/model.py:214-221) on top of the sub-modules' own initialisers.
  interpret_ft_L2_B2_N200.npz   the interpretation use of the path (run.py:351-434): forward hooks with retain_grad()
                                on `region_embed` and `encoder.blocks.1`, inputs with requires_grad, backward of one
                                strand of the prediction at a focus region; captured: hook output shapes, strided
                                samples and norms of the three Jacobians.
Data only (numbers produced by running the reference); no reference source is stored.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import get_cpu as O  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden_extra")
SAMPLES = 211


def sample(t):
    t = t.detach().double().flatten()
    idx = np.unique(np.linspace(0, t.numel() - 1, min(SAMPLES, t.numel())).astype(np.int64))
    return idx, t[idx].numpy().astype(np.float32), np.array([t.sum().item(), t.norm().item()])


def init_case(kind, L, seed):
    torch.manual_seed(seed)
    model = build_reference_model(kind, num_layers=L)
    data = {"meta": np.array([L, seed], dtype=np.int64), "kind": np.array(kind)}
    for name, p in model.state_dict().items():
        t = p.detach().double().flatten()
        data["chk/" + name] = np.array([t.sum().item(), (t * t).sum().item(), t[0].item(), t[-1].item()])
    np.savez_compressed(os.path.join(OUT, f"init_{kind}_L{L}_seed{seed}.npz"), **data)
    print("init", kind, L, seed, len(data) - 2, "tensors")


def interpret_case(L=2, B=2, N=200, seed=21, focus=57, strand=1):
    torch.manual_seed(0)
    model = build_reference_model("finetune", num_layers=L)
    model.load_state_dict(O.fill_state_dict("finetune", seed=seed, num_layers=L), strict=True)
    model.eval()
    batch = O.make_batch(B, N, seed)
    captured = {}

    def capture(name):                       # run.py:359-372
        def hook(module, inp, output):
            if isinstance(output, tuple):
                output = output[0]
            output.retain_grad()
            captured[name] = output
        return hook

    hooks = [model.get_submodule(n).register_forward_hook(capture(n)) for n in ("region_embed", "encoder.blocks.1")]
    inp = model.get_input(batch)
    for t in inp.values():
        t.requires_grad = True               # run.py:381-384
    out = model(**inp)
    pred, _ = model.before_loss(out, batch)
    m = torch.zeros_like(pred["exp"])
    m[:, focus, strand] = 1                  # run.py:399-404
    pred["exp"].backward(m)
    for h in hooks:
        h.remove()
    data = {"meta": np.array([L, B, N, seed, focus, strand], dtype=np.int64),
            "pred": pred["exp"].detach().numpy().astype(np.float32)}
    for name, t in (("input", inp["region_motif"]), ("region_embed", captured["region_embed"]),
                    ("encoder.blocks.1", captured["encoder.blocks.1"])):
        data["shape/" + name] = np.array(t.shape, dtype=np.int64)
        i, v, s = sample(t.grad)
        data["gidx/" + name], data["gval/" + name], data["gsum/" + name] = i, v, s
        if name != "input":
            i, v, s = sample(t)
            data["vidx/" + name], data["vval/" + name], data["vsum/" + name] = i, v, s
    # the focus row carries most of the input Jacobian: keep it whole
    data["input_focus_rows"] = inp["region_motif"].grad[:, focus].numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, f"interpret_ft_L{L}_B{B}_N{N}.npz"), **data)
    print("interpret:", {k: tuple(v) for k, v in data.items() if k.startswith("shape/")})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for kind in ("finetune", "pretrain", "atac"):
        init_case(kind, 2, 1234)
    init_case("finetune", 12, 7)
    interpret_case()
