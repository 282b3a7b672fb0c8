"""Generate tests/golden_extra/optim_groups_<kind>_L<L>.json from the REFERENCE's own optimizer code (container-only).

    python oracle/make_golden_optim.py

`/root/reference/get_model/optim.py` (get_num_layer_for_vit, LayerDecayValueAssigner, get_parameter_groups,
create_optimizer; lines 16-144) is imported unmodified.  Its one import that is not installed here is
`timm.optim.lookahead.Lookahead` (optim.py:11), a wrapper class the region configs never select (`opt: adamw`); it is
replaced by an in-memory placeholder class.  The reference models are built by oracle/ref_import.py and handed to the
reference's `create_optimizer` exactly as `LitModel.configure_optimizers` does (run.py:477-497): layer-decay values
0.75^(L+1-i), skip list = model.no_weight_decay().  Recorded per parameter: the group it landed in (index in the
optimizer's param_groups), that group's weight_decay and lr_scale, plus the optimizer class and its defaults.
Data only (names and numbers the reference produced); no reference source is stored.
"""
import importlib
import json
import os
import sys
import types
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_import import REFERENCE_ROOT, build_reference_model  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden_extra")


def load_reference_optim():
    build_reference_model("finetune", num_layers=1)          # installs the package shell + stand-ins
    if "timm.optim" not in sys.modules:
        topt = types.ModuleType("timm.optim")
        look = types.ModuleType("timm.optim.lookahead")

        class Lookahead:                                      # never constructed: the configs use plain "adamw"
            def __init__(self, *a, **k):
                raise RuntimeError("Lookahead placeholder: not used by the region configs")

        look.Lookahead = Lookahead
        topt.lookahead = look
        sys.modules["timm"].optim = topt
        sys.modules.update({"timm.optim": topt, "timm.optim.lookahead": look})
    assert REFERENCE_ROOT in sys.path
    return importlib.import_module("get_model.optim")


def case(ropt, kind, L, frozen=()):
    model = build_reference_model(kind, num_layers=L)
    if frozen:
        model.freeze_layers(patterns_to_freeze=list(frozen), invert_match=False)
    num_layers = model.cfg.encoder.num_layers                                     # run.py:478-481
    assigner = ropt.LayerDecayValueAssigner([0.75 ** (num_layers + 1 - i) for i in range(num_layers + 2)])
    args = SimpleNamespace(opt="adamw", lr=1e-4, weight_decay=0.05, opt_eps=1e-8, opt_betas=[0.9, 0.999])   # finetune_tutorial.yaml:29-35
    opt = ropt.create_optimizer(args, model, skip_list=model.no_weight_decay(),
                                get_num_layer=assigner.get_layer_id, get_layer_scale=assigner.get_scale)
    name_of = {id(p): n for n, p in model.named_parameters()}
    params = {}
    for gi, g in enumerate(opt.param_groups):
        for p in g["params"]:
            params[name_of[id(p)]] = [gi, float(g["weight_decay"]), float(g["lr_scale"]), float(g["lr"])]
    rec = {"kind": kind, "num_layers": L, "frozen": list(frozen), "optimizer_class": type(opt).__name__,
           "defaults": {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in opt.defaults.items()
                        if k in ("lr", "betas", "eps", "weight_decay")},
           "n_groups": len(opt.param_groups),
           "layer_ids": {n: int(ropt.get_num_layer_for_vit(n, num_layers + 2)) for n in params},
           "params": params}
    tag = f"optim_groups_{kind}_L{L}" + ("_frozen" if frozen else "")
    with open(os.path.join(OUT, tag + ".json"), "w") as f:
        json.dump(rec, f, indent=0, sort_keys=True)
    print(tag, len(params), "parameters in", len(opt.param_groups), "groups")


if __name__ == "__main__":
    ropt = load_reference_optim()
    case(ropt, "finetune", 12)
    case(ropt, "pretrain", 12)
    case(ropt, "atac", 3)
    case(ropt, "finetune", 2, frozen=("encoder.blocks.0", "region_embed"))
