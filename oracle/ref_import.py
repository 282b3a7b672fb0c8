"""TEST INFRASTRUCTURE ONLY (container-only): import the reference's own region-model code.

Loads ``/root/reference/get_model/model/{model,transformer,modules}.py`` unmodified so that
golden vectors can be generated from the *reference's* arithmetic (oracle/make_golden.py) and
so that the CPU restatement in oracle/get_cpu.py can be checked against it.

The reference imports four third-party packages that are not installed in this image and cannot be
installed (no network): ``timm`` (transformer.py:7), ``torchmetrics`` (model.py:8),
``hydra.utils.instantiate`` (model.py:9) and ``omegaconf`` (model.py:10).  None of them does
arithmetic on the eval-mode hot path; the only one that touches numbers at all is timm's
``DropPath`` (train mode only).  They are replaced by the minimal in-memory modules below
(SURVEY.md section 8c).  Consequence, recorded in DESIGN.md: train-mode stochastic depth is pinned
"by definition" (per-sample Bernoulli keep, 1/keep_prob scaling); golden vectors use eval mode, or
train mode with an injected keep mask consumed identically by both sides.

Nothing here is shipped, and nothing under ``/root/reference`` exists on the GPU box: this module
is imported only by oracle/make_golden.py and by the tests that are skipped when the reference is
absent.
"""
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


class AttrDict(dict):
    """Attribute-access dict standing in for omegaconf.DictConfig (read-only use in the reference)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: to_attr(v) for k, v in d.items()})
    return d


class DropPath(nn.Module):
    """Stochastic depth as timm defines it: per-sample keep ~ Bernoulli(1-p), scaled by 1/(1-p).

    ``injected`` (a list of [B] 0/1 tensors, consumed front to back, one per call -- a Block calls
    its DropPath twice per forward, transformer.py:247-248) replaces the random draw so two
    implementations can consume the same masks.
    """

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep
        self.injected = None

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep_prob = 1.0 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        if self.injected is not None:
            mask = self.injected.pop(0).to(x.dtype).reshape(shape).clone()
        else:
            mask = x.new_empty(shape).bernoulli_(keep_prob)
        if keep_prob > 0.0 and self.scale_by_keep:
            mask = mask / keep_prob
        return x * mask


def _install_standins():
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        timm_models = types.ModuleType("timm.models")
        timm_layers = types.ModuleType("timm.models.layers")
        timm_layers.DropPath = DropPath
        timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
        timm.models = timm_models
        timm_models.layers = timm_layers
        sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.layers": timm_layers})
    if "torchmetrics" not in sys.modules:
        tm = types.ModuleType("torchmetrics")

        class _Noop(nn.Module):
            def forward(self, *a, **k):
                return torch.zeros(())

        for name in ("PearsonCorrCoef", "SpearmanCorrCoef", "MeanSquaredError", "R2Score"):
            setattr(tm, name, type(name, (_Noop,), {}))
        sys.modules["torchmetrics"] = tm
    if "hydra" not in sys.modules:
        hydra = types.ModuleType("hydra")
        hydra_utils = types.ModuleType("hydra.utils")

        def instantiate(cfg, *a, **k):
            raise RuntimeError("hydra.utils.instantiate stand-in: not used on the region path")

        hydra_utils.instantiate = instantiate
        hydra.utils = hydra_utils
        sys.modules.update({"hydra": hydra, "hydra.utils": hydra_utils})
    if "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")
        oc.MISSING = "???"
        oc.DictConfig = AttrDict
        sys.modules["omegaconf"] = oc


def load_reference():
    """Returns the reference's ``get_model.model.model`` module (imported from /root/reference)."""
    _install_standins()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # get_model/__init__.py pulls in the Lightning harness; bypass it with an empty package shell so
    # only the model files execute.
    if "get_model" not in sys.modules:
        pkg = types.ModuleType("get_model")
        pkg.__path__ = [REFERENCE_ROOT + "/get_model"]
        sys.modules["get_model"] = pkg
        sub = types.ModuleType("get_model.model")
        sub.__path__ = [REFERENCE_ROOT + "/get_model/model"]
        sys.modules["get_model.model"] = sub
    import importlib

    return importlib.import_module("get_model.model.model")


def region_cfg(kind, num_layers=12, embed_dim=768, num_heads=12, num_motif=283, dropout=0.1,
               output_dim=None):
    """The ``model.cfg`` tree of config/model/GETRegion{Finetune,Pretrain,FinetuneATAC}.yaml with
    interpolations resolved and the Hydra-instantiated loss module in place."""
    if output_dim is None:
        output_dim = {"finetune": 2, "atac": 1, "pretrain": num_motif}[kind]
    cfg = {
        "num_regions": 900, "num_motif": num_motif, "embed_dim": embed_dim, "num_layers": num_layers,
        "num_heads": num_heads, "dropout": dropout, "output_dim": output_dim, "flash_attn": False,
        "pool_method": "mean",
        "region_embed": {"num_features": num_motif, "embed_dim": embed_dim},
        "encoder": {"num_heads": num_heads, "embed_dim": embed_dim, "num_layers": num_layers,
                    "drop_path_rate": dropout, "drop_rate": 0, "attn_drop_rate": 0,
                    "use_mean_pooling": False, "flash_attn": False},
        "mask_token": {"embed_dim": embed_dim, "std": 0.02},
    }
    if kind == "pretrain":
        cfg["head_mask"] = {"in_features": embed_dim, "out_features": output_dim}
        cfg["loss"] = {"components": {"masked": nn.MSELoss(reduction="mean")}, "weights": {"masked": 1.0}}
        cfg["metrics"] = {"components": {"masked": ["pearson", "mse", "r2"]}}
    else:
        name = "exp" if kind == "finetune" else "atpm"
        cfg["head_exp"] = {"embed_dim": embed_dim, "output_dim": output_dim, "use_atac": False}
        cfg["loss"] = {"components": {name: nn.PoissonNLLLoss(log_input=False, reduction="mean")},
                       "weights": {name: 1.0}}
        cfg["metrics"] = {"components": {name: ["pearson", "spearman", "r2"]}}
    return to_attr(cfg)


def build_reference_model(kind, **kw):
    ref = load_reference()
    cls = {"finetune": ref.GETRegionFinetune, "pretrain": ref.GETRegionPretrain,
           "atac": ref.GETRegionFinetuneATAC}[kind]
    return cls(region_cfg(kind, **kw))
