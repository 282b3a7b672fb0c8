"""Mirror of get_model/model/model.py for the region models (drop-in: same class names, `cfg` constructor,
forward signatures, get_input / before_loss / loss / after_loss protocol, state_dict keys).

Only the `_target_` strings of config/model/GETRegion{Finetune,FinetuneATAC,Pretrain}.yaml change:
    get_model.model.model.GETRegionFinetune  ->  get_model_amd.model.model.GETRegionFinetune
"""
import logging
from collections.abc import Mapping

import torch
import torch.nn as nn
from torch.nn.init import trunc_normal_

from .. import functional as Fn
from .. import kernels as K
from .modules import BaseModule, ExpressionHead, RegionEmbed
from .transformer import GETTransformer


def _get(cfg, key, default=None):
    if isinstance(cfg, Mapping):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class GETLoss(nn.Module):
    """reference: model/model.py:125-157.  `cfg.components[name]` arrives as an instantiated nn.Module (Hydra
    builds `_target_: torch.nn.PoissonNLLLoss` / `torch.nn.MSELoss` recursively); those two run as the fused
    HIP loss, any other module is called as is."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        comps, weights = _get(cfg, "components"), _get(cfg, "weights")
        self.losses = {name: (component, weights[f"{name}"]) for name, component in comps.items()}

    @staticmethod
    def _apply_one(loss_fn, weight, pred, obs):
        if pred.is_cuda:
            if (isinstance(loss_fn, nn.PoissonNLLLoss) and not loss_fn.log_input and not loss_fn.full
                    and loss_fn.reduction == "mean"):
                return Fn.FusedLoss.apply(pred, obs, K.LOSS_POISSON, float(weight), float(loss_fn.eps))
            if isinstance(loss_fn, nn.MSELoss) and loss_fn.reduction == "mean":
                return Fn.FusedLoss.apply(pred, obs, K.LOSS_MSE, float(weight), 0.0)
        return loss_fn(pred, obs) * weight

    def forward(self, pred, obs):
        return {f"{name}_loss": self._apply_one(loss_fn, weight, pred[name], obs[name])
                for name, (loss_fn, weight) in self.losses.items()}

    def freeze_component(self, component_name):
        if component_name in self.losses:
            self.losses[component_name] = (self.losses[component_name][0], 0)
        else:
            raise ValueError(f"Component '{component_name}' not found in the loss function.")


class RegressionMetrics(nn.Module):
    """reference: model/model.py:160-197 (torchmetrics Pearson/Spearman/MSE/R2 over flattened tensors).
    Evaluation-only and off the hot path; restated with plain torch ops so `model.metrics` exists."""

    def __init__(self, _cfg_):
        super().__init__()
        self.cfg = _cfg_
        self.components = {t: list(names) for t, names in _get(_cfg_, "components").items()}

    @staticmethod
    def _rank(x):
        """Ranks 1..n with TIES AVERAGED (torchmetrics SpearmanCorrCoef, model/model.py:173-174, ranks tied values
        by the mean of their positions; expression labels are mostly tied zeros)."""
        n = x.numel()
        order = torch.argsort(x, stable=True)
        xs = x[order]
        pos = torch.arange(1, n + 1, device=x.device, dtype=torch.float64)
        new_run = torch.ones(n, dtype=torch.bool, device=x.device)
        if n > 1:
            new_run[1:] = xs[1:] != xs[:-1]
        run = torch.cumsum(new_run.to(torch.int64), 0) - 1                  # run index of every sorted element
        nrun = int(run[-1].item()) + 1 if n else 0
        tot = torch.zeros(nrun, dtype=torch.float64, device=x.device).index_add_(0, run, pos)
        cnt = torch.zeros(nrun, dtype=torch.float64, device=x.device).index_add_(0, run, torch.ones_like(pos))
        ranks = torch.empty(n, dtype=torch.float64, device=x.device)
        ranks[order] = (tot / cnt)[run]
        return ranks.to(torch.float32)

    @staticmethod
    def _pearson(a, b):
        a = a - a.mean()
        b = b - b.mean()
        return (a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-12)

    def _metric(self, name, p, o):
        p, o = p.reshape(-1).float(), o.reshape(-1).float()
        if name == "pearson":
            return self._pearson(p, o)
        if name == "spearman":
            return self._pearson(self._rank(p), self._rank(o))
        if name == "mse":
            return ((p - o) ** 2).mean()
        if name == "r2":
            return 1 - ((o - p) ** 2).sum() / ((o - o.mean()) ** 2).sum().clamp_min(1e-12)
        raise ValueError(f"Unsupported metric: {name}")

    def forward(self, _pred_, _obs_):
        return {f"{t}_{m}": self._metric(m, _pred_[t], _obs_[t]) for t, names in self.components.items() for m in names}


class BaseGETModel(BaseModule):
    """reference: model/model.py:207-290."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.cfg = cfg
        self.loss = GETLoss(cfg.loss)
        self.metrics = RegressionMetrics(cfg.metrics)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_input(self, batch):
        raise NotImplementedError

    def forward(self, batch):
        raise NotImplementedError

    def before_loss(self, output, batch):
        raise NotImplementedError

    def after_loss(self, loss):
        if isinstance(loss, dict):
            return sum(loss.values())
        return loss

    def freeze_layers(self, patterns_to_freeze=None, invert_match=False):
        if patterns_to_freeze is None:
            patterns_to_freeze = []
        for name, param in self.named_parameters():
            matches_pattern = any(pattern in name for pattern in patterns_to_freeze)
            should_freeze = matches_pattern if not invert_match else not matches_pattern
            if should_freeze:
                param.requires_grad = False
                logging.debug(f"Freezed weights of {name}")

    def generate_dummy_data(self):
        raise NotImplementedError

    def get_layer(self, layer_name):
        if hasattr(self, layer_name):
            return getattr(self, layer_name)
        raise ValueError(f"Layer '{layer_name}' not found in the model.")

    def get_layer_names(self):
        return list(self._modules.keys())

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}


class GETRegionPretrain(BaseGETModel):
    """reference: model/model.py:499-542."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.region_embed = RegionEmbed(cfg.region_embed)
        self.encoder = GETTransformer(**cfg.encoder)
        self.head_mask = nn.Linear(**cfg.head_mask)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, cfg.mask_token.embed_dim))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, cfg.embed_dim))
        trunc_normal_(self.mask_token, std=cfg.mask_token.std)
        self.apply(self._init_weights)

    def get_input(self, batch):
        return {"region_motif": batch["region_motif"], "mask": batch["mask"].unsqueeze(-1).bool()}

    def forward(self, region_motif, mask):
        rows = Fn.MaskedRows(mask)              # device-side row lists of x[:, 1:][mask]; see functional.MaskedRows
        x = self.region_embed(region_motif, mask_token=self.mask_token, mask=mask)      # embed + blend (:520-522)
        x = Fn.concat_cls(x, self.cls_token)                                             # :523-525
        x, _ = self.encoder(x)
        x_masked = Fn.MaskedHead.apply(x, self.head_mask.weight, self.head_mask.bias, rows)   # :527-528
        x_masked._getmi_rows = rows
        return x_masked, region_motif, mask

    def before_loss(self, output, batch):
        x_masked, x_original, loss_mask = output
        B, _, C = x_original.shape
        pred = {"masked": x_masked}
        rows = getattr(x_masked, "_getmi_rows", None)
        if rows is not None and rows.mask is loss_mask and x_original.is_cuda:
            # same rows, same order as x_original[loss_mask.squeeze(-1)] (:535), from the row list forward already built
            target = x_original.reshape(-1, C).index_select(0, rows.region_rows[:rows.n])
        else:
            target = x_original[loss_mask.squeeze(-1)].reshape(-1, C)
        obs = {"masked": target}
        return pred, obs

    def generate_dummy_data(self):
        B, R, M = 2, 900, 283
        return {"region_motif": torch.randn(B, R, M).float().abs(), "mask": torch.rand(B, R, 1) < 0.5}


class GETRegionFinetune(BaseGETModel):
    """reference: model/model.py:554-589."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.region_embed = RegionEmbed(cfg.region_embed)
        self.encoder = GETTransformer(**cfg.encoder)
        self.head_exp = ExpressionHead(cfg.head_exp)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, cfg.embed_dim))
        self.apply(self._init_weights)

    def get_input(self, batch, perturb=False):
        return {"region_motif": batch["region_motif"]}

    def forward(self, region_motif):
        x = self.region_embed(region_motif)
        x = Fn.concat_cls(x, self.cls_token)            # model/model.py:571-573
        x, _ = self.encoder(x)
        return self.head_exp(x, fused_softplus=True)          # drops CLS, Linear, Softplus (one kernel)

    def before_loss(self, output, batch):
        return {"exp": output}, {"exp": batch["exp_label"]}

    def generate_dummy_data(self):
        B, R, M = 2, 900, 283
        return {"region_motif": torch.randn(B, R, M).float().abs()}


class GETRegionFinetuneATAC(BaseGETModel):
    """reference: model/model.py:661-698."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.region_embed = RegionEmbed(cfg.region_embed)
        self.encoder = GETTransformer(**cfg.encoder)
        self.head_exp = ExpressionHead(cfg.head_exp)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, cfg.embed_dim))
        self.apply(self._init_weights)

    def get_input(self, batch):
        inp = batch["region_motif"].clone()
        inp[:, :, -1] = 1
        return {"region_motif": inp}

    def forward(self, region_motif):
        x = self.region_embed(region_motif)
        x = Fn.concat_cls(x, self.cls_token)            # model/model.py:571-573
        x, _ = self.encoder(x)
        return self.head_exp(x, fused_softplus=True)

    def before_loss(self, output, batch):
        return {"atpm": output}, {"atpm": batch["region_motif"][:, :, -1].unsqueeze(-1)}

    def generate_dummy_data(self):
        B, R, M = 2, 900, 283
        return {"region_motif": torch.randn(B, R, M).float().abs()}
