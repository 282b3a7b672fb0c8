"""Optimizer and learning-rate schedule of the GET training step.

Host-side mirror of the reference's `get_model/optim.py` (parameter groups, layer ids, `create_optimizer`) and
`utils.cosine_scheduler` (utils.py:314-345), with the same names and argument meaning, plus `FusedAdamW`: one HIP
launch (`getmi_adamw_step`) updates every parameter and writes the bf16 GEMM operand copies of the weights, where
the reference runs `torch.optim.AdamW` and lets autocast re-cast the weights in every forward.
"""
import ctypes
import json
import logging
import math

import numpy as np
import torch

from . import functional as Fn
from . import lib


# ---- layer ids / parameter groups (optim.py:16-84) ---------------------------------------------------------
def get_num_layer_for_vit(var_name, num_max_layer):
    """Layer id used for layer-wise grouping: embeddings/tokens 0, encoder block i -> i+1, the rest last."""
    if var_name in ("cls_token", "mask_token", "pos_embed") or var_name.startswith("region_embed"):
        return 0
    if var_name.startswith("rel_pos_bias"):
        return num_max_layer - 1
    parts = var_name.split(".")
    if var_name.startswith("blocks"):
        return int(parts[1]) + 1
    if var_name.startswith("encoder.blocks"):
        return int(parts[2]) + 1
    return num_max_layer - 1


class LayerDecayValueAssigner(object):
    def __init__(self, values):
        self.values = values

    def get_scale(self, layer_id):
        return self.values[layer_id]

    def get_layer_id(self, var_name):
        return get_num_layer_for_vit(var_name, len(self.values))


def get_parameter_groups(model, weight_decay=1e-5, skip_list=(), get_num_layer=None, get_layer_scale=None):
    """1-D parameters, `.bias` and names in skip_list get weight_decay 0; groups are keyed by (layer id, decay).
    `lr_scale` is stored in every group exactly as the reference does -- and, as there, nothing ever applies it
    (run.py:517-528 scales all groups by the same LambdaLR factor)."""
    names, groups = {}, {}
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        no_decay = param.ndim == 1 or name.endswith(".bias") or name in skip_list
        key = "no_decay" if no_decay else "decay"
        layer_id = None
        if get_num_layer is not None:
            layer_id = get_num_layer(name)
            key = "layer_%d_%s" % (layer_id, key)
        if key not in groups:
            scale = get_layer_scale(layer_id) if get_layer_scale is not None else 1.0
            groups[key] = {"weight_decay": 0.0 if no_decay else weight_decay, "params": [], "lr_scale": scale}
            names[key] = {"weight_decay": groups[key]["weight_decay"], "params": [], "lr_scale": scale}
        groups[key]["params"].append(param)
        names[key]["params"].append(name)
    logging.debug("Param groups = %s" % json.dumps(names, indent=2))
    return list(groups.values())


def create_optimizer(args, model, get_num_layer=None, get_layer_scale=None, filter_bias_and_bn=True, skip_list=None):
    """args: object with .opt, .lr, .weight_decay and optionally .opt_eps, .opt_betas, .momentum
    (config/finetune_tutorial.yaml:29-35).  "adamw" -> FusedAdamW on GPU parameters."""
    opt_lower = args.opt.lower()
    weight_decay = args.weight_decay
    if weight_decay and filter_bias_and_bn:
        if skip_list is not None:
            skip = skip_list
        elif hasattr(model, "no_weight_decay"):
            skip = model.no_weight_decay()
        else:
            skip = {}
        parameters = get_parameter_groups(model, weight_decay, skip, get_num_layer, get_layer_scale)
        weight_decay = 0.0
    else:
        parameters = list(model.parameters())
    opt_args = dict(lr=args.lr, weight_decay=weight_decay)
    if getattr(args, "opt_eps", None) is not None:
        opt_args["eps"] = args.opt_eps
    if getattr(args, "opt_betas", None) is not None:
        opt_args["betas"] = tuple(args.opt_betas)
    opt_split = opt_lower.split("_")
    if len(opt_split) > 1:
        raise NotImplementedError("lookahead wrappers (timm) are not part of the region training configs")
    kind = opt_split[-1]
    if kind in ("sgd", "nesterov", "momentum"):
        opt_args.pop("eps", None)
        return torch.optim.SGD(parameters, momentum=args.momentum, nesterov=kind != "momentum", **opt_args)
    if kind == "adam":
        return torch.optim.Adam(parameters, **opt_args)
    if kind == "adamw":
        return FusedAdamW(parameters, **opt_args)
    raise ValueError("Invalid optimizer %r" % args.opt)


def configure_optimizer(model, optimizer_cfg):
    """What LitModel.configure_optimizers does before the schedule (run.py:477-497): layer-wise groups with the
    0.75^(L+1-i) values (stored, inert), weight-decay skip list from the model."""
    num_layers = model.cfg.encoder.num_layers if hasattr(model.cfg, "encoder") else 0
    assigner = LayerDecayValueAssigner([0.75 ** (num_layers + 1 - i) for i in range(num_layers + 2)])
    return create_optimizer(optimizer_cfg, model, skip_list=model.no_weight_decay(),
                            get_num_layer=assigner.get_layer_id, get_layer_scale=assigner.get_scale)


# ---- schedule (utils.py:314-345, run.py:505-528) -------------------------------------------------------------
def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0,
                     warmup_steps=-1):
    """Per-step values: linear warm-up start_warmup_value -> base_value, then a half cosine to final_value."""
    warmup_iters = warmup_epochs * niter_per_ep
    if warmup_steps > 0:
        warmup_iters = warmup_steps
    warmup = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.array([])
    n = epochs * niter_per_ep - warmup_iters
    i = np.arange(n)
    cosine = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * i / n)) if n > 0 else np.array([])
    schedule = np.concatenate((warmup, cosine))
    assert len(schedule) == epochs * niter_per_ep
    return schedule


def make_lr_scheduler(optimizer, schedule, base_lr):
    """LambdaLR stepped once per batch; the factor for step s is schedule[s-1]/base_lr -- including the
    reference's index quirk that step 0 reads the LAST schedule entry (run.py:517)."""
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda step: schedule[step - 1] / base_lr)


# ---- fused AdamW -----------------------------------------------------------------------------------------------
class _Group(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("lr", "weight_decay", "beta1", "beta2", "eps", "step_size",
                                              "inv_sqrt_bc2", "reserved")]


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, no amsgrad) in one launch.

    State per parameter uses torch's names (`step`, `exp_avg`, `exp_avg_sq`), so optimizer state dicts are
    interchangeable with torch.optim.AdamW.  For parameters the fused model uses as GEMM weights the kernel also
    rewrites their cached bf16 copies (functional.WEIGHTS) in place, so no cast kernels run in the next step."""

    getmi_refreshes_weight_cache = True   # functional._after_optimizer_step leaves the cache epoch to step()
    CHUNK = 4096

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._sig = None
        self._jobs = None
        self._nflat = self._ntile = 0

    def _state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._sig = None            # the cached job table points at the moment tensors that were just replaced

    def _build(self, entries):
        """entries: (p, grad, m, v, wn, wt, group index) -> device job table."""
        flat, tile = [], []
        for p, g, m, v, wn, wt, gi in entries:
            base = [p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), wn.data_ptr() if wn is not None else 0,
                    wt.data_ptr() if wt is not None else 0]
            n = p.numel()
            if wt is not None and p.ndim == 2 and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0:
                rows, cols = p.shape
                r0, c0 = np.meshgrid(np.arange(0, rows, 64), np.arange(0, cols, 64), indexing="ij")
                a = r0.ravel().astype(np.int64) | (c0.ravel().astype(np.int64) << 32)
                b = np.full_like(a, cols | (rows << 20) | (gi << 40))
                tile.append(np.concatenate([np.tile(np.array(base, dtype=np.int64), (a.size, 1)), a[:, None], b[:, None]], 1))
            else:
                start = np.arange(0, n, self.CHUNK, dtype=np.int64)
                cnt = np.minimum(self.CHUNK, n - start)
                rec = np.tile(np.array(base, dtype=np.int64), (start.size, 1))
                rec[:, 5] = 0
                flat.append(np.concatenate([rec, start[:, None], (cnt | (gi << 40))[:, None]], 1))
        parts = flat + tile
        self._nflat = int(sum(x.shape[0] for x in flat))
        self._ntile = int(sum(x.shape[0] for x in tile))
        table = np.ascontiguousarray(np.concatenate(parts, 0)) if parts else np.zeros((0, 8), np.int64)
        self._jobs = torch.from_numpy(table).to(entries[0][0].device) if parts else None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        cache = Fn.WEIGHTS
        entries, sig = [], []
        table = {}      # (param group, step count) -> row of the hyper-parameter table: bias correction is per parameter
        retag = []      # in torch.optim.AdamW, so parameters of one group whose step counts differ get rows of their own
        stepped = []
        # pass 1 validates and collects WITHOUT touching any state: a raise below leaves every step count (and with it
        # the bias correction of a retried step) as it was
        for gi, grp in enumerate(self.param_groups):
            for p in grp["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdamW runs on the GPU only (no CPU fallback)")
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedAdamW needs dense contiguous fp32 parameters and gradients")
                st = self.state[p]
                t_next = (float(st["step"]) if len(st) else 0.0) + 1.0
                table.setdefault((gi, t_next), len(table))
                stepped.append((gi, p, t_next))
        if len(table) > 64:
            raise RuntimeError("FusedAdamW: more than 64 (parameter group, step count) combinations")
        cache.epoch += 1            # every cached copy is stale from here on, except the ones re-tagged below
        for gi, p, t_next in stepped:
            st = self._state(p)
            st["step"] += 1
            ti = table[(gi, t_next)]
            wn = cache.peek(p, "n")
            wt = cache.peek(p, "t")
            if wt is not None and not (p.ndim == 2 and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0):
                wt = None
            entries.append((p, p.grad, st["exp_avg"], st["exp_avg_sq"], wn, wt, ti))
            # every pointer the job table holds is part of the signature: load_state_dict() replaces the moment
            # tensors while parameter and gradient addresses stay put
            sig.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                        0 if wn is None else wn.data_ptr(), 0 if wt is None else wt.data_ptr(), ti))
            if wn is not None:
                retag.append((p, "n"))
            if wt is not None:
                retag.append((p, "t"))
        groups = (_Group * max(len(table), 1))()
        for (gi, t), ti in table.items():
            grp = self.param_groups[gi]
            b1, b2 = grp["betas"]
            g = groups[ti]
            g.lr, g.weight_decay, g.beta1, g.beta2, g.eps = grp["lr"], grp["weight_decay"], b1, b2, grp["eps"]
            g.step_size = grp["lr"] / (1.0 - b1 ** t)
            g.inv_sqrt_bc2 = 1.0 / math.sqrt(1.0 - b2 ** t)
        if not entries:
            return loss
        sig = tuple(sig)
        if sig != self._sig:       # pointers are stable with ddp.GradBuckets; rebuilt only when something moved
            self._build(entries)
            self._sig = sig
        lib.call("getmi_adamw_step", self._jobs, self._nflat, self._ntile, ctypes.addressof(groups), len(table))
        for p, kind in retag:
            cache.retag(p, kind)
        return loss
