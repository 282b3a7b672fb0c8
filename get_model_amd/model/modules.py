"""Mirror of get_model/model/modules.py for the region path: BaseModule, RegionEmbed, ExpressionHead.

Same class names, constructor (`cfg`) and parameter names (`embed.weight`, `head.weight`, ...), so reference
checkpoints load unchanged; the arithmetic runs in the HIP kernels (get_model_amd.functional).
"""
import torch
import torch.nn as nn
from torch.nn.init import trunc_normal_

from .. import functional as Fn


def dict_to_device(d, device):
    for key, value in d.items():
        if isinstance(value, torch.Tensor):
            d[key] = value.to(device)
    return d


class BaseModule(nn.Module):
    """reference: model/modules.py:41-78."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    def generate_dummy_data(self, batch_size=1):
        raise NotImplementedError("Dummy data generation must be implemented in subclasses.")

    def forward(self, x):
        raise NotImplementedError("Forward function must be implemented in subclasses.")

    def test(self, device="cuda"):
        x = self.generate_dummy_data()
        self.to(device)
        x = dict_to_device(x, device)
        return self(**x)

    def freeze_parameters(self):
        freezed = getattr(self.cfg, "freezed", False)
        if freezed:
            if freezed is True:
                for param in self.parameters():
                    param.requires_grad = False
            else:
                for name, param in self.named_parameters():
                    if freezed in name:
                        param.requires_grad = False


class RegionEmbed(BaseModule):
    """reference: model/modules.py:93-106 (`nn.Linear(num_features, embed_dim)` named `embed`).

    `forward(x)` returns the [B,N,D] fp32 embedding like the reference (what a forward hook on `region_embed`
    captures, run.py:359-384).  The tensor is a strided view: rows 1..N of a [B,N+1,D] buffer whose row 0 the CLS
    concat of the region models (functional.concat_cls) fills in place.  GETRegionPretrain also passes
    `mask_token=` / `mask=`: the blend of model/model.py:520-522 is then fused into the GEMM epilogue, so for that
    model the hooked output already holds the mask tokens (the one visible difference from the reference).
    """

    def __init__(self, cfg):
        super().__init__(cfg)
        self.embed = nn.Linear(cfg.num_features, cfg.embed_dim)

    def forward(self, x, mask_token=None, mask=None, **kwargs):
        return Fn.RegionEmbedTokens.apply(x, self.embed.weight, self.embed.bias, mask_token, mask)

    def generate_dummy_data(self, batch_size=1):
        return torch.rand(batch_size, 5, self.cfg.num_features)


class ExpressionHead(BaseModule):
    """reference: model/modules.py:239-258 (`nn.Linear(embed_dim, output_dim)` named `head`).

    `forward(tokens, softplus=True)` consumes the encoder's bf16 token tensor [B,N+1,D] (CLS first), drops the
    CLS row, applies the Linear and the Softplus of model/model.py:575-576 in one kernel.
    """

    def __init__(self, cfg):
        super().__init__(cfg)
        self.use_atac = getattr(cfg, "use_atac", False)
        if self.use_atac:
            raise NotImplementedError("head_exp.use_atac=True is not part of the region hot path "
                                      "(all shipped GETRegion* YAMLs set it to false)")
        self.head = nn.Linear(cfg.embed_dim, cfg.output_dim)
        trunc_normal_(self.head.weight, std=0.02)
        self.head.weight.data.mul_(0.001)
        self.head.bias.data.mul_(0.001)

    def forward(self, x, atac=None, softplus=True):
        if atac is not None:
            raise NotImplementedError("atac input to ExpressionHead is not supported on the region hot path")
        if not softplus:
            raise NotImplementedError("ExpressionHead without the fused Softplus is not implemented")
        return Fn.ExpressionHeadSoftplus.apply(x, self.head.weight, self.head.bias)
