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

    `forward(x)` keeps the reference's contract: x[B,N,D] (any float dtype, on the GPU) -> logits [B,N,output_dim],
    no CLS handling, no Softplus (functional.HeadLinear; the same GEMM kernels, off the training path).
    The region models call `forward(tokens, fused_softplus=True)` instead: `tokens` is the encoder's bf16 tensor
    [B,N+1,D] (CLS first); one kernel drops the CLS row, applies the Linear and the Softplus of model/model.py:575-576
    and returns [B,N,output_dim].  That form refuses anything but the encoder's own output, so a caller who slices the
    CLS row off first (as model/model.py:575 does) gets an error instead of silently losing region 0.
    (A forward hook on `head_exp` therefore sees softplus'ed predictions where the reference's sees logits.)
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

    def forward(self, x, atac=None, fused_softplus=False):
        if atac is not None:
            raise NotImplementedError("atac input to ExpressionHead is not supported on the region hot path")
        if not fused_softplus:
            return Fn.HeadLinear.apply(x, self.head.weight, self.head.bias)
        if not getattr(x, "_getmi_encoder_tokens", False) or x.dtype != torch.bfloat16 or x.dim() != 3:
            raise RuntimeError("ExpressionHead(fused_softplus=True) takes the encoder's own output (bf16 [B,N+1,D], CLS "
                               "row included); for any other tensor call head(x) -- the reference's [B,N,D] -> logits "
                               "contract -- and apply Softplus yourself")
        return Fn.ExpressionHeadSoftplus.apply(x, self.head.weight, self.head.bias)
