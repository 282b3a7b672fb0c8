"""Mirror of get_model/model/transformer.py for the region path: Attention, Mlp, Block, GETTransformer.

The submodule tree and parameter names equal the reference's (state_dict compatible):
  blocks.i.norm1.{weight,bias}, blocks.i.attn.{q_bias,v_bias,qkv.weight,proj.weight,proj.bias},
  blocks.i.norm2.{weight,bias}, blocks.i.mlp.{fc1,fc2}.{weight,bias}, norm.{weight,bias}.
`Block.forward` runs as ONE fused autograd operator (functional.EncoderBlock); `Attention` and `Mlp` are
parameter containers (their own forward is not on the hot path and is not implemented).
"""
from functools import partial

import torch
import torch.nn as nn

from .. import functional as Fn


class DropPath(nn.Module):
    """Stochastic depth per sample (timm semantics, call sites model/transformer.py:220,247-248):
    keep ~ Bernoulli(1-p) per sample, kept samples scaled by 1/(1-p).  Returns the per-sample FACTOR [B]
    (the scaling itself is fused into the proj / fc2 GEMM epilogues).  `injected`: optional list of [B] 0/1
    tensors consumed front to back (parity tests feed the oracle's masks)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.injected = None

    def factor(self, batch, device):
        if self.drop_prob == 0.0 or not self.training:
            return None
        keep_prob = 1.0 - self.drop_prob
        if self.injected:
            keep = self.injected.pop(0).to(device=device, dtype=torch.float32)
        else:
            keep = torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep_prob)
        return (keep / keep_prob).contiguous()

    def extra_repr(self):
        return f"drop_prob={self.drop_prob:.3f}"


def _refuse_forward_hook(self, *args, **kwargs):
    """`Attention` / `Mlp` (and the `nn.Linear`s inside them) are parameter containers here: their forward never runs,
    so a forward hook on them -- any module name is legal in the reference's interpretation config, run.py:359-384 --
    would silently capture nothing.  Refused with a pointer to the modules whose outputs do exist."""
    raise RuntimeError(f"get_model_amd: a forward hook on {type(self).__name__} would never fire -- attention and MLP "
                       "run fused inside Block.forward (functional.EncoderBlock).  Hook `encoder.blocks.<i>` (the "
                       "block's output), `region_embed` or `encoder` instead.")


class _FusedLinear(nn.Linear):
    """nn.Linear whose arithmetic runs inside the fused block kernels: a genuine nn.Linear for state_dict / LoRA
    parametrisations (run_region.py:319-331), but forward hooks are refused (they would never fire)."""

    register_forward_hook = _refuse_forward_hook
    register_forward_pre_hook = _refuse_forward_hook


class Attention(nn.Module):
    """Parameters of model/transformer.py:15-44 (packed qkv without bias + separate q_bias / v_bias)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0,
                 attn_head_dim=None):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads if attn_head_dim is None else attn_head_dim
        all_head_dim = head_dim * num_heads
        self.scale = qk_scale or head_dim ** -0.5
        if head_dim != 64 or qk_scale is not None:
            raise NotImplementedError("the MI355X attention kernel is built for head_dim 64 with the default scale")
        if attn_drop or proj_drop:
            raise NotImplementedError("attention / projection dropout > 0 is not on the region hot path "
                                      "(drop_rate and attn_drop_rate are 0 in every GETRegion* YAML)")
        self.qkv = _FusedLinear(dim, all_head_dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(all_head_dim))
            self.v_bias = nn.Parameter(torch.zeros(all_head_dim))
        else:
            raise NotImplementedError("qkv_bias=False is not used by GETTransformer (default True)")
        self.proj = _FusedLinear(all_head_dim, dim)

    def forward(self, x, attention_mask=None, attention_bias=None):
        raise NotImplementedError("Attention runs fused inside Block.forward (functional.EncoderBlock)")

    register_forward_hook = _refuse_forward_hook
    register_forward_pre_hook = _refuse_forward_hook


class Mlp(nn.Module):
    """Parameters of model/transformer.py:153-169 (fc1 -> erf-GELU -> fc2)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if act_layer is not nn.GELU or drop:
            raise NotImplementedError("the fused MLP kernel implements exact GELU without dropout")
        self.fc1 = _FusedLinear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = _FusedLinear(hidden_features, out_features)

    def forward(self, x):
        raise NotImplementedError("Mlp runs fused inside Block.forward (functional.EncoderBlock)")

    register_forward_hook = _refuse_forward_hook
    register_forward_pre_hook = _refuse_forward_hook


class Block(nn.Module):
    """model/transformer.py:180-257, gamma_1-is-None branch (init_values = 0 in GETTransformer)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0, attn_drop=0,
                 drop_path=0.1, init_values=0.001, act_layer=nn.GELU, norm_layer=nn.LayerNorm, attn_head_dim=None,
                 flash_attn=False):
        super().__init__()
        if init_values and init_values > 0:
            raise NotImplementedError("LayerScale (init_values > 0) is not used by the region models")
        self.norm1 = norm_layer(dim)
        # `flash_attn` selects Attention_Flash in the reference (broken there); here attention is
        # always the fused kernel, so the flag is accepted and ignored.
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop, attn_head_dim=attn_head_dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.gamma_1, self.gamma_2 = None, None

    def forward(self, x, attention_mask=None, attention_bias=None, drop_path_factors=None, upstream=None):
        """upstream: the Block whose output x is (set by GETTransformer.forward; lets this block's backward hand that
        block the bf16 gradient of its MLP branch, functional._GradHandoff)."""
        if attention_mask is not None or attention_bias is not None:
            raise NotImplementedError("attention mask / bias are not taken on the region path "
                                      "(model/model.py:574 calls the encoder without them)")
        s1 = s2 = None
        if drop_path_factors is not None:      # drawn for all blocks at once by GETTransformer.forward
            s1, s2 = drop_path_factors
        elif isinstance(self.drop_path, DropPath):
            s1 = self.drop_path.factor(x.shape[0], x.device)
            s2 = self.drop_path.factor(x.shape[0], x.device)
        a, m = self.attn, self.mlp
        self.last_scale2 = s2
        x = Fn.EncoderBlock.apply(x, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.q_bias, a.v_bias,
                                  a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight,
                                  m.fc1.bias, m.fc2.weight, m.fc2.bias, s1, s2, a.num_heads, self.norm1.eps,
                                  upstream is not None, None if upstream is None else upstream.last_scale2)
        return x, None


class GETTransformer(nn.Module):
    """model/transformer.py:301-367.  Returns (tokens, attn_output); `tokens` are the final-LayerNorm outputs
    in bf16 (the dtype the heads consume)."""

    def __init__(self, embed_dim, num_heads=8, num_layers=8, mlp_ratio=4, qkv_bias=True, qk_scale=None, drop_rate=0,
                 attn_drop_rate=0, drop_path_rate=0.1, norm_layer=partial(nn.LayerNorm, eps=1e-6), init_values=0,
                 use_mean_pooling=False, flash_attn=False, *args, **kwargs):
        super().__init__()
        if use_mean_pooling:
            raise NotImplementedError("use_mean_pooling=True is not used by the region models")
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, num_layers)]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer,
                  init_values=init_values, flash_attn=flash_attn) for i in range(num_layers)])
        self.norm = norm_layer(embed_dim)
        self.fc_norm = None
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _draw_drop_path(self, batch, device):
        """All stochastic-depth factors of one forward from ONE uniform draw ([L, 2, B]; 3 launches instead of two
        bernoulli + divide per block).  None when not training, when nothing drops, or when a test injected masks
        (then every block draws / pops its own, DropPath.factor)."""
        if not self.training:
            return None
        dps = [blk.drop_path for blk in self.blocks]
        if any(isinstance(d, DropPath) and d.injected for d in dps):
            return None
        probs = [d.drop_prob if isinstance(d, DropPath) else 0.0 for d in dps]
        if max(probs) == 0.0:
            return None
        keep = torch.tensor([1.0 - p for p in probs], dtype=torch.float32, device=device).view(-1, 1, 1)
        f = (torch.rand(len(dps), 2, batch, device=device) < keep).to(torch.float32) / keep
        return [None if p == 0.0 else (f[i, 0], f[i, 1]) for i, p in enumerate(probs)]

    def forward(self, x, mask=None, return_attns=False, bias=None):
        if return_attns:
            raise NotImplementedError("return_attns=True would materialise the [B,H,T,T] probabilities the fused "
                                      "attention kernel never forms")
        factors = self._draw_drop_path(x.shape[0], x.device)
        prev = None
        for i, blk in enumerate(self.blocks):
            x, _ = blk(x, mask, bias, drop_path_factors=None if factors is None else factors[i], upstream=prev)
            prev = blk
        x = Fn.FinalNorm.apply(x, self.norm.weight, self.norm.bias, self.norm.eps, prev is not None,
                               None if prev is None else prev.last_scale2)
        x._getmi_encoder_tokens = True      # what ExpressionHead(fused_softplus=True) insists on
        return x, None
