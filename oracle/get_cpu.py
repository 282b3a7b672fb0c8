"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch CPU (fp32, plain torch ops) restatement of the reference's region-transformer hot path:
GETRegionFinetune / GETRegionFinetuneATAC / GETRegionPretrain forward, loss and (via autograd) backward.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
file, and only as the checker / the CPU baseline -- never as a fallback of the product path.

Parity pinning: the reference ships no tests or golden vectors for this path, so
this restatement is pinned against outputs of the reference itself, imported in the build container by
``oracle/ref_import.py`` and captured by ``oracle/make_golden.py`` into ``tests/golden/*.npz``
(``tests/test_oracle_golden.py`` checks every fixture; tolerance 2e-5 relative, fp32).

Each function cites the reference lines (relative to /root/reference/get_model) it restates.
State is a plain ``dict[str, Tensor]`` with exactly the reference's state_dict keys.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-6          # model/transformer.py:315  norm_layer=partial(nn.LayerNorm, eps=1e-6)
POISSON_EPS = 1e-8     # torch.nn.PoissonNLLLoss default eps (config/model/GETRegionFinetune.yaml:34-38)


# --------------------------------------------------------------------------------------------
# deterministic, framework-independent tensors (weights and inputs are never committed)
# --------------------------------------------------------------------------------------------
def hash_uniform(n, seed):
    """n float64 values in [0,1): splitmix64 of (index, seed), integer arithmetic only."""
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def hash_tensor(shape, seed, lo=-1.0, hi=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(n, seed)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32)).reshape(shape)


def state_dict_spec(kind, num_layers=12, embed_dim=768, num_motif=283, output_dim=None, mlp_ratio=4):
    """Ordered (name, shape) list == reference state_dict layout (probed)."""
    D, F_ = embed_dim, embed_dim * mlp_ratio
    if output_dim is None:
        output_dim = {"finetune": 2, "atac": 1, "pretrain": num_motif}[kind]
    spec = []
    if kind == "pretrain":
        spec.append(("mask_token", (1, 1, D)))
    spec += [("cls_token", (1, 1, D)),
             ("region_embed.embed.weight", (D, num_motif)), ("region_embed.embed.bias", (D,))]
    for i in range(num_layers):
        p = f"encoder.blocks.{i}."
        spec += [(p + "norm1.weight", (D,)), (p + "norm1.bias", (D,)),
                 (p + "attn.q_bias", (D,)), (p + "attn.v_bias", (D,)),
                 (p + "attn.qkv.weight", (3 * D, D)),
                 (p + "attn.proj.weight", (D, D)), (p + "attn.proj.bias", (D,)),
                 (p + "norm2.weight", (D,)), (p + "norm2.bias", (D,)),
                 (p + "mlp.fc1.weight", (F_, D)), (p + "mlp.fc1.bias", (F_,)),
                 (p + "mlp.fc2.weight", (D, F_)), (p + "mlp.fc2.bias", (D,))]
    spec += [("encoder.norm.weight", (D,)), ("encoder.norm.bias", (D,))]
    if kind == "pretrain":
        spec += [("head_mask.weight", (output_dim, D)), ("head_mask.bias", (output_dim,))]
    else:
        spec += [("head_exp.head.weight", (output_dim, D)), ("head_exp.head.bias", (output_dim,))]
    return spec


def fill_state_dict(kind, seed=0, **kw):
    """Deterministic non-trivial weights: xavier-uniform-scaled matrices (model/model.py:214-221 scale),
    and -- unlike a fresh init -- NON-zero biases, LN affine != (1,0), non-zero q_bias/v_bias/cls_token so
    every term of the path is exercised."""
    sd = {}
    for idx, (name, shape) in enumerate(state_dict_spec(kind, **kw)):
        s = seed * 1000 + idx + 1
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "encoder.norm.weight":
            t = hash_tensor(shape, s, 0.8, 1.2)
        elif len(shape) == 2:
            a = math.sqrt(6.0 / (shape[0] + shape[1]))
            t = hash_tensor(shape, s, -a, a)
        elif name in ("cls_token", "mask_token"):
            t = hash_tensor(shape, s, -0.5, 0.5)
        else:  # biases
            t = hash_tensor(shape, s, -0.1, 0.1)
        sd[name] = t
    return sd


def make_batch(B, N, seed, num_motif=283, kind="finetune"):
    """Synthetic batch with the dataset's keys/dtypes (dataset/zarr_dataset.py:1161-1169)."""
    u = hash_tensor((B, N, num_motif), 7000 + seed, 0.0, 1.0)
    z = hash_tensor((B, N, num_motif), 7100 + seed, 0.0, 1.0)
    region_motif = torch.where(z < 0.6, torch.zeros_like(u), u)      # ~60 % zeros, rest U[0,1)
    region_motif[:, :, -1] = hash_tensor((B, N), 7200 + seed, 0.01, 1.0)  # ATAC column in (0,1]
    lab = hash_tensor((B, N, 2), 7300 + seed, 0.0, 4.0)
    lz = hash_tensor((B, N, 2), 7400 + seed, 0.0, 1.0)
    exp_label = torch.where(lz < 0.9, torch.zeros_like(lab), lab)     # 90 % zeros (non-TSS)
    mask = (hash_tensor((B, N), 7500 + seed, 0.0, 1.0) < 0.5).to(torch.int64)
    return {"region_motif": region_motif.contiguous(), "exp_label": exp_label.contiguous(), "mask": mask}


def drop_path_rates(num_layers, drop_path_rate):
    # model/transformer.py:323-325
    return [x.item() for x in torch.linspace(0, drop_path_rate, num_layers)]


# --------------------------------------------------------------------------------------------
# the path
# --------------------------------------------------------------------------------------------
def attention(x, sd, p, num_heads):
    """model/transformer.py:46-99 (mask=None, bias=None branch)."""
    B, T, C = x.shape
    d = C // num_heads
    qkv_bias = torch.cat((sd[p + "attn.q_bias"], torch.zeros_like(sd[p + "attn.v_bias"]), sd[p + "attn.v_bias"]))
    qkv = F.linear(x, sd[p + "attn.qkv.weight"], qkv_bias)                 # :57
    qkv = qkv.reshape(B, T, 3, num_heads, d).permute(2, 0, 3, 1, 4)       # :58
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (d ** -0.5)                                                    # :61
    attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)                       # :62,93
    x = (attn @ v).transpose(1, 2).reshape(B, T, C)                        # :96
    return F.linear(x, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"]), attn   # :97


def mlp(x, sd, p):
    """model/transformer.py:171-177: fc1 -> exact (erf) GELU -> fc2."""
    h = F.linear(x, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    return F.linear(F.gelu(h), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def drop_path(x, keep, keep_prob):
    """timm DropPath semantics (call site model/transformer.py:220,247-248): per-sample keep mask,
    kept samples scaled by 1/keep_prob.  ``keep`` is a [B] 0/1 tensor or None (= identity)."""
    if keep is None:
        return x
    return x * (keep.to(x.dtype) / keep_prob).reshape(-1, 1, 1)


def block(x, sd, i, num_heads, keep1=None, keep2=None, keep_prob=1.0):
    """model/transformer.py:240-248 (gamma_1 is None branch)."""
    p = f"encoder.blocks.{i}."
    h = F.layer_norm(x, (x.shape[-1],), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
    a, _ = attention(h, sd, p, num_heads)
    x = x + drop_path(a, keep1, keep_prob)
    h = F.layer_norm(x, (x.shape[-1],), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
    x = x + drop_path(mlp(h, sd, p), keep2, keep_prob)
    return x


def encoder(x, sd, num_layers, num_heads, keep_masks=None, drop_path_rate=0.1):
    """model/transformer.py:358-367.  keep_masks: None (eval) or list of (keep1, keep2) per block."""
    dpr = drop_path_rates(num_layers, drop_path_rate)
    for i in range(num_layers):
        k1 = k2 = None
        if keep_masks is not None and dpr[i] > 0.0:       # nn.Identity when rate == 0 (:220)
            k1, k2 = keep_masks[i]
        x = block(x, sd, i, num_heads, k1, k2, 1.0 - dpr[i])
    return F.layer_norm(x, (x.shape[-1],), sd["encoder.norm.weight"], sd["encoder.norm.bias"], LN_EPS)


def embed(region_motif, sd, mask=None):
    """model/modules.py:101-103 + model/model.py:520-525 (pretrain blend) / :571-573 (cls concat)."""
    x = F.linear(region_motif, sd["region_embed.embed.weight"], sd["region_embed.embed.bias"])
    B, N, C = x.shape
    if mask is not None:
        w = mask.to(x.dtype)
        x = x * (1 - w) + sd["mask_token"].expand(B, N, -1) * w
    return torch.cat((sd["cls_token"].expand(B, -1, -1), x), dim=1)


def finetune_forward(sd, region_motif, num_layers=12, num_heads=12, keep_masks=None, drop_path_rate=0.1):
    """GETRegionFinetune.forward model/model.py:568-577 (also ATAC variant :677-686; head width from sd)."""
    x = encoder(embed(region_motif, sd), sd, num_layers, num_heads, keep_masks, drop_path_rate)
    x = x[:, 1:]
    return F.softplus(F.linear(x, sd["head_exp.head.weight"], sd["head_exp.head.bias"]))


def pretrain_forward(sd, region_motif, mask, num_layers=12, num_heads=12, keep_masks=None, drop_path_rate=0.1):
    """GETRegionPretrain.forward model/model.py:518-529.  mask: bool [B,N,1]."""
    x = encoder(embed(region_motif, sd, mask), sd, num_layers, num_heads, keep_masks, drop_path_rate)
    C = x.shape[-1]
    x = x[:, 1:][mask.squeeze(-1)].reshape(-1, C)
    return F.linear(x, sd["head_mask.weight"], sd["head_mask.bias"])


def poisson_nll(pred, obs):
    """torch.nn.PoissonNLLLoss(log_input=False, full=False, reduction='mean')."""
    return (pred - obs * torch.log(pred + POISSON_EPS)).mean()


def finetune_loss(sd, batch, kind="finetune", **kw):
    """get_input -> forward -> before_loss -> GETLoss -> after_loss
    (run.py:210-224; model/model.py:563-566,579-583,144-149,238-244; ATAC :670-675,688-692)."""
    rm = batch["region_motif"]
    if kind == "atac":
        inp = rm.clone()
        inp[:, :, -1] = 1
        pred = finetune_forward(sd, inp, **kw)
        obs = rm[:, :, -1].unsqueeze(-1)
    else:
        pred = finetune_forward(sd, rm, **kw)
        obs = batch["exp_label"]
    return poisson_nll(pred, obs) * 1.0, pred


def pretrain_loss(sd, batch, **kw):
    """model/model.py:512-516,531-536 + MSELoss(mean) (config/model/GETRegionPretrain.yaml:31-37)."""
    rm = batch["region_motif"]
    mask = batch["mask"].unsqueeze(-1).bool()
    pred = pretrain_forward(sd, rm, mask, **kw)
    obs = rm[mask.squeeze(-1)].reshape(-1, rm.shape[-1])
    return F.mse_loss(pred, obs) * 1.0, pred


def loss_and_grads(kind, sd, batch, **kw):
    """One fwd+bwd 'training step' of the oracle; returns (loss, pred, {name: grad})."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    if kind == "pretrain":
        loss, pred = pretrain_loss(leaf, batch, **kw)
    else:
        loss, pred = finetune_loss(leaf, batch, kind=kind, **kw)
    loss.backward()
    return loss.detach(), pred.detach(), {k: v.grad for k, v in leaf.items()}
