"""Autograd operators of the GET region path, each one a sequence of libgetmi355 kernel launches.

Granularity follows the reference's module boundaries (`region_embed`, `encoder.blocks.i`,
`encoder.norm`+head must stay real submodules whose outputs are autograd tensors), fused *inside* them:

  RegionEmbedTokens   model/modules.py:101-103 + model/model.py:520-522   (embed, mask-token blend)
  ConcatCLS           model/model.py:523-525, 571-573   (CLS row, filled in place)
  EncoderBlock        model/transformer.py:240-248 (LN, qkv, attention, proj, residual, LN, MLP, residual)
  FinalNorm           model/transformer.py:364
  ExpressionHeadSoftplus   model/model.py:575-576
  MaskedHead          model/model.py:527-528
  FusedLoss           model/model.py:144-149 with PoissonNLLLoss / MSELoss

Numerics: fp32 master weights and fp32 residual stream; bf16 MFMA operands with fp32 accumulation; bf16
saved activations.  There is no eager fallback: CPU tensors raise.
"""
import os
import weakref

import torch

from . import kernels as K

BF16 = torch.bfloat16


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("get_model_amd: the GET hot path runs only on an MI355X GPU (HIP kernels); got a CPU "
                           "tensor and there is deliberately no CPU fallback")


class _WeightCache:
    """bf16 (and transposed bf16) copies of fp32 master weights, refreshed when the parameter changes
    (in-place optimizer steps bump `_version`)."""

    def __init__(self):
        self._c = {}
        self.epoch = 0      # bumped after every optimizer step (see _after_optimizer_step)

    def get(self, w, kind, make):
        if not isinstance(w, torch.nn.Parameter):
            return make(w)          # e.g. a LoRA-parametrised weight: new tensor every access
        key = (id(w), kind)
        hit = self._c.get(key)
        tag = (w._version, w.data_ptr(), tuple(w.shape), self.epoch)
        # id() and data_ptr() are both recycled once a model is freed: the entry also pins the identity of the
        # parameter it was made from (weak reference, dropped together with the parameter).
        if hit is not None and hit[0] == tag and hit[2]() is w:
            return hit[1]
        val = make(w.detach())
        self._c[key] = (tag, val, weakref.ref(w, lambda _r, k=key, c=self._c: c.pop(k, None) if (k in c and c[k][2]() is None) else None))
        return val

    def peek(self, w, kind):
        """The cached copy of `w` if it is current, else None (never creates one)."""
        hit = self._c.get((id(w), kind))
        if hit is None or hit[2]() is not w or hit[0][:3] != (w._version, w.data_ptr(), tuple(w.shape)) or \
                hit[0][3] < self.epoch - 1:
            return None
        return hit[1]

    def retag(self, w, kind):
        """Declare the cached copy current again (its storage was rewritten in place, e.g. by optim.FusedAdamW)."""
        key = (id(w), kind)
        hit = self._c[key]
        self._c[key] = ((w._version, w.data_ptr(), tuple(w.shape), self.epoch), hit[1], hit[2])

    def clear(self):
        self._c.clear()


WEIGHTS = _WeightCache()


def _after_optimizer_step(optimizer, args, kwargs):
    # Fused / foreach optimizers (torch.optim.AdamW(fused=True), apex, ...) update parameters without bumping
    # `_version`, so the version tag alone would keep serving stale bf16 copies: every optimizer step of any
    # optimizer in the process starts a new cache epoch.  Code that writes `param.data` by hand must call
    # WEIGHTS.clear() itself.
    if not getattr(optimizer, "getmi_refreshes_weight_cache", False):   # optim.FusedAdamW maintains the copies itself
        WEIGHTS.epoch += 1


from torch.optim.optimizer import register_optimizer_step_post_hook  # noqa: E402

register_optimizer_step_post_hook(_after_optimizer_step)


def w_bf16(w):
    return WEIGHTS.get(w, "n", lambda t: K.cast_bf16(t.contiguous()))


def w_bf16_t(w):
    return WEIGHTS.get(w, "t", lambda t: K.cast_transpose(t.contiguous()))


def grad_buffer(w):
    """Destination for a weight gradient: a fresh alias of the DDP bucket slot (ddp.GradBuckets sets
    `getmi_grad_buffer`) when the parameter has no gradient yet, so autograd adopts it as `w.grad` with no copy;
    None (-> a new tensor that autograd adds into w.grad) otherwise, e.g. on later micro-batches."""
    buf = getattr(w, "getmi_grad_buffer", None)
    if buf is None or w.grad is not None or getattr(w, "_getmi_slot_taken", False):
        return None
    # ONE backward node per backward pass may own the slot: a parameter used by two nodes of one graph (two forwards
    # before one backward, siamese towers) would otherwise have both write the same memory and autograd would sum two
    # aliases of it (2 x the last gradient).  The flag is cleared by the post-accumulate-grad hook of ddp.GradBuckets
    # (i.e. once autograd has accumulated every contribution of this pass) and at the start of each step.
    w._getmi_slot_taken = True
    return buf.view(buf.shape)


# Optional (GETMI_WGRAD_STREAM=1): run the weight-gradient GEMMs, which have no consumer inside backward, on a
# second HIP stream next to the data-gradient chain so that they fill the CUs a 603-tile (N = 768) GEMM leaves
# idle in its last partial round.  Measured on MI355X at the bench shape it LOSES 3 % (42.1 vs 40.8 ms/step):
# both kernels are 1-workgroup-per-CU designs that want the whole L2/LDS, so it is off by default.  Joined
# before the block's backward returns (gradients are then ordered on the main stream for autograd, the DDP
# hooks and the optimizer).
_SIDE = {}
_USE_SIDE = os.environ.get("GETMI_WGRAD_STREAM", "0") == "1"


class _SideStream:
    def __init__(self, device):
        self.main = torch.cuda.current_stream(device)
        self.side = None
        if _USE_SIDE:
            key = (device.index, self.main.cuda_stream)
            if key not in _SIDE:
                _SIDE[key] = torch.cuda.Stream(device)
            self.side = _SIDE[key]
        self.used = False

    def run(self, fn):
        """fn() on the side stream, after everything issued so far on the main stream."""
        if self.side is None:
            return fn()
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            out = fn()
        self.used = True
        return out

    def join(self):
        if self.used:
            self.main.wait_stream(self.side)
            self.used = False


class _GradHandoff:
    """dz / bias-gradient of a block's fc2 branch, produced early by its CONSUMER's LayerNorm backward.

    Block i+1 (or the final norm) ends its backward with the LayerNorm-backward kernel that writes dx, the fp32
    gradient of block i's output.  Block i would start by reading dx again to form dz2 = bf16(scale2 * dx) and its
    column sums (`scale_cast`, 39 us per layer at the bench shape); the LayerNorm-backward kernel can emit both on
    the way out.  The downstream backward stores them here keyed by the dx tensor it returns (kept alive, so its
    address cannot be reused) and by the DropPath scale it used; block i takes them only if the gradient it
    receives IS that tensor and its own scale2 IS that scale -- anything else (hooks, gradient accumulation from
    several consumers, blocks wired differently) falls back to scale_cast."""

    def __init__(self):
        self.slot = None

    def put(self, dx, scale, dz, dbias):
        self.slot = (dx, scale, dz, dbias)

    def take(self, dx_out, scale):
        slot, self.slot = self.slot, None
        if slot is None or not _USE_HANDOFF:
            return None
        dx, sc, dz, dbias = slot
        same_scale = (sc is None and scale is None) or (sc is not None and scale is not None and
                                                         sc.data_ptr() == scale.data_ptr() and sc.shape == scale.shape)
        if same_scale and dx.data_ptr() == dx_out.data_ptr() and dx.numel() == dx_out.numel() and \
                dx_out.dtype == torch.float32:
            return dz, dbias
        return None


_USE_HANDOFF = os.environ.get("GETMI_HANDOFF", "1") != "0"      # 0: always scale_cast (A/B measurements)
_ATTN_LO = os.environ.get("GETMI_ATTN_LO", "1") != "0"          # 0: delta from the bf16 attention output alone (A/B)
HANDOFF = _GradHandoff()


class _GradStream:
    """bf16 gradient stream between consecutive encoder blocks (default; GETMI_BF16_GRAD=0 turns it off).

    PyTorch's engine casts every gradient to the dtype of the forward tensor it belongs to, and the residual stream is
    fp32, so a bf16 dx cannot travel through autograd itself without a cast pass that would cost more than it saves.
    The bf16 tensor travels HERE instead, keyed by an id the producing block gave its forward output, and autograd is
    handed a zero-stride all-zero placeholder of the right shape (no memory, no kernel).  The consumer (the block that
    produced that forward tensor) takes the slot; if the gradient autograd delivers is not the placeholder itself --
    a hook changed it, or a second consumer of the tensor added its own contribution to the zeros -- that tensor is
    ADDED to the slot's gradient, so the result stays right.  The one thing the placeholder would get wrong is somebody
    LOOKING at the gradient of a block's output (retain_grad, tensor hooks: the interpretation mode of run.py:359-384);
    the producer checks the forward tensor for those and then returns the real fp32 gradient through autograd instead."""

    def __init__(self):
        self.slots = {}
        self.counter = 0
        self.zero = {}

    def new_id(self):
        self.counter += 1
        return self.counter

    def put(self, gid, payload):
        if len(self.slots) > 64:          # slots whose consumer never ran (an exception in backward): drop them
            self.slots.clear()
        self.slots[gid] = payload

    def take(self, gid):
        return self.slots.pop(gid, None) if gid is not None else None

    def placeholder(self, shape, device):
        z = self.zero.get(device)
        if z is None:
            z = self.zero[device] = torch.zeros((), dtype=torch.float32, device=device)
        return z.expand(shape)

    def is_placeholder(self, t):
        z = self.zero.get(t.device)
        return z is not None and t.data_ptr() == z.data_ptr() and all(st == 0 for st in t.stride())

    @staticmethod
    def observed(t):
        """Somebody may look at t.grad or hook it: the real gradient has to go through autograd."""
        return t is None or t.retains_grad or bool(getattr(t, "_backward_hooks", None))


# Default ON since round 3: -0.57 ms per C2 step (34.16 -> 33.57, three interleaved rounds, profiles/r03_ab_bf16grad.txt)
# with every parity bar of tests/test_model_parity_gpu.py met at unchanged measured errors (gradient cosines vs the
# oracle identical to 5 digits: the rounding of dx to bf16 is below what the bf16 GEMM operands already carry).
# GETMI_BF16_GRAD=0 restores the fp32 stream.
_BF16_GRAD = os.environ.get("GETMI_BF16_GRAD", "1") != "0"
_BLOCKED_AUX = os.environ.get("GETMI_BLOCKED_AUX", "1") != "0"      # gelu'(u) in the kernels' blocked layout (EncoderBlock)
STREAM = _GradStream()


def _pad_to(n, m):
    return (n + m - 1) // m * m


def w_bf16_kpad(w, kp):
    """[N,K] fp32 -> bf16 [N,Kp] zero-padded along K."""
    return WEIGHTS.get(w, ("kpad", kp), lambda t: K.cast_pad(t.contiguous(), kp))


# --------------------------------------------------------------------------------------------
def _token_layout(x):
    """(B, N, D) if x[B,N,D] fp32 is laid out as rows 1..N of a contiguous [B,N+1,D] token buffer, else None."""
    if x.dim() != 3 or x.dtype != torch.float32:
        return None
    B, N, D = x.shape
    if x.stride() != ((N + 1) * D, D, 1) or x.storage_offset() < D:
        return None
    if x.untyped_storage().nbytes() < (x.storage_offset() - D + B * (N + 1) * D) * 4:
        return None
    return B, N, D


def _token_alias(x):
    """The [B,N+1,D] tensor whose rows 1..N are x (see _token_layout; no copy, not tracked by autograd)."""
    B, N, D = x.shape
    return torch.empty(0, dtype=x.dtype, device=x.device).set_(
        x.untyped_storage(), x.storage_offset() - D, (B, N + 1, D), ((N + 1) * D, D, 1))


class RegionEmbedTokens(torch.autograd.Function):
    """x[B,N,D] fp32 = where(mask, mask_token, region_motif @ W^T + b)   (model/modules.py:101-103; with mask_token
    also the blend of model/model.py:520-525).

    The result is written as rows 1..N of a [B,N+1,D] buffer and returned as that strided [B,N,D] view, so that the
    CLS concat of model/model.py:571-573 (ConcatCLS below) fills row 0 in place instead of copying every token;
    `region_embed`'s output keeps the reference's shape for forward hooks (run.py:359-384)."""

    @staticmethod
    def forward(ctx, region_motif, weight, bias, mask_token, mask):
        _need_gpu(region_motif)
        B, N, Fin = region_motif.shape
        D = weight.shape[0]
        Kp = _pad_to(Fin, 64)
        a = K.cast_pad(region_motif.detach().reshape(B * N, Fin).contiguous().float(), Kp)
        wb = w_bf16_kpad(weight, Kp)
        mask_u8 = None if mask is None else mask.reshape(B * N).to(torch.uint8).contiguous()
        x0 = torch.empty(B, N + 1, D, dtype=torch.float32, device=region_motif.device)
        K.gemm_tn(K.EPI_EMBED, a, wb, x0, bias=bias.detach().contiguous(), mask=mask_u8,
                  mask_token=None if mask_token is None else mask_token.detach().reshape(D).contiguous(),
                  regions_per_sample=N)
        ctx.save_for_backward(a, mask_u8)
        ctx.dims = (B, N, Fin, D, Kp)
        ctx.has_mask_token = mask_token is not None
        ctx.weight = weight
        out = x0[:, 1:]
        out._getmi_token_rows = True        # row 0 of every sample is ours to fill (ConcatCLS)
        return out

    @staticmethod
    def backward(ctx, dx):
        a, mask_u8 = ctx.saved_tensors
        B, N, Fin, D, Kp = ctx.dims
        if _token_layout(dx) is not None:      # the gradient ConcatCLS hands back: rows 1..N of block 0's dx, no copy
            dx0 = _token_alias(dx)
        else:
            dx0 = torch.zeros(B, N + 1, D, dtype=torch.float32, device=dx.device)
            dx0[:, 1:] = dx
        dz, _dcls, dmt, dbias = K.embed_bwd_prep(dx0, mask_u8, B, N, D)
        dW = dIn = None
        if ctx.needs_input_grad[1]:
            dW = K.gemm_wgrad(dz, a)[:, :Fin]          # [D, Kp] -> [D, Fin]
        if ctx.needs_input_grad[0]:
            # d(region_motif) = dz @ W   (interpretation / Jacobian use, run.py:374-380; not on the training path)
            wt = K.cast_pad(ctx.weight.detach().t().contiguous(), _pad_to(D, 64))     # [Fin, D]
            fin_p = _pad_to(Fin, 4)
            if fin_p != Fin:
                wt = torch.cat([wt, wt.new_zeros(fin_p - Fin, wt.shape[1])], 0)
            out = torch.empty(B * N, fin_p, dtype=torch.float32, device=dz.device)
            K.gemm_tn(K.EPI_F32, dz, wt, out)
            dIn = out[:, :Fin].reshape(B, N, Fin)
        return (dIn, dW, dbias if ctx.needs_input_grad[2] else None,
                dmt.view(1, 1, D) if (ctx.has_mask_token and ctx.needs_input_grad[3]) else None, None)


class ConcatCLS(torch.autograd.Function):
    """tokens[B,N+1,D] = cat(cls_token.expand(B,1,D), x)   (model/model.py:523-525, 571-573).  For the output of
    RegionEmbedTokens the CLS row is filled in place; any other x is copied."""

    @staticmethod
    def forward(ctx, x, cls_token, in_place):
        _need_gpu(x)
        B, N, D = x.shape
        if in_place and _token_layout(x) is not None:
            full = _token_alias(x)
        else:
            full = torch.empty(B, N + 1, D, dtype=torch.float32, device=x.device)
            full[:, 1:] = x.detach()
        K.fill_cls(full, cls_token.detach().reshape(D).float().contiguous(), B, N + 1, D)
        return full

    @staticmethod
    def backward(ctx, dfull):
        dcls = dfull[:, 0].sum(0).view(1, 1, -1) if ctx.needs_input_grad[1] else None
        return dfull[:, 1:], dcls, None


def concat_cls(x, cls_token):
    in_place = bool(getattr(x, "_getmi_token_rows", False))
    full = ConcatCLS.apply(x, cls_token, in_place)
    if in_place:
        # row 0 behind x now belongs to `full`: a second concat_cls on the same region_embed output (two towers, another
        # cls_token) must take the copy path instead of overwriting the CLS row under the first result
        x._getmi_token_rows = False
    return full


# --------------------------------------------------------------------------------------------
class EncoderBlock(torch.autograd.Function):
    """One pre-LN transformer block on the fp32 residual stream x[B,T,D].

    scale1 / scale2: per-sample DropPath factors keep/keep_prob ([B] fp32) or None (identity).

    Two modes, fixed at forward time (ctx.fold):
      * fold (default, _BF16_GRAD): the DropPath factors are folded into the SAVED ACTIVATIONS -- the attention output is
        stored as s1 * O, gelu(u) and gelu'(u) as s2 * gelu(u), s2 * gelu'(u) (and the residual GEMMs add
        resid + acc + s * bias) -- so backward needs no scaled copy of the residual gradient: every GEMM consumes the
        bf16 dx of the gradient stream itself, LayerNorm backward writes dx only (10 bytes per element instead of 16),
        and attention skips dropped samples in both directions.  dx travels between blocks through STREAM (_GradStream).
      * legacy (GETMI_BF16_GRAD=0): fp32 dx through autograd, dz = bf16(s * dx) handed over by _GradHandoff."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, qkv_w, q_bias, v_bias, proj_w, proj_b, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b,
                scale1, scale2, num_heads, eps, up_block=False, up_scale=None):
        """up_block / up_scale: x is the output of another EncoderBlock whose fc2-branch DropPath factor was
        up_scale (None = identity): this block's backward then also emits what that block's backward starts from."""
        _need_gpu(x)
        B, T, D = x.shape
        M = B * T
        H = num_heads
        if D != H * 64:
            raise RuntimeError(f"get_model_amd attention kernel needs head_dim 64 (embed_dim {D}, heads {H})")
        dev = x.device
        fold = _BF16_GRAD
        x2 = x.detach().reshape(M, D).contiguous()
        att_scale = 64 ** -0.5
        # --- attention branch ---
        h1, mean1, rstd1 = K.layernorm_fwd(x2, n1w.detach(), n1b.detach(), eps)
        qkv = torch.empty(M, 3 * D, dtype=BF16, device=dev)
        K.gemm_tn(K.EPI_BF16, h1, w_bf16(qkv_w), qkv, bias=K.qkv_bias(q_bias.detach(), v_bias.detach()))
        # (the residual of the bf16 output is kept only when q_bias will get a gradient: see kernels.attention_fwd)
        needs_bwd = any(ctx.needs_input_grad)      # all False under no_grad / inference_mode
        a_lo = None
        s_att = scale1 if fold else None           # fold: a = s1 * O, dropped samples skipped
        if needs_bwd and ctx.needs_input_grad[4] and _ATTN_LO:
            a, lse, a_lo = K.attention_fwd(qkv, B, T, H, att_scale, want_lo=True, sample_scale=s_att)
        else:
            a, lse = K.attention_fwd(qkv, B, T, H, att_scale, sample_scale=s_att)
        x_mid = torch.empty(M, D, dtype=torch.float32, device=dev)
        K.gemm_tn(K.EPI_BIAS_RESID, a, w_bf16(proj_w), x_mid, bias=proj_b.detach(), resid=x2, rowscale=scale1,
                  rows_per_sample=T, prescaled=fold and scale1 is not None)
        # --- MLP branch ---
        h2, mean2, rstd2 = K.layernorm_fwd(x_mid, n2w.detach(), n2b.detach(), eps)
        Fh = fc1_w.shape[0]
        # u holds gelu'(fc1 output), not the pre-activation: the backward epilogue is then a plain multiply.  Nobody but
        # that epilogue reads it, so where the kernels allow it travels in their own blocked layout (getmi355.h)
        blocked = needs_bwd and _BLOCKED_AUX and K.gemm_tn_blocked_ok(M, Fh, D)
        u = torch.empty(K.blocked_rows(M) if blocked else M, Fh, dtype=BF16, device=dev) if needs_bwd else None
        g = torch.empty(M, Fh, dtype=BF16, device=dev)                            # (inference: gelu'(.) is not stored)
        K.gemm_tn(K.EPI_BIAS_GELUG, h2, w_bf16(fc1_w), u, out1=g, bias=fc1_b.detach(),
                  rowscale=scale2 if fold else None, rows_per_sample=T, M=M, blocked=blocked)
        x_out = torch.empty(M, D, dtype=torch.float32, device=dev)
        K.gemm_tn(K.EPI_BIAS_RESID, g, w_bf16(fc2_w), x_out, bias=fc2_b.detach(), resid=x_mid, rowscale=scale2,
                  rows_per_sample=T, prescaled=fold and scale2 is not None)
        ctx.save_for_backward(x2, mean1, rstd1, h1, qkv, lse, a, x_mid, mean2, rstd2, h2, u, g, scale1, scale2,
                              n1w, n2w, a_lo)
        ctx.weights = (qkv_w, proj_w, fc1_w, fc2_w)
        ctx.dims = (B, T, D, H, att_scale)
        ctx.up = (bool(up_block), up_scale)
        ctx.fold = fold
        ctx.blocked = blocked
        out = x_out.view(B, T, D)
        ctx.gids = (None, None)
        ctx.x_ref = None
        if fold and needs_bwd:
            gid = STREAM.new_id()
            ctx.gids = (getattr(x, "_getmi_gid", None) if up_block else None, gid)
            ctx.x_ref = x if up_block else None         # looked at in backward: is anybody observing its gradient?
            out._getmi_gid = gid
        return out

    @staticmethod
    def backward(ctx, dx_out):
        if ctx.fold:
            return EncoderBlock._backward_fold(ctx, dx_out)
        return EncoderBlock._backward_legacy(ctx, dx_out)

    @staticmethod
    def _backward_fold(ctx, dx_out):
        (x2, mean1, rstd1, h1, qkv, lse, a, x_mid, mean2, rstd2, h2, u, g, scale1, scale2, n1w, n2w, a_lo) = ctx.saved_tensors
        qkv_w, proj_w, fc1_w, fc2_w = ctx.weights
        B, T, D, H, att_scale = ctx.dims
        M = B * T
        dev = dx_out.device
        need = ctx.needs_input_grad
        in_gid, out_gid = ctx.gids
        # ---- the gradient of this block's output: bf16 from the stream, or whatever autograd delivered ----
        slot = STREAM.take(out_gid)
        dxb = dfc2_b = None
        if slot is not None:
            sdx, sc_, _unused, sdb = slot
            same_scale = (sc_ is None and scale2 is None) or (sc_ is not None and scale2 is not None and
                                                               sc_.data_ptr() == scale2.data_ptr())
            if STREAM.is_placeholder(dx_out):
                if same_scale:
                    dxb, dfc2_b = sdx, sdb
                else:                       # (cannot happen with GETTransformer's wiring; kept correct anyway)
                    dx_out = sdx.float()
            else:       # autograd delivered more than the placeholder (a hook, a second consumer of the block's output)
                dx_out = sdx.float().add_(dx_out.reshape(M, D))
        if dxb is None:
            full = dx_out.reshape(M, D).float().contiguous()
            dxb, db0 = K.scale_cast(full, None, T)
            dfc2_b = db0 if scale2 is None else K.scale_cast(full, scale2, T)[1]
        # ---- MLP branch (g = s2 * gelu(u) and u = s2 * gelu'(u) carry the DropPath factor) ----
        dfc2_w = K.gemm_wgrad(dxb, g, out=grad_buffer(fc2_w)) if need[12] else None
        du = torch.empty(M, u.shape[1], dtype=BF16, device=dev)
        dfc1_b = torch.empty(u.shape[1], dtype=torch.float32, device=dev) if need[11] else None
        K.gemm_tn(K.EPI_DMUL, dxb, w_bf16_t(fc2_w), du, aux=u, dbias=dfc1_b, blocked=ctx.blocked)   # (dx @ W2) * s2 * gelu'(.), + column sums
        dfc1_w = K.gemm_wgrad(du, h2, out=grad_buffer(fc1_w)) if need[10] else None
        dh2 = torch.empty(M, D, dtype=BF16, device=dev)
        K.gemm_tn(K.EPI_BF16, du, w_bf16_t(fc1_w), dh2)                        # du @ W1
        # dx_mid (bf16) only; the proj bias gradient = column sums of s1 * dx_mid, formed without storing that tensor
        dx_mid, _, dn2w, dn2b, dproj_b = K.layernorm_bwd(dh2, x_mid, mean2, rstd2, n2w.detach(), dres=dxb,
                                                         rowscale=scale1, rows_per_sample=T, want_dz=False,
                                                         want_dbias=True, dx_dtype=BF16)
        # ---- attention branch (a = s1 * O) ----
        da = torch.empty(M, D, dtype=BF16, device=dev)
        K.gemm_tn(K.EPI_BF16, dx_mid, w_bf16_t(proj_w), da)                    # dx_mid @ Wproj = gradient w.r.t. a
        dq_bias = dv_bias = None
        if need[4] or need[5]:
            dqkv, dq_bias, dv_bias = K.attention_bwd(qkv, a, da, lse, B, T, H, att_scale, want_bias=True, out_lo=a_lo,
                                                     sample_scale=scale1)
        else:
            dqkv = K.attention_bwd(qkv, a, da, lse, B, T, H, att_scale, sample_scale=scale1)
        dproj_w = dqkv_w = None
        if need[6] and need[3]:
            dproj_w, dqkv_w = K.gemm_wgrad_pair(dx_mid, a, dqkv, h1, out_a=grad_buffer(proj_w), out_b=grad_buffer(qkv_w))
        else:
            if need[6]:
                dproj_w = K.gemm_wgrad(dx_mid, a, out=grad_buffer(proj_w))
            if need[3]:
                dqkv_w = K.gemm_wgrad(dqkv, h1, out=grad_buffer(qkv_w))
        dh1 = torch.empty(M, D, dtype=BF16, device=dev)
        K.gemm_tn(K.EPI_BF16, dqkv, w_bf16_t(qkv_w), dh1)                      # dqkv @ Wqkv
        up_block, up_scale = ctx.up
        # the gradient handed to the block below: bf16 through STREAM when that block will take it and nobody is
        # watching the tensor's gradient; else fp32 through autograd
        to_stream = up_block and in_gid is not None and need[0] and not STREAM.observed(ctx.x_ref)
        dx_in, _, dn1w, dn1b, dbias_up = K.layernorm_bwd(dh1, x2, mean1, rstd1, n1w.detach(), dres=dx_mid,
                                                         rowscale=up_scale if to_stream else None, rows_per_sample=T,
                                                         want_dz=False, want_dbias=to_stream,
                                                         dx_dtype=BF16 if to_stream else torch.float32)
        if to_stream:
            STREAM.put(in_gid, (dx_in, up_scale, None, dbias_up))
            dx_ret = STREAM.placeholder((B, T, D), dev)
        else:
            dx_ret = dx_in.view(B, T, D)
        return (dx_ret, dn1w, dn1b, dqkv_w, dq_bias, dv_bias, dproj_w, dproj_b, dn2w, dn2b, dfc1_w,
                dfc1_b, dfc2_w, dfc2_b, None, None, None, None, None, None)

    @staticmethod
    def _backward_legacy(ctx, dx_out):
        (x2, mean1, rstd1, h1, qkv, lse, a, x_mid, mean2, rstd2, h2, u, g, scale1, scale2, n1w, n2w, a_lo) = ctx.saved_tensors
        qkv_w, proj_w, fc1_w, fc2_w = ctx.weights
        B, T, D, H, att_scale = ctx.dims
        M = B * T
        dev = dx_out.device
        need = ctx.needs_input_grad
        dx_out = dx_out.reshape(M, D).contiguous()
        side = _SideStream(dev)
        # ---- MLP branch ----
        # (each weight-gradient GEMM is issued on the side stream BEFORE the data-gradient GEMM that shares its
        #  operand, so it waits only for the operand's producer and the two run side by side)
        hit = HANDOFF.take(dx_out, scale2)
        dz2, dfc2_b = hit if hit is not None else K.scale_cast(dx_out, scale2, T)
        dfc2_w = side.run(lambda: K.gemm_wgrad(dz2, g, out=grad_buffer(fc2_w))) if need[12] else None
        du = torch.empty(M, u.shape[1], dtype=BF16, device=dev)
        dfc1_b = torch.empty(u.shape[1], dtype=torch.float32, device=dev) if need[11] else None
        K.gemm_tn(K.EPI_DMUL, dz2, w_bf16_t(fc2_w), du, aux=u, dbias=dfc1_b, blocked=ctx.blocked)   # (dz2 @ W2) * gelu'(.), + column sums
        dfc1_w = side.run(lambda: K.gemm_wgrad(du, h2, out=grad_buffer(fc1_w))) if need[10] else None
        dh2 = torch.empty(M, D, dtype=BF16, device=dev)
        K.gemm_tn(K.EPI_BF16, du, w_bf16_t(fc1_w), dh2)                        # du @ W1
        dx_mid, dz1, dn2w, dn2b, dproj_b = K.layernorm_bwd(dh2, x_mid, mean2, rstd2, n2w.detach(), dres=dx_out,
                                                           rowscale=scale1, rows_per_sample=T, want_dz=True,
                                                           want_dbias=True)
        # ---- attention branch ----
        # (attn.proj's weight gradient has 9 output tiles: it shares a launch with attn.qkv's further down)
        pair = need[6] and need[3] and side.side is None
        dproj_w = None
        if need[6] and not pair:
            dproj_w = side.run(lambda: K.gemm_wgrad(dz1, a, out=grad_buffer(proj_w)))
        da = torch.empty(M, D, dtype=BF16, device=dev)
        K.gemm_tn(K.EPI_BF16, dz1, w_bf16_t(proj_w), da)                       # dz1 @ Wproj
        # q_bias / v_bias gradients (the K third has no bias, model/transformer.py:53) come out of the attention
        # backward's fp32 accumulators
        dq_bias = dv_bias = None
        if need[4] or need[5]:
            dqkv, dq_bias, dv_bias = K.attention_bwd(qkv, a, da, lse, B, T, H, att_scale, want_bias=True, out_lo=a_lo)
        else:
            dqkv = K.attention_bwd(qkv, a, da, lse, B, T, H, att_scale)
        if pair:
            dproj_w, dqkv_w = K.gemm_wgrad_pair(dz1, a, dqkv, h1, out_a=grad_buffer(proj_w), out_b=grad_buffer(qkv_w))
        else:
            dqkv_w = side.run(lambda: K.gemm_wgrad(dqkv, h1, out=grad_buffer(qkv_w))) if need[3] else None
        dh1 = torch.empty(M, D, dtype=BF16, device=dev)
        K.gemm_tn(K.EPI_BF16, dqkv, w_bf16_t(qkv_w), dh1)                      # dqkv @ Wqkv
        up_block, up_scale = ctx.up
        dx_in, dz_up, dn1w, dn1b, dbias_up = K.layernorm_bwd(dh1, x2, mean1, rstd1, n1w.detach(), dres=dx_mid,
                                                             rowscale=up_scale if up_block else None,
                                                             rows_per_sample=T, want_dz=up_block, want_dbias=up_block)
        if up_block:
            HANDOFF.put(dx_in, up_scale, dz_up, dbias_up)
        side.join()
        return (dx_in.view(B, T, D), dn1w, dn1b, dqkv_w, dq_bias, dv_bias, dproj_w, dproj_b, dn2w, dn2b, dfc1_w,
                dfc1_b, dfc2_w, dfc2_b, None, None, None, None, None, None)


# --------------------------------------------------------------------------------------------
class FinalNorm(torch.autograd.Function):
    """encoder.norm: fp32 tokens -> bf16 normalised tokens (consumed by the heads)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, up_block=False, up_scale=None):
        _need_gpu(x)
        B, T, D = x.shape
        x2 = x.detach().reshape(B * T, D).contiguous()
        y, mean, rstd = K.layernorm_fwd(x2, w.detach(), b.detach(), eps)
        ctx.save_for_backward(x2, mean, rstd, w)
        ctx.dims = (B, T, D)
        ctx.up = (bool(up_block), up_scale)      # as in EncoderBlock.forward
        ctx.fold = _BF16_GRAD
        ctx.in_gid = getattr(x, "_getmi_gid", None) if (ctx.fold and up_block) else None
        ctx.x_ref = x if ctx.in_gid is not None else None
        return y.view(B, T, D)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, w = ctx.saved_tensors
        B, T, D = ctx.dims
        dy = dy.reshape(B * T, D)
        if dy.dtype != BF16:
            dy = dy.to(BF16)
        up_block, up_scale = ctx.up
        if ctx.fold:        # see _GradStream / EncoderBlock._backward_fold
            to_stream = ctx.in_gid is not None and ctx.needs_input_grad[0] and not STREAM.observed(ctx.x_ref)
            dx, _, dw, db, dbias_up = K.layernorm_bwd(dy.contiguous(), x2, mean, rstd, w.detach(),
                                                      rowscale=up_scale if to_stream else None, rows_per_sample=T,
                                                      want_dz=False, want_dbias=to_stream,
                                                      dx_dtype=BF16 if to_stream else torch.float32)
            if to_stream:
                STREAM.put(ctx.in_gid, (dx, up_scale, None, dbias_up))
                return STREAM.placeholder((B, T, D), dy.device), dw, db, None, None, None
            return dx.view(B, T, D), dw, db, None, None, None
        dx, dz_up, dw, db, dbias_up = K.layernorm_bwd(dy.contiguous(), x2, mean, rstd, w.detach(),
                                                      rowscale=up_scale if up_block else None, rows_per_sample=T,
                                                      want_dz=up_block, want_dbias=up_block)
        if up_block:
            HANDOFF.put(dx, up_scale, dz_up, dbias_up)
        return dx.view(B, T, D), dw, db, None, None, None


class ExpressionHeadSoftplus(torch.autograd.Function):
    """pred[B,N,out] = softplus(hf[:, 1:] @ W^T + b) on bf16 normalised tokens hf[B,N+1,D]."""

    @staticmethod
    def forward(ctx, hf, W, b):
        _need_gpu(hf)
        B, T, D = hf.shape
        hf2 = hf.detach().reshape(B * T, D).contiguous()
        Wd = W.detach().contiguous()
        logits, pred = K.head_fwd(hf2, Wd, b.detach().contiguous(), B, T - 1)
        ctx.save_for_backward(hf2, logits, Wd)
        ctx.dims = (B, T, D)
        return pred

    @staticmethod
    def backward(ctx, dpred):
        hf2, logits, Wd = ctx.saved_tensors
        B, T, D = ctx.dims
        dh, dW, db = K.head_bwd(dpred.contiguous().float(), logits, hf2, Wd, B, T - 1)
        return dh.view(B, T, D), dW, db


class HeadLinear(torch.autograd.Function):
    """logits[..., out] = x[..., D] @ W^T + b for ANY float GPU tensor x: the reference's stand-alone
    `ExpressionHead.forward` contract (model/modules.py:255-258: a plain nn.Linear on [B,N,D], no CLS handling, no
    Softplus).  Off the training path (the models call the fused ExpressionHeadSoftplus); same GEMM kernels."""

    @staticmethod
    def forward(ctx, x, W, b):
        _need_gpu(x)
        D = x.shape[-1]
        out_dim = W.shape[0]
        if W.shape[1] != D or D % 64 != 0:
            raise RuntimeError(f"HeadLinear: last dimension {D} does not match the head weight {tuple(W.shape)} "
                               f"(or is not a multiple of 64)")
        x2 = x.detach().reshape(-1, D)
        xb = x2.contiguous() if x2.dtype == BF16 else K.cast_pad(x2.float().contiguous(), D)
        M = xb.shape[0]
        Np = _pad_to(out_dim, 4)
        ctx.dims = (tuple(x.shape), out_dim, x.dtype)
        ctx.W = W
        ctx.save_for_backward(xb)
        if M == 0:
            return torch.zeros(*x.shape[:-1], out_dim, dtype=torch.float32, device=x.device)
        bd = b.detach().float()
        bp = bd if Np == out_dim else torch.cat([bd, bd.new_zeros(Np - out_dim)])
        out = torch.empty(M, Np, dtype=torch.float32, device=x.device)
        K.gemm_tn(K.EPI_F32, xb, w_rows_padded(W, Np), out, bias=bp.contiguous())
        return out[:, :out_dim].reshape(*x.shape[:-1], out_dim)

    @staticmethod
    def backward(ctx, dout):
        (xb,) = ctx.saved_tensors
        W = ctx.W
        shape, out_dim, in_dtype = ctx.dims
        D = shape[-1]
        M = xb.shape[0]
        dev = dout.device
        if M == 0:
            return (torch.zeros(shape, dtype=in_dtype, device=dev), torch.zeros_like(W),
                    torch.zeros(out_dim, dtype=torch.float32, device=dev))
        Kp = _pad_to(out_dim, 64)
        dop = K.cast_pad(dout.reshape(M, out_dim).contiguous().float(), Kp)
        dx = None
        if ctx.needs_input_grad[0]:
            dxb = torch.empty(M, D, dtype=BF16, device=dev)
            K.gemm_tn(K.EPI_BF16, dop, w_t_cols_padded(W, Kp), dxb)
            dx = dxb.to(in_dtype).reshape(shape)
        dW = K.gemm_wgrad(dop, xb)[:out_dim] if ctx.needs_input_grad[1] else None
        db = K.colsum_bf16(dop)[:out_dim] if ctx.needs_input_grad[2] else None
        return dx, dW, db


class MaskedRows:
    """Row lists of the masked regions of one batch (GETRegionPretrain: `x[:, 1:][mask]` and `x_original[mask]`,
    model/model.py:527,535), built on the device at the START of forward.

    The number of masked rows sets the shape of the model's output, so the host has to learn it; what it must not do
    is wait for the encoder to find out (torch.nonzero after the encoder drains the launch queue and the GPU idles
    while the host catches up).  The compaction kernel and an asynchronous 4-byte copy into pinned memory are
    enqueued before the embed GEMM; `n` waits on the copy's event only when the head needs the shape, at which
    point the copy finished long ago and the encoder's launches are already queued behind it."""

    def __init__(self, mask):
        _need_gpu(mask)
        m = mask.reshape(mask.shape[0], -1)
        self.B, self.N = m.shape
        self.mask = mask
        self.mask_u8 = m.reshape(-1).to(torch.uint8).contiguous()
        self.token_rows, self.region_rows, count = K.mask_compact(self.mask_u8, self.B, self.N)
        self._host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        self._host.copy_(count, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()
        self._count = count          # keeps the device word alive until the copy has run
        self._n = None

    @property
    def n(self):
        if self._n is None:
            self._event.synchronize()
            self._n = int(self._host[0])
        return self._n


def w_rows_padded(w, rows_p):
    """[R,K] fp32 -> bf16 [Rp,K], zero rows appended (cached like the other operand copies)."""
    def make(t):
        pad = t.new_zeros(rows_p - t.shape[0], t.shape[1])
        return K.cast_bf16(torch.cat([t, pad], 0).contiguous())
    return WEIGHTS.get(w, ("rpad", rows_p), make)


def w_t_cols_padded(w, cols_p):
    """[R,K] fp32 -> bf16 [K,Rp] = W^T with zero columns appended."""
    def make(t):
        wt = t.new_zeros(t.shape[1], cols_p)
        wt[:, :t.shape[0]] = t.t()
        return K.cast_bf16(wt)
    return WEIGHTS.get(w, ("tpad", cols_p), make)


class MaskedHead(torch.autograd.Function):
    """x_masked[n_masked, out] = hf[:, 1:][mask] @ W^T + b   (row order = torch boolean indexing order).
    rows: MaskedRows of the same mask."""

    @staticmethod
    def forward(ctx, hf, W, b, rows):
        _need_gpu(hf)
        B, T, D = hf.shape
        out_dim = W.shape[0]
        if (rows.B, rows.N) != (B, T - 1):
            raise RuntimeError("MaskedHead: mask shape does not match the token tensor")
        n = rows.n
        idx = rows.token_rows[:n]
        hf2 = hf.detach().reshape(B * T, D).contiguous()
        Np = _pad_to(out_dim, 4)
        ctx.dims = (B, T, D, out_dim, n)
        ctx.W = W
        if n == 0:
            ctx.save_for_backward(idx, hf2)
            return torch.zeros(0, out_dim, dtype=torch.float32, device=hf.device)
        hsel = K.gather_rows(hf2, idx)
        bd = b.detach()
        bp = bd if Np == out_dim else torch.cat([bd, bd.new_zeros(Np - out_dim)])
        out = torch.empty(n, Np, dtype=torch.float32, device=hf.device)
        K.gemm_tn(K.EPI_F32, hsel, w_rows_padded(W, Np), out, bias=bp.contiguous())
        ctx.save_for_backward(idx, hsel)
        return out[:, :out_dim]

    @staticmethod
    def backward(ctx, dout):
        idx, hsel = ctx.saved_tensors
        W = ctx.W
        B, T, D, out_dim, n = ctx.dims
        dev = dout.device
        if n == 0:
            return (torch.zeros(B, T, D, dtype=BF16, device=dev), torch.zeros_like(W),
                    torch.zeros(out_dim, dtype=torch.float32, device=dev), None)
        Kp = _pad_to(out_dim, 64)
        dop = K.cast_pad(dout.contiguous().float(), Kp)                       # [n, Kp] bf16
        dhsel = torch.empty(n, D, dtype=BF16, device=dev)
        K.gemm_tn(K.EPI_BF16, dop, w_t_cols_padded(W, Kp), dhsel)              # dout @ W
        dW = K.gemm_wgrad(dop, hsel)[:out_dim] if ctx.needs_input_grad[1] else None
        db = K.colsum_bf16(dop)[:out_dim] if ctx.needs_input_grad[2] else None
        dh = K.gather_rows(dhsel, idx, n_out=B * T, scatter=True)
        return dh.view(B, T, D), dW, db, None


class FusedLoss(torch.autograd.Function):
    """weight * mean-reduced Poisson NLL (log_input=False, full=False) or MSE; forward also produces the
    gradient, backward only scales it by the (device-resident) upstream scalar."""

    @staticmethod
    def forward(ctx, pred, obs, kind, weight, eps):
        _need_gpu(pred)
        C = pred.shape[-1]
        if pred.numel() == 0:
            # no row to reduce over (a pre-training batch without a masked region): the reference's mean-reduced
            # torch.nn.MSELoss / PoissonNLLLoss return NaN for empty inputs, and so does this
            ctx.save_for_backward(torch.zeros_like(pred, dtype=torch.float32).reshape(-1, max(C, 1)))
            ctx.weight = float(weight)
            ctx.shape = pred.shape
            ctx.empty = True
            return torch.full((), float("nan"), dtype=torch.float32, device=pred.device)
        ctx.empty = False
        p2 = pred.detach().reshape(-1, C)
        o2 = obs.detach().to(torch.float32).reshape(-1, C)
        if p2.stride(-1) != 1:
            p2 = p2.contiguous()
        if o2.stride(-1) != 1:
            o2 = o2.contiguous()
        loss, grad = K.loss_fwd(kind, p2, o2, weight=float(weight), eps=eps)
        ctx.save_for_backward(grad)
        ctx.weight = float(weight)
        ctx.shape = pred.shape
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        if ctx.empty:
            return grad.view(ctx.shape), None, None, None, None
        d = K.scale_by_scalar(grad, gout.contiguous().float(), ctx.weight)
        return d.view(ctx.shape), None, None, None, None
