"""Tensor-level wrappers of the C-ABI kernels (one Python function per entry point of include/getmi355.h).

torch is used only to allocate device buffers and to provide the stream; all arithmetic happens in
libgetmi355.so.  Every function requires CUDA(HIP) tensors and raises RuntimeError otherwise.
"""
import torch

from . import lib

EPI_BF16, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_DGELU, EPI_EMBED, EPI_F32, EPI_BIAS_GELUG, EPI_DMUL = range(8)
BF16 = torch.bfloat16
F32 = torch.float32

# bench.py sets PROFILE = [] to collect (kernel name, algorithmic flops, start event, end event) per GEMM launch,
# recorded on the stream the kernel is launched on (torch's current stream).
PROFILE = None
# when a set: only these kernel groups get events (an event pair per launch keeps consecutive kernels from overlapping
# their ramp-up / tail: ~0.9 ms per C2 step with every GEMM and attention launch bracketed, ~0.2 ms with one group)
PROFILE_ONLY = None
_EPI_NAMES = ("bf16", "bias_gelu", "bias_resid", "dgelu", "embed", "f32", "bias_gelu", "dgelu")   # 6, 7: training forms


def _prof_begin(name):
    if PROFILE is None or (PROFILE_ONLY is not None and name not in PROFILE_ONLY):
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, name, flops):
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    PROFILE.append((name, flops, e0, e1))


def _chk(*ts):
    for t in ts:
        if t is not None and not (t.is_cuda and t.is_contiguous()):
            raise RuntimeError("getmi355 kernels need contiguous tensors on the GPU (no CPU fallback exists)")


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def cast_pad(src, cp=None):
    """fp32 [R,C] -> bf16 [R,Cp] (zero padded)."""
    _chk(src)
    R, C = src.shape
    cp = C if cp is None else cp
    dst = torch.empty(R, cp, dtype=BF16, device=src.device)
    lib.call("getmi_cast_pad_f32_bf16", src, dst, R, C, cp)
    return dst


def cast_bf16(src):
    """fp32 tensor -> bf16 tensor of the same shape (numel % 4 == 0)."""
    _chk(src)
    n = src.numel()
    dst = torch.empty(src.shape, dtype=BF16, device=src.device)
    lib.call("getmi_cast_pad_f32_bf16", src, dst, 1, n, n)
    return dst


def cast_transpose(src):
    """fp32 [R,C] -> bf16 [C,R]."""
    _chk(src)
    R, C = src.shape
    dst = torch.empty(C, R, dtype=BF16, device=src.device)
    lib.call("getmi_cast_transpose_f32_bf16", src, dst, R, C)
    return dst


def qkv_bias(q_bias, v_bias):
    _chk(q_bias, v_bias)
    D = q_bias.numel()
    out = torch.empty(3 * D, dtype=F32, device=q_bias.device)
    lib.call("getmi_qkv_bias", q_bias, v_bias, out, D)
    return out


def gemm_tn(epi, A, B, out0, out1=None, bias=None, resid=None, rowscale=None, rows_per_sample=0, aux=None,
            mask=None, mask_token=None, regions_per_sample=0, M=None, dbias=None, prescaled=False, blocked=False):
    """out = epilogue(A[M,K] @ B[N,K]^T).  A, B bf16 row-major; out0/out1 preallocated (out0 may be None for the
    two-output GELU epilogues: only gelu(u) is stored).  blocked: out0 of EPI_BIAS_GELUG / aux of EPI_DMUL is a
    `blocked_rows(M)` x N buffer in the hand-over layout of getmi355.h (flags bit 1; ask gemm_tn_blocked_ok first)."""
    _chk(A, B, out0, out1, bias, resid, rowscale, aux, mask, mask_token)
    M = A.shape[0] if M is None else M
    K = A.shape[1]
    N = B.shape[0]
    assert B.shape[1] == K and A.dtype == BF16 and B.dtype == BF16
    ws = None if dbias is None else _ws(lib.query("getmi_gemm_tn_workspace_bytes", M, N), A.device)
    e0 = _prof_begin("gemm_tn_kernel<" + _EPI_NAMES[epi] + ">")
    ldo = (out0 if out0 is not None else out1).shape[-1]
    lib.call("getmi_gemm_tn", epi, A, B, M, N, K, A.stride(0), B.stride(0), out0, out1, ldo, bias,
             resid, rowscale, rows_per_sample, aux, 0 if aux is None else aux.shape[-1], mask, mask_token,
             regions_per_sample, dbias, ws, 0 if ws is None else ws.numel(), (1 if prescaled else 0) | (2 if blocked else 0))
    _prof_end(e0, "gemm_tn_kernel<" + _EPI_NAMES[epi] + ">", 2.0 * M * N * K)
    return out0


def gemm_tn_blocked_ok(M, N, K):
    """May fc1's forward hand gelu'(u) to fc2's data gradient in the blocked layout (same answer for both calls)?"""
    return bool(lib.query("getmi_gemm_tn_blocked_ok", M, N, K))


def blocked_rows(M):
    return (M + 15) // 16 * 16


def gemm_wgrad(X, Y, out=None, accumulate=False):
    """out[N1,N2] (+)= X[M,N1]^T @ Y[M,N2]  (fp32)."""
    _chk(X, Y, out)
    M, N1 = X.shape
    N2 = Y.shape[1]
    assert Y.shape[0] == M and X.dtype == BF16 and Y.dtype == BF16
    if out is None:
        out = torch.empty(N1, N2, dtype=F32, device=X.device)
        accumulate = False
    nbytes = lib.query("getmi_gemm_wgrad_workspace_bytes", M, N1, N2)
    ws = _ws(nbytes, X.device)
    e0 = _prof_begin("gemm_wgrad_kernel(+reduce_slabs)")
    lib.call("getmi_gemm_wgrad", X, Y, M, N1, N2, X.stride(0), Y.stride(0), out, int(accumulate), ws, ws.numel())
    _prof_end(e0, "gemm_wgrad_kernel(+reduce_slabs)", 2.0 * M * N1 * N2)
    return out


def gemm_wgrad_pair(Xa, Ya, Xb, Yb, out_a=None, out_b=None):
    """(Xa^T @ Ya, Xb^T @ Yb) for operands with the same number of rows, in one launch."""
    _chk(Xa, Ya, Xb, Yb, out_a, out_b)
    M = Xa.shape[0]
    assert Ya.shape[0] == M and Xb.shape[0] == M and Yb.shape[0] == M
    N1a, N2a, N1b, N2b = Xa.shape[1], Ya.shape[1], Xb.shape[1], Yb.shape[1]
    if out_a is None:
        out_a = torch.empty(N1a, N2a, dtype=F32, device=Xa.device)
    if out_b is None:
        out_b = torch.empty(N1b, N2b, dtype=F32, device=Xa.device)
    ws = _ws(lib.query("getmi_gemm_wgrad_pair_workspace_bytes", M, N1a, N2a, N1b, N2b), Xa.device)
    e0 = _prof_begin("gemm_wgrad_kernel(+reduce_slabs)")
    lib.call("getmi_gemm_wgrad_pair", Xa, Ya, N1a, N2a, Xa.stride(0), Ya.stride(0), out_a, 0, Xb, Yb, N1b, N2b, Xb.stride(0),
             Yb.stride(0), out_b, 0, M, ws, ws.numel())
    _prof_end(e0, "gemm_wgrad_kernel(+reduce_slabs)", 2.0 * M * (N1a * N2a + N1b * N2b))
    return out_a, out_b


def layernorm_fwd(x, gamma, beta, eps=1e-6):
    """x fp32 [M,D] -> (y bf16 [M,D], mean [M], rstd [M])."""
    _chk(x, gamma, beta)
    M, D = x.shape
    y = torch.empty(M, D, dtype=BF16, device=x.device)
    mean = torch.empty(M, dtype=F32, device=x.device)
    rstd = torch.empty(M, dtype=F32, device=x.device)
    e0 = _prof_begin("ln_fwd_kernel")
    lib.call("getmi_layernorm_fwd", x, gamma, beta, y, mean, rstd, M, D, float(eps))
    _prof_end(e0, "ln_fwd_kernel", float(M) * D * 6)         # algorithmic bytes: x fp32 in, y bf16 out
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dres=None, rowscale=None, rows_per_sample=0, want_dz=False,
                  want_dbias=False, dx_dtype=F32):
    """Returns (dx, dz bf16|None, dgamma, dbeta, dbias_prev|None).  dres may be fp32 or bf16; dx_dtype picks the element
    type of dx (the "bf16 gradient stream" of functional.py passes bf16 on both sides).  With a bf16 dx and no row scale
    dz IS dx (same tensor: the kernel does not write it twice)."""
    _chk(dy, x, mean, rstd, gamma, dres, rowscale)
    M, D = x.shape
    dev = x.device
    dx = torch.empty(M, D, dtype=dx_dtype, device=dev)
    alias = want_dz and dx_dtype == BF16 and rowscale is None
    dz = torch.empty(M, D, dtype=BF16, device=dev) if (want_dz and not alias) else None
    dgamma = torch.empty(D, dtype=F32, device=dev)
    dbeta = torch.empty(D, dtype=F32, device=dev)
    dbias = torch.empty(D, dtype=F32, device=dev) if want_dbias else None
    ws = _ws(lib.query("getmi_layernorm_bwd_workspace_bytes", M, D), dev)
    e0 = _prof_begin("ln_bwd_kernel")
    res_b = dres is not None and dres.dtype == BF16
    if res_b or dx_dtype == BF16:
        lib.call("getmi_layernorm_bwd_stream", dy, x, mean, rstd, gamma, dres, int(res_b), rowscale, rows_per_sample, dx,
                 int(dx_dtype == BF16), dz, dgamma, dbeta, dbias, 0, M, D, ws, ws.numel())
    else:
        lib.call("getmi_layernorm_bwd", dy, x, mean, rstd, gamma, dres, rowscale, rows_per_sample, dx, dz, dgamma, dbeta,
                 dbias, 0, M, D, ws, ws.numel())
    # (for this HBM-bound kernel the profile's second field is its ALGORITHMIC BYTES, not FLOPs: dy bf16 + x fp32 +
    #  dx (+ dres) in their stream type (+ dz bf16) per element)
    sb = 2 if dx_dtype == BF16 else 4
    rb = 0 if dres is None else (2 if res_b else 4)
    _prof_end(e0, "ln_bwd_kernel", float(M) * D * (2 + 4 + sb + rb + (2 if dz is not None else 0)))
    return dx, (dx if alias else dz), dgamma, dbeta, dbias


def attention_fwd(qkv, B, T, H, scale, want_lo=False, sample_scale=None):
    """qkv bf16 [B*T, 3*H*64] (packed [B,T,3,H,64]) -> (out bf16 [B*T, H*64], lse fp32 [B,H,T]); with want_lo also
    the rounding residual bf16(O - out) that attention_bwd takes as out_lo."""
    _chk(qkv, sample_scale)
    out = torch.empty(B * T, H * 64, dtype=BF16, device=qkv.device)
    lo = torch.empty(B * T, H * 64, dtype=BF16, device=qkv.device) if want_lo else None
    lse = torch.empty(B, H, T, dtype=F32, device=qkv.device)
    e0 = _prof_begin("attn_fwd_kernel")
    lib.call("getmi_attention_fwd", qkv, out, lo, lse, B, T, H, float(scale), sample_scale)
    _prof_end(e0, "attn_fwd_kernel", 4.0 * B * H * T * T * 64)
    return (out, lse, lo) if want_lo else (out, lse)


def attention_bwd(qkv, out, dout, lse, B, T, H, scale, want_bias=False, out_lo=None, sample_scale=None):
    """-> dqkv, or (dqkv, dq_bias [H*64], dv_bias [H*64]) with want_bias (column sums of the Q / V thirds, fp32)."""
    _chk(qkv, out, dout, lse, out_lo, sample_scale)
    dqkv = torch.empty_like(qkv)
    dqb = torch.empty(H * 64, dtype=F32, device=qkv.device) if want_bias else None
    dvb = torch.empty(H * 64, dtype=F32, device=qkv.device) if want_bias else None
    ws = _ws(lib.query("getmi_attention_bwd_workspace_bytes", B, T, H), qkv.device)
    e0 = _prof_begin("attn_bwd_kernel(+delta)")
    lib.call("getmi_attention_bwd", qkv, out, out_lo, dout, lse, dqkv, dqb, dvb, B, T, H, float(scale), sample_scale, ws,
             ws.numel())
    _prof_end(e0, "attn_bwd_kernel(+delta)", 10.0 * B * H * T * T * 64)
    return (dqkv, dqb, dvb) if want_bias else dqkv


def fill_cls(x0, cls_token, B, T, D):
    _chk(x0, cls_token)
    lib.call("getmi_fill_cls", x0, cls_token, B, T, D)


def embed_bwd_prep(dx0, mask_u8, B, N, D):
    """-> (dz bf16 [B*N,D], dcls [D], dmask_token [D], dbias [D])."""
    _chk(dx0, mask_u8)
    dev = dx0.device
    dz = torch.empty(B * N, D, dtype=BF16, device=dev)
    dcls = torch.empty(D, dtype=F32, device=dev)
    dmt = torch.empty(D, dtype=F32, device=dev)
    dbias = torch.empty(D, dtype=F32, device=dev)
    ws = _ws(lib.query("getmi_embed_bwd_workspace_bytes", B, N, D), dev)
    lib.call("getmi_embed_bwd_prep", dx0, mask_u8, dz, dcls, dmt, dbias, 0, B, N, D, ws, ws.numel())
    return dz, dcls, dmt, dbias


def head_fwd(hf, W, bias, B, N):
    _chk(hf, W, bias)
    out_dim, D = W.shape
    logits = torch.empty(B, N, out_dim, dtype=F32, device=hf.device)
    pred = torch.empty(B, N, out_dim, dtype=F32, device=hf.device)
    lib.call("getmi_head_fwd", hf, W, bias, logits, pred, B, N, D, out_dim)
    return logits, pred


def head_bwd(dpred, logits, hf, W, B, N):
    """-> (dh bf16 [B*(N+1), D], dW [out,D], db [out])."""
    _chk(dpred, logits, hf, W)
    out_dim, D = W.shape
    dev = hf.device
    dh = torch.empty(B * (N + 1), D, dtype=BF16, device=dev)
    dwb = torch.empty(out_dim * D + out_dim, dtype=F32, device=dev)
    ws = _ws(lib.query("getmi_head_bwd_workspace_bytes", B, N, D, out_dim), dev)
    lib.call("getmi_head_bwd", dpred, logits, hf, W, dh, dwb, 0, B, N, D, out_dim, ws, ws.numel())
    return dh, dwb[: out_dim * D].view(out_dim, D), dwb[out_dim * D:]


LOSS_POISSON, LOSS_MSE = 0, 1


def loss_fwd(kind, pred, obs, weight=1.0, eps=1e-8):
    """pred/obs: 2-D views [R, C] with unit column stride -> (loss scalar tensor, grad_unit like pred)."""
    assert pred.dim() == 2 and obs.dim() == 2 and pred.stride(1) == 1 and obs.stride(1) == 1
    if not (pred.is_cuda and obs.is_cuda):
        raise RuntimeError("getmi355 kernels need GPU tensors")
    R, C = pred.shape
    loss = torch.empty((), dtype=F32, device=pred.device)
    grad = torch.empty(R, C, dtype=F32, device=pred.device)
    ws = _ws(lib.query("getmi_loss_workspace_bytes"), pred.device)
    lib.call("getmi_loss_fwd", kind, pred, obs, loss, grad, R, C, pred.stride(0), obs.stride(0), C, float(weight),
             float(eps), ws, ws.numel())
    return loss, grad


def scale_by_scalar(x, scalar, weight=1.0):
    _chk(x, scalar)
    out = torch.empty_like(x)
    lib.call("getmi_scale_by_scalar", x, scalar, float(weight), out, x.numel())
    return out


def scale_cast(dx, rowscale=None, rows_per_sample=0):
    """dx fp32 [M,D] -> (dz bf16 = rowscale*dx, dbias [D] = column sums of dz)."""
    _chk(dx, rowscale)
    M, D = dx.shape
    dz = torch.empty(M, D, dtype=BF16, device=dx.device)
    dbias = torch.empty(D, dtype=F32, device=dx.device)
    ws = _ws(lib.query("getmi_scale_cast_workspace_bytes", M, D), dx.device)
    lib.call("getmi_scale_cast", dx, rowscale, rows_per_sample, dz, dbias, 0, M, D, ws, ws.numel())
    return dz, dbias


def gather_rows(src, idx, n_out=None, scatter=False):
    """gather: dst[i] = src[idx[i]]; scatter: dst[idx[i]] = src[i] (dst zero-initialised, n_out rows)."""
    _chk(src, idx)
    D = src.shape[1]
    n = idx.numel()
    if scatter:
        dst = torch.zeros(n_out, D, dtype=BF16, device=src.device)
    else:
        dst = torch.empty(n, D, dtype=BF16, device=src.device)
    lib.call("getmi_gather_rows_bf16", src, idx, dst, n, D, int(scatter))
    return dst


def mask_compact(mask_u8, B, N):
    """mask uint8 [B*N] -> (token_rows int64 [B*N], region_rows int64 [B*N], count int32 [1]); only the first
    count[0] entries of the row lists are defined.  No host synchronisation."""
    _chk(mask_u8)
    dev = mask_u8.device
    tok = torch.empty(B * N, dtype=torch.int64, device=dev)
    reg = torch.empty(B * N, dtype=torch.int64, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    lib.call("getmi_mask_compact", mask_u8, B, N, tok, reg, cnt)
    return tok, reg, cnt


def colsum_bf16(X, M=None):
    """Column sums of a bf16 matrix (row stride may exceed the column count: column slices are fine)."""
    if not (X.is_cuda and X.stride(1) == 1):
        raise RuntimeError("getmi355 kernels need GPU tensors with unit column stride")
    M = X.shape[0] if M is None else M
    N = X.shape[1]
    out = torch.empty(N, dtype=F32, device=X.device)
    ws = _ws(lib.query("getmi_colsum_workspace_bytes", N), X.device)
    lib.call("getmi_colsum_bf16", X, out, 0, M, N, X.stride(0), ws, ws.numel())
    return out


def scale_cast_noalloc(dx, rowscale=None, rows_per_sample=0):
    """scale_cast through the entry point that uses the library's built-in scratch buffer (no workspace argument; for
    callers that cannot allocate).  The two result tensors here are the caller's outputs, as in the C ABI."""
    _chk(dx, rowscale)
    M, D = dx.shape
    dz = torch.empty(M, D, dtype=BF16, device=dx.device)
    dbias = torch.empty(D, dtype=F32, device=dx.device)
    lib.call("getmi_scale_cast_noalloc", dx, rowscale, rows_per_sample, dz, dbias, 0, M, D)
    return dz, dbias
