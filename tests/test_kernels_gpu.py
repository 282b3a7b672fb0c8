"""GPU: every C-ABI kernel against a plain torch fp32 restatement of the same op on the same (bf16-rounded)
inputs.  Tolerances are stated per test: bf16 outputs carry one rounding (rel 2^-8 = 3.9e-3)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from get_model_amd import kernels
    return kernels


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def bf(x):
    return x.to(torch.bfloat16)


def close(a, b, rtol, atol, what=""):
    a = a.float()
    b = b.float()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol).sum().item()
    assert bad == 0, f"{what}: {bad}/{a.numel()} mismatches, max err {err.max().item():.4e}, ref max {b.abs().max().item():.3e}"


def test_cast_and_transpose(K):
    x = rnd(37, 283, seed=1)
    y = K.cast_pad(x, 320)
    assert y.shape == (37, 320)
    assert torch.equal(y[:, :283], bf(x)) and (y[:, 283:] == 0).all()
    w = rnd(96, 200, seed=2)
    assert torch.equal(K.cast_transpose(w), bf(w).t().contiguous())
    v = rnd(3 * 768, seed=3)
    assert torch.equal(K.cast_bf16(v), bf(v))
    qb = K.qkv_bias(v[:768].contiguous(), v[768:1536].contiguous())
    assert torch.equal(qb[:768], v[:768]) and (qb[768:1536] == 0).all() and torch.equal(qb[1536:], v[768:1536])


@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (201, 768, 768), (804, 2304, 768), (333, 3072, 768),
                                    (402, 768, 3072), (1000, 768, 320),
                                    # M >= 2048 and N >= 512 select the 256x256 deep-pipelined kernel
                                    (2048, 512, 64), (2304, 768, 768), (2500, 2304, 128), (2049, 768, 3072),
                                    (4000, 516, 320), (6030, 3072, 768)])
def test_gemm_tn_plain_asymmetric(K, M, N, K_):
    # asymmetric integer-valued operands: exact in bf16/fp32, catches any row/col or k-order mix-up
    A = ((torch.arange(M * K_).reshape(M, K_) * 7 + 3) % 13 - 6).float().cuda()
    B = ((torch.arange(N * K_).reshape(N, K_) * 5 + 1) % 11 - 5).float().cuda()
    ref = A @ B.t()
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    K.gemm_tn(K.EPI_F32, bf(A), bf(B), out)
    assert torch.equal(out, ref)
    outb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    K.gemm_tn(K.EPI_BF16, bf(A), bf(B), outb)
    assert torch.equal(outb, bf(ref))


@pytest.mark.parametrize("M,N,K_", [(51456, 768, 128), (51456, 2304, 64), (30000, 768, 64), (9999, 768, 64)])
def test_gemm_tn_split_launches(K, M, N, K_):
    """Row-split launches with mixed tile heights (plan_rows in gemm.hip): bit-exact integer product for the bf16 output,
    and the fp32 residual epilogue with a per-sample row scale whose sample boundaries do not line up with the cut."""
    A = ((torch.arange(M * K_, device="cuda").reshape(M, K_) * 7 + 3) % 13 - 6).float()
    B = ((torch.arange(N * K_, device="cuda").reshape(N, K_) * 5 + 1) % 11 - 5).float()
    ref = A @ B.t()
    outb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    K.gemm_tn(K.EPI_BF16, bf(A), bf(B), outb)
    assert torch.equal(outb, bf(ref))
    T = 201 if M % 201 == 0 else 3
    resid = rnd(M, N, seed=7)
    bias = rnd(N, scale=0.1, seed=6)
    rs = ((torch.arange(M // T, device="cuda") % 3).float() + 0.5)
    x = torch.empty(M, N, dtype=torch.float32, device="cuda")
    K.gemm_tn(K.EPI_BIAS_RESID, bf(A), bf(B), x, bias=bias, resid=resid, rowscale=rs, rows_per_sample=T)
    refx = resid + (ref + bias) * rs.repeat_interleave(T)[:M, None]
    close(x, refx, 1e-5, 1e-3, "resid-split")


@pytest.mark.parametrize("M", [402, 2211])
def test_gemm_tn_epilogues(K, M):
    N, K_, T = 768, 768, 201
    A, W = bf(rnd(M, K_, seed=4)), bf(rnd(N, K_, scale=0.05, seed=5))
    bias = rnd(N, scale=0.1, seed=6)
    ref = A.float() @ W.float().t() + bias
    # bias -> bf16
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    K.gemm_tn(K.EPI_BF16, A, W, o, bias=bias)
    close(o, ref, 4e-3, 1e-3, "bias")
    # bias + gelu
    u = torch.empty_like(o)
    g = torch.empty_like(o)
    K.gemm_tn(K.EPI_BIAS_GELU, A, W, u, out1=g, bias=bias)
    close(u, ref, 4e-3, 1e-3, "gelu.u")
    close(g, F.gelu(ref), 4e-3, 1e-3, "gelu.g")
    # bias + residual + per-sample scale (fp32 out)
    resid = rnd(M, N, seed=7)
    rs = (torch.arange(M // T, device="cuda") % 2).float() / 0.9
    x = torch.empty(M, N, dtype=torch.float32, device="cuda")
    K.gemm_tn(K.EPI_BIAS_RESID, A, W, x, bias=bias, resid=resid, rowscale=rs, rows_per_sample=T)
    refx = resid + ref * rs.repeat_interleave(T)[:, None]
    close(x, refx, 1e-5, 1e-4, "resid")
    K.gemm_tn(K.EPI_BIAS_RESID, A, W, x, bias=bias, resid=resid)
    close(x, resid + ref, 1e-5, 1e-4, "resid-noscale")
    # dgelu
    pre = bf(rnd(M, N, seed=8))
    d = torch.empty_like(o)
    dbias = torch.empty(N, device="cuda")
    K.gemm_tn(K.EPI_DGELU, A, W, d, aux=pre, dbias=dbias)
    xp = pre.float().requires_grad_(True)
    F.gelu(xp).sum().backward()
    refd = (A.float() @ W.float().t()) * xp.grad
    close(d, refd, 4e-3, 2e-3, "dgelu")
    close(dbias, refd.sum(0), 1e-3, 2e-3 * math.sqrt(M), "dgelu.colsum")
    # training forms: forward stores gelu'(u) next to gelu(u); backward multiplies by the stored derivative
    gp = torch.empty_like(o)
    g2 = torch.empty_like(o)
    K.gemm_tn(K.EPI_BIAS_GELUG, A, W, gp, out1=g2, bias=bias)
    assert torch.equal(g2, g)
    xr = ref.clone().requires_grad_(True)
    F.gelu(xr).sum().backward()
    close(gp, xr.grad, 4e-3, 1e-3, "gelug.derivative")
    d2 = torch.empty_like(o)
    dbias2 = torch.empty(N, device="cuda")
    K.gemm_tn(K.EPI_DMUL, A, W, d2, aux=pre, dbias=dbias2)
    refm = (A.float() @ W.float().t()) * pre.float()
    close(d2, refm, 4e-3, 2e-3, "dmul")
    close(dbias2, refm.sum(0), 1e-3, 2e-3 * math.sqrt(M), "dmul.colsum")


def test_gemm_tn_embed(K):
    B_, Nr, D, Kp = 3, 50, 768, 320
    A = bf(rnd(B_ * Nr, Kp, seed=9))
    W = bf(rnd(D, Kp, scale=0.05, seed=10))
    bias = rnd(D, scale=0.1, seed=11)
    mt = rnd(D, seed=12)
    mask = (torch.rand(B_ * Nr, generator=torch.Generator().manual_seed(1)) < 0.5).to(torch.uint8).cuda()
    x0 = torch.full((B_ * (Nr + 1), D), 777.0, device="cuda")
    K.gemm_tn(K.EPI_EMBED, A, W, x0, bias=bias, mask=mask, mask_token=mt, regions_per_sample=Nr)
    ref = A.float() @ W.float().t() + bias
    ref = torch.where(mask.bool()[:, None], mt[None, :], ref)
    x0 = x0.view(B_, Nr + 1, D)
    assert (x0[:, 0] == 777.0).all()
    close(x0[:, 1:].reshape(-1, D), ref, 1e-5, 1e-4, "embed")
    cls = rnd(D, seed=13)
    K.fill_cls(x0, cls, B_, Nr + 1, D)
    assert torch.equal(x0[:, 0], cls.expand(B_, D))


@pytest.mark.parametrize("M,N1,N2", [(64, 128, 128), (201, 768, 768), (1005, 768, 2304), (804, 3072, 768),
                                     (130, 768, 320), (4000, 288, 768),
                                     # M >= 4096 with N1, N2 multiples of 256: the 256x256 deep-pipelined kernel
                                     (4096, 256, 256), (4100, 768, 768), (5001, 256, 512), (8200, 768, 2304),
                                     (12345, 3072, 768)])
def test_gemm_wgrad_asymmetric(K, M, N1, N2):
    X = ((torch.arange(M * N1).reshape(M, N1) * 7 + 3) % 5 - 2).float().cuda()
    Y = ((torch.arange(M * N2).reshape(M, N2) * 3 + 1) % 7 - 3).float().cuda()
    ref = X.t() @ Y
    out = K.gemm_wgrad(bf(X), bf(Y))
    assert torch.equal(out, ref), f"max err {(out - ref).abs().max().item()}"
    out2 = K.gemm_wgrad(bf(X), bf(Y), out=out.clone(), accumulate=True)
    assert torch.equal(out2, 2 * ref)


@pytest.mark.parametrize("M", [1, 7, 804, 5000])
def test_layernorm_fwd_bwd(K, M):
    D, T = 768, 201
    x = rnd(M, D, scale=2.0, seed=14) + 0.3
    gamma = 1 + rnd(D, scale=0.1, seed=15)
    beta = rnd(D, scale=0.1, seed=16)
    y, mean, rstd = K.layernorm_fwd(x, gamma, beta, 1e-6)
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (D,), gr, br, 1e-6)
    close(y, ref, 4e-3, 1e-3, "ln.y")
    close(mean, x.mean(1), 1e-5, 1e-5, "ln.mean")
    dy = bf(rnd(M, D, seed=17))
    dres = rnd(M, D, seed=18)
    nb = (M + T - 1) // T
    rs = (torch.arange(nb, device="cuda") % 3).float() / 0.9
    ref.backward(dy.float())
    dx, dz, dg, db, dbias = K.layernorm_bwd(dy, x, mean, rstd, gamma, dres=dres, rowscale=rs, rows_per_sample=T,
                                            want_dz=True, want_dbias=True)
    refdx = xr.grad + dres
    close(dx, refdx, 1e-4, 1e-4, "ln.dx")
    refdz = refdx * rs.repeat_interleave(T)[:M, None]
    close(dz, refdz, 4e-3, 1e-3, "ln.dz")
    close(dg, gr.grad, 1e-3, 1e-3 * math.sqrt(M), "ln.dgamma")
    close(db, br.grad, 1e-3, 1e-3 * math.sqrt(M), "ln.dbeta")
    close(dbias, refdz.sum(0), 1e-3, 1e-3 * math.sqrt(M), "ln.dbias_prev")
    dx2, dz2, _, _, _ = K.layernorm_bwd(dy, x, mean, rstd, gamma)
    close(dx2, xr.grad, 1e-4, 1e-4, "ln.dx-nores")
    assert dz2 is None


def attn_ref(qkv, B, T, H, scale):
    q, k, v = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q * scale) @ k.transpose(-2, -1)
    p = s.softmax(-1)
    o = (p @ v).transpose(1, 2).reshape(B * T, H * 64)
    return o, torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,T,H", [(2, 201, 12), (1, 32, 1), (3, 37, 2), (1, 128, 3), (2, 224, 2), (2, 225, 1), (2, 256, 2),
                                   (1, 257, 2), (1, 449, 2), (1, 2001, 2), (1, 513, 1)])
def test_attention_fwd_bwd(K, B, T, H):
    scale = 0.125
    qkv = bf(rnd(B * T, 3 * H * 64, scale=1.5, seed=19))
    out, lse = K.attention_fwd(qkv, B, T, H, scale)
    qr = qkv.float().requires_grad_(True)
    ref, ref_lse = attn_ref(qr, B, T, H, scale)
    close(out, ref, 8e-3, 4e-3, "attn.out")
    close(lse, ref_lse, 1e-4, 2e-3, "attn.lse")
    dout = bf(rnd(B * T, H * 64, seed=20))
    ref.backward(dout.float())
    dqkv = K.attention_bwd(qkv, out, dout, lse, B, T, H, scale)
    g = qr.grad
    # normalised max-abs (gradients span orders of magnitude): 1.5 % of the tensor's max
    err = (dqkv.float() - g).abs().max().item()
    assert err <= 1.5e-2 * g.abs().max().item(), f"attn.dqkv err {err:.3e} vs max {g.abs().max().item():.3e}"
    cos = F.cosine_similarity(dqkv.float().flatten(), g.flatten(), dim=0).item()
    assert cos > 0.9995, f"attn.dqkv cosine {cos}"
    # fused bias gradients: column sums of the Q / V thirds over all tokens, from the fp32 accumulators; the same
    # dqkv comes back, and the sums match the fp32 reference (scaled by the size of the summands: dq sums cancel)
    dqkv2, dqb, dvb = K.attention_bwd(qkv, out, dout, lse, B, T, H, scale, want_bias=True)
    assert torch.equal(dqkv2, dqkv)
    D = H * 64
    for name, got, third in (("dq_bias", dqb, g[:, :D]), ("dv_bias", dvb, g[:, 2 * D:])):
        ref_sum = third.sum(0)
        tol = 4e-3 * third.abs().sum(0).max().item()
        err = (got - ref_sum).abs().max().item()
        assert err <= tol, f"attn.{name} err {err:.3e} > {tol:.3e}"


@pytest.mark.parametrize("B,T,H", [(2, 201, 12), (1, 32, 1), (3, 37, 2), (2, 70, 1), (1, 97, 2), (2, 256, 2), (1, 300, 1),
                                   (1, 1000, 2)])
def test_attention_bwd_with_output_residual(K, B, T, H):
    """out_lo path (delta from O + its rounding residual): dqkv as without it to rounding, and the q_bias sums -- a pure
    cancellation sum -- an order of magnitude closer to the fp32 reference than the 4e-3 bar of the plain path."""
    scale = 0.125
    qkv = bf(rnd(B * T, 3 * H * 64, scale=1.5, seed=29))
    out, lse, lo = K.attention_fwd(qkv, B, T, H, scale, want_lo=True)
    out_p, lse_p = K.attention_fwd(qkv, B, T, H, scale)
    assert torch.equal(out, out_p) and torch.equal(lse, lse_p)
    qr = qkv.float().requires_grad_(True)
    ref, _ = attn_ref(qr, B, T, H, scale)
    # lo is the rounding residual of the kernel's own fp32 output (which carries the bf16 rounding of P): within half an
    # ulp of `out`, and out + lo no further from the reference than out
    assert (lo.float().abs() <= out.float().abs() * 2.0 ** -8 + 1e-30).all()
    close(out.float() + lo.float(), ref, 8e-3, 4e-3, "attn.out+lo")
    dout = bf(rnd(B * T, H * 64, seed=30))
    ref.backward(dout.float())
    g = qr.grad
    dqkv, dqb, dvb = K.attention_bwd(qkv, out, dout, lse, B, T, H, scale, want_bias=True, out_lo=lo)
    err = (dqkv.float() - g).abs().max().item()
    assert err <= 1.5e-2 * g.abs().max().item(), f"attn.dqkv(lo) err {err:.3e}"
    assert F.cosine_similarity(dqkv.float().flatten(), g.flatten(), dim=0).item() > 0.9995
    D = H * 64
    for name, got, third, rel in (("dq_bias", dqb, g[:, :D], 1e-3), ("dv_bias", dvb, g[:, 2 * D:], 4e-3)):
        tol = rel * third.abs().sum(0).max().item()
        err = (got - third.sum(0)).abs().max().item()
        assert err <= tol, f"attn.{name}(lo) err {err:.3e} > {tol:.3e}"


def test_attention_online_softmax_rescale(K):
    # a late key with a huge score forces the running-max rescale branch in a late chunk
    B, T, H = 1, 201, 1
    qkv = bf(rnd(B * T, 3 * 64, scale=0.5, seed=21)).view(T, 3, 64).clone()
    qkv[190, 1] = qkv[5, 0] * 40          # key 190 aligned with query 5
    qkv = qkv.view(T, 192).contiguous()
    out, lse = K.attention_fwd(qkv, B, T, H, 0.125)
    ref, ref_lse = attn_ref(qkv, B, T, H, 0.125)
    close(out, ref, 8e-3, 4e-3, "attn.out(rescale)")
    close(lse, ref_lse, 1e-4, 2e-3, "attn.lse(rescale)")


def test_head_and_loss(K):
    B_, N, D = 3, 50, 768
    for out_dim in (1, 2):
        hf = bf(rnd(B_ * (N + 1), D, seed=22))
        W = rnd(out_dim, D, scale=0.05, seed=23)
        b = rnd(out_dim, scale=0.1, seed=24) + 0.5
        logits, pred = K.head_fwd(hf, W, b, B_, N)
        hr = hf.float().view(B_, N + 1, D).requires_grad_(True)
        Wr = W.clone().requires_grad_(True)
        br = b.clone().requires_grad_(True)
        refl = F.linear(hr[:, 1:], Wr, br)
        refp = F.softplus(refl)
        close(logits, refl, 1e-4, 1e-4, "head.logits")
        close(pred, refp, 1e-4, 1e-4, "head.pred")
        obs = rnd(B_, N, out_dim, seed=25).abs()
        loss, gu = K.loss_fwd(K.LOSS_POISSON, pred.view(-1, out_dim), obs.view(-1, out_dim))
        refloss = F.poisson_nll_loss(refp, obs, log_input=False)
        close(loss, refloss, 1e-5, 1e-6, "poisson")
        refloss.backward()
        dpred = K.scale_by_scalar(gu, torch.ones((), device="cuda"), 1.0).view(B_, N, out_dim)
        dh, dW, db = K.head_bwd(dpred, logits, hf, W, B_, N)
        close(dh.view(B_, N + 1, D), hr.grad, 8e-3, 1e-7, "head.dh")
        assert (dh.view(B_, N + 1, D)[:, 0] == 0).all()
        close(dW, Wr.grad, 1e-3, 1e-5, "head.dW")
        close(db, br.grad, 1e-3, 1e-6, "head.db")
    # MSE with padded prediction rows
    p = rnd(40, 288, seed=26)
    o = rnd(40, 283, seed=27)
    loss, gu = K.loss_fwd(K.LOSS_MSE, p[:, :283], o, weight=2.0)
    pr = p[:, :283].clone().requires_grad_(True)
    ref = F.mse_loss(pr, o)
    ref.backward()
    close(loss, 2.0 * ref, 1e-5, 1e-6, "mse")
    close(gu, pr.grad, 1e-5, 1e-8, "mse.grad")


def test_embed_bwd_prep_gather_colsum(K):
    B_, N, D = 3, 40, 768
    T = N + 1
    dx0 = rnd(B_ * T, D, seed=28)
    mask = (torch.rand(B_ * N, generator=torch.Generator().manual_seed(2)) < 0.5).cuda()
    dz, dcls, dmt, dbias = K.embed_bwd_prep(dx0, mask.to(torch.uint8), B_, N, D)
    v = dx0.view(B_, T, D)
    reg = v[:, 1:].reshape(-1, D)
    close(dcls, v[:, 0].sum(0), 1e-5, 1e-5, "dcls")
    close(dmt, reg[mask].sum(0), 1e-5, 1e-4, "dmask_token")
    close(dbias, reg[~mask].sum(0), 1e-5, 1e-4, "dbias")
    assert torch.equal(dz, bf(torch.where(mask[:, None], torch.zeros_like(reg), reg)))
    dz2, _, dmt2, dbias2 = K.embed_bwd_prep(dx0, None, B_, N, D)
    assert torch.equal(dz2, bf(reg)) and (dmt2 == 0).all()
    close(dbias2, reg.sum(0), 1e-5, 1e-4, "dbias-nomask")
    # gather / scatter
    src = bf(rnd(100, D, seed=29))
    idx = torch.tensor([5, 0, 99, 42, 7], device="cuda")
    gth = K.gather_rows(src, idx)
    assert torch.equal(gth, src[idx])
    sc = K.gather_rows(gth, idx, n_out=100, scatter=True)
    ref = torch.zeros_like(src)
    ref[idx] = src[idx]
    assert torch.equal(sc, ref)
    # column sums
    X = bf(rnd(777, 2304, seed=30))
    close(K.colsum_bf16(X), X.float().sum(0), 1e-4, 1e-3, "colsum")


def test_errors_are_loud(K):
    A = torch.zeros(8, 100, dtype=torch.bfloat16, device="cuda")   # K=100 not a multiple of 64
    out = torch.empty(8, 8, dtype=torch.float32, device="cuda")
    with pytest.raises(RuntimeError, match="multiple of 64"):
        K.gemm_tn(K.EPI_F32, A, A[:8].contiguous(), out)
    with pytest.raises(RuntimeError):
        K.layernorm_fwd(torch.zeros(4, 100, device="cuda"), torch.zeros(100, device="cuda"), torch.zeros(100, device="cuda"))
    with pytest.raises(RuntimeError, match="GPU"):
        K.cast_pad(torch.zeros(4, 8))


def test_cu_limit_shrinks_grids_without_changing_results(K):
    """getmi_set_cu_limit: the persistent GEMM and the weight-gradient GEMM size their grids for fewer CUs (used
    when RCCL's workgroups share the GPU); same results (wgrad up to the fp32 summation order of its M-split)."""
    from get_model_amd import lib
    M, N, Kd = 4096 + 200, 768, 512
    g = torch.Generator(device="cpu").manual_seed(5)
    A = (torch.randn(M, Kd, generator=g) * 0.5).bfloat16().cuda()
    W = (torch.randn(N, Kd, generator=g) * 0.5).bfloat16().cuda()
    Y = (torch.randn(M, N, generator=g) * 0.5).bfloat16().cuda()
    total = lib.query("getmi_cu_count")
    try:
        out0 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        K.gemm_tn(K.EPI_BF16, A, W, out0)
        w0 = K.gemm_wgrad(Y, A)
        assert lib.query("getmi_set_cu_limit", total - 56) == 0
        assert lib.query("getmi_cu_count") == total - 56
        out1 = torch.empty_like(out0)
        K.gemm_tn(K.EPI_BF16, A, W, out1)
        w1 = K.gemm_wgrad(Y, A)
        torch.cuda.synchronize()
        assert torch.equal(out0, out1)
        torch.testing.assert_close(w1, w0, rtol=1e-4, atol=1e-3)
    finally:
        lib.query("getmi_set_cu_limit", 0)
    assert lib.query("getmi_cu_count") == total


def test_gemm_wgrad_pair_equals_two_calls(K):
    """Two weight gradients over the same rows in one launch (big-tile path) and the two-launch fallback."""
    for M, (N1a, N2a), (N1b, N2b) in ((4096 + 77, (256, 512), (768, 256)), (300, (128, 64), (64, 192))):
        g = torch.Generator(device="cpu").manual_seed(M)
        Xa, Ya = bf(torch.randn(M, N1a, generator=g).cuda()), bf(torch.randn(M, N2a, generator=g).cuda())
        Xb, Yb = bf(torch.randn(M, N1b, generator=g).cuda()), bf(torch.randn(M, N2b, generator=g).cuda())
        oa, ob = K.gemm_wgrad_pair(Xa, Ya, Xb, Yb)
        ra, rb_ = Xa.float().t() @ Ya.float(), Xb.float().t() @ Yb.float()
        close(oa, ra, 1e-4, 1e-3 * math.sqrt(M), "pair.a")
        close(ob, rb_, 1e-4, 1e-3 * math.sqrt(M), "pair.b")


def test_cu_limit_is_read_once_per_call_and_workspaces_stay_valid(K):
    """getmi_set_cu_limit (ddp: CUs left to RCCL while an all-reduce overlaps backward): results do not depend on the
    limit, and a workspace sized under one limit is large enough under any other (queries answer for all CUs)."""
    from get_model_amd import lib
    M, N1, N2 = 8192 + 40, 768, 768
    g = torch.Generator(device="cpu").manual_seed(9)
    X = (torch.randn(M, N1, generator=g) * 0.5).bfloat16().cuda()
    Y = (torch.randn(M, N2, generator=g) * 0.5).bfloat16().cuda()
    total = lib.query("getmi_cu_count")
    ref = K.gemm_wgrad(X, Y)
    try:
        sizes = set()
        for lim in (0, total - 16, total // 2, 64):
            lib.query("getmi_set_cu_limit", lim)
            assert lib.query("getmi_cu_count") == (lim if lim else total)
            sizes.add(lib.query("getmi_gemm_wgrad_workspace_bytes", M, N1, N2))
            out = K.gemm_wgrad(X, Y)
            torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-3)
            # fused column sums of the dGELU-form GEMM: the partial-row layout follows the tile choice of THIS call
            A = (torch.randn(4096, 768, generator=g) * 0.5).bfloat16().cuda()
            W = (torch.randn(1024, 768, generator=g) * 0.05).bfloat16().cuda()
            aux = torch.ones(4096, 1024, dtype=torch.bfloat16, device="cuda")
            o = torch.empty(4096, 1024, dtype=torch.bfloat16, device="cuda")
            db = torch.empty(1024, device="cuda")
            K.gemm_tn(K.EPI_DMUL, A, W, o, aux=aux, dbias=db)
            torch.testing.assert_close(db, (A.float() @ W.float().t()).sum(0), rtol=2e-2, atol=0.5)
        assert len(sizes) == 1
    finally:
        lib.query("getmi_set_cu_limit", 0)


def test_second_generation_gemm_kernel_stays_correct():
    """gemm_tn2_kernel (two workgroups per CU; off by default) is selected per process by
    GETMI_GEMM_GEN=2: the GEMM tests of this file are re-run under it in a child process (bit-exact integer products,
    every epilogue, the split-launch shapes), and once more with the pre-round-2 tile order (GETMI_GEMM_GROUP=0)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (round 3: + the asymmetric 208-row tile of the persistent kernel, off by default: GETMI_GEMM_BM=208)
    for extra in ({"GETMI_GEMM_GEN": "2"}, {"GETMI_GEMM_GEN": "1", "GETMI_GEMM_GROUP": "0"}, {"GETMI_GEMM_BM": "208"}):
        env = dict(os.environ)
        env.update(extra)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_kernels_gpu.py"), "-q", "-x",
                            "-m", "gpu", "-k", "gemm_tn", "-p", "no:cacheprovider"], env=env, cwd=root, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, (extra, r.stdout[-2000:], r.stderr[-1000:])


def test_staged_attention_backward_stays_correct():
    """The staged attention backward (attn_bwd_kernel: key blocks of 256, everything staged before the first slice) is kept
    behind GETMI_ATTN_STREAM=0; the default is the streamed kernel.  The attention tests of this file are re-run under the
    switch in a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["GETMI_ATTN_STREAM"] = "0"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_kernels_gpu.py"), "-q", "-x",
                        "-m", "gpu", "-k", "test_attention", "-p", "no:cacheprovider"], env=env, cwd=root, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-1000:])


# ---- DropPath fold (round 3): the per-sample factor lives in the saved activations --------------------------------
@pytest.mark.parametrize("B,T,H", [(4, 201, 3), (3, 37, 2), (3, 300, 1)])
def test_attention_with_sample_scale(K, B, T, H):
    """out = s_b * Attn(q, k, v) with s_b in {0, 1/keep}: forward rows, the residual, and the backward given
    dout = dL/d(out) against autograd of the same expression; samples with s_b = 0 give exact zeros everywhere."""
    scale = 0.125
    qkv = bf(rnd(B * T, 3 * H * 64, scale=1.5, seed=41))
    sc = torch.tensor([1.0 / 0.9, 0.0, 1.0 / 0.9, 1.0 / 0.9][:B], device="cuda")
    out, lse, lo = K.attention_fwd(qkv, B, T, H, scale, want_lo=True, sample_scale=sc)
    qr = qkv.float().requires_grad_(True)
    o_ref, _ = attn_ref(qr, B, T, H, scale)
    ref = o_ref * sc.repeat_interleave(T).unsqueeze(1)
    close(out.float() + lo.float(), ref, 8e-3, 4e-3, "attn.out(scaled)")
    assert (out.view(B, T, -1)[1] == 0).all() and (lo.view(B, T, -1)[1] == 0).all()
    dout = bf(rnd(B * T, H * 64, seed=42))
    ref.backward(dout.float())
    g = qr.grad
    dqkv, dqb, dvb = K.attention_bwd(qkv, out, dout, lse, B, T, H, scale, want_bias=True, out_lo=lo, sample_scale=sc)
    assert (dqkv.view(B, T, -1)[1] == 0).all()
    err = (dqkv.float() - g).abs().max().item()
    assert err <= 1.5e-2 * g.abs().max().item(), f"attn.dqkv(scaled) err {err:.3e}"
    assert F.cosine_similarity(dqkv.float().flatten(), g.flatten(), dim=0).item() > 0.9995
    D = H * 64
    for name, got, third, rel in (("dq_bias", dqb, g[:, :D], 1e-3), ("dv_bias", dvb, g[:, 2 * D:], 4e-3)):
        tol = rel * third.abs().sum(0).max().item()
        e = (got - third.sum(0)).abs().max().item()
        assert e <= tol, f"attn.{name}(scaled) err {e:.3e} > {tol:.3e}"


@pytest.mark.parametrize("M,T", [(804, 201), (4020, 201), (296, 37)])
def test_gemm_epilogues_with_folded_row_scale(K, M, T):
    """Epilogue 6 with a row scale (both outputs x s_b) and epilogue 2 with the `prescaled` flag
    (out = resid + acc + s_b * bias), small-tile and 256 x 256 persistent kernels (M >= 2048)."""
    D, Fh = 768, 3072
    Bn = M // T
    sc = ((torch.arange(Bn) % 3 != 1).float() / 0.9).cuda()
    srow = sc.repeat_interleave(T).unsqueeze(1)
    x = bf(rnd(M, D, seed=51))
    w1 = bf(rnd(Fh, D, scale=0.05, seed=52))
    b1 = rnd(Fh, scale=0.1, seed=53)
    u = torch.empty(M, Fh, dtype=torch.bfloat16, device="cuda")
    g = torch.empty_like(u)
    K.gemm_tn(K.EPI_BIAS_GELUG, x, w1, u, out1=g, bias=b1, rowscale=sc, rows_per_sample=T)
    pre = (x.float() @ w1.float().t() + b1).requires_grad_(True)
    gref = F.gelu(pre)
    gref.sum().backward()
    close(g, gref * srow, 8e-3, 2e-3, "geluG.g(scaled)")
    close(u, pre.grad * srow, 8e-3, 2e-3, "geluG.g'(scaled)")
    # the same call without out0 (inference) stores the same g
    g2 = torch.empty_like(g)
    K.gemm_tn(K.EPI_BIAS_GELUG, x, w1, None, out1=g2, bias=b1, rowscale=sc, rows_per_sample=T)
    assert torch.equal(g2, g)
    # residual epilogue: A already carries the factor
    w2 = bf(rnd(D, Fh, scale=0.03, seed=54))
    b2 = rnd(D, scale=0.1, seed=55)
    resid = rnd(M, D, seed=56)
    out = torch.empty(M, D, device="cuda")
    K.gemm_tn(K.EPI_BIAS_RESID, g, w2, out, bias=b2, resid=resid, rowscale=sc, rows_per_sample=T, prescaled=True)
    ref = resid + g.float() @ w2.float().t() + srow * b2
    close(out, ref, 2e-3, 2e-3, "resid(prescaled)")
    out2 = torch.empty(M, D, device="cuda")
    K.gemm_tn(K.EPI_BIAS_RESID, g, w2, out2, bias=b2, resid=resid, rowscale=sc, rows_per_sample=T)
    close(out2, resid + srow * (g.float() @ w2.float().t() + b2), 2e-3, 2e-3, "resid(unfolded)")


def test_layernorm_bwd_bias_sums_without_the_scaled_copy(K):
    """bf16 stream forms: dres / dx in bf16, and dbias_prev = column sums of rowscale * dx WITHOUT a dz output."""
    M, D, T = 603, 768, 201
    x = rnd(M, D, seed=61)
    gamma = rnd(D, scale=0.2, seed=62) + 1.0
    beta = rnd(D, scale=0.1, seed=63)
    y, mean, rstd = K.layernorm_fwd(x, gamma, beta)
    dy = bf(rnd(M, D, seed=64))
    dres = bf(rnd(M, D, seed=65))
    sc = torch.tensor([1.25, 0.0, 1.25], device="cuda")
    dx, dz, dg, dbt, dbias = K.layernorm_bwd(dy, x, mean, rstd, gamma, dres=dres, rowscale=sc, rows_per_sample=T,
                                             want_dz=False, want_dbias=True, dx_dtype=torch.bfloat16)
    assert dz is None and dx.dtype == torch.bfloat16
    dx_f, dz_f, dg_f, dbt_f, dbias_f = K.layernorm_bwd(dy, x, mean, rstd, gamma, dres=dres.float(), rowscale=sc,
                                                       rows_per_sample=T, want_dz=True, want_dbias=True)
    close(dx, dx_f, 4e-3, 1e-4, "ln.dx(bf16 stream)")
    close(dbias, dbias_f, 1e-4, 1e-3, "ln.dbias(no dz)")
    close(dg, dg_f, 1e-5, 1e-4, "ln.dgamma")
    # no row scale + bf16 dx: dz is dx itself
    dx2, dz2, _, _, db2 = K.layernorm_bwd(dy, x, mean, rstd, gamma, dres=dres, want_dz=True, want_dbias=True,
                                          dx_dtype=torch.bfloat16)
    assert dz2 is dx2
    close(db2, dx2.float().sum(0), 2e-3, 0.5, "ln.dbias(alias)")      # (sums are taken before the bf16 rounding of dx)


def _deblock(ub, M, N):
    """Row-major view of a tensor in the blocked hand-over layout of getmi355.h (flags bit 1 of getmi_gemm_tn)."""
    nrb, ncb = (M + 15) // 16, N // 64
    v = ub.view(nrb, ncb, 2, 4, 16, 2, 4)                    # [rb][cb][p][c][r][which piece of the pair][4 columns]
    # column = 64cb + 32p + 16*piece + 4c + j, row = 16rb + r
    rm = v.permute(0, 4, 1, 2, 5, 3, 6).reshape(nrb * 16, N)  # [rb][r][cb][p][piece][c][j]
    return rm[:M]


@pytest.mark.parametrize("M,N,Kd,T", [(2060, 512, 128, 103), (4020, 3072, 768, 201), (51456, 768, 256, 201)])
def test_blocked_handover_of_the_gelu_derivative(K, M, N, Kd, T):
    """flags bit 1: epilogue 6 writes gelu'(u) in the blocked layout and epilogue 7 reads it back -- bit for bit the
    row-major results (M not a multiple of 16, 224- and 256-row tiles, with the folded row scale)."""
    assert K.gemm_tn_blocked_ok(M, N, Kd)
    assert not K.gemm_tn_blocked_ok(300, N, Kd) and not K.gemm_tn_blocked_ok(M, N + 32, Kd)
    Bn = M // T
    sc = ((torch.arange(Bn) % 4 != 2).float() / 0.9).cuda()
    x = bf(rnd(M, Kd, seed=71))
    w1 = bf(rnd(N, Kd, scale=0.05, seed=72))
    b1 = rnd(N, scale=0.1, seed=73)
    u = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    g = torch.empty_like(u)
    K.gemm_tn(K.EPI_BIAS_GELUG, x, w1, u, out1=g, bias=b1, rowscale=sc, rows_per_sample=T)
    ub = torch.full((K.blocked_rows(M), N), float("nan"), dtype=torch.bfloat16, device="cuda")
    gb = torch.empty_like(g)
    K.gemm_tn(K.EPI_BIAS_GELUG, x, w1, ub, out1=gb, bias=b1, rowscale=sc, rows_per_sample=T, M=M, blocked=True)
    assert torch.equal(gb, g)
    assert torch.equal(_deblock(ub, M, N), u)
    dy = bf(rnd(M, Kd, seed=74))
    w2t = bf(rnd(N, Kd, scale=0.05, seed=75))
    du, db = torch.empty_like(u), torch.empty(N, device="cuda")
    K.gemm_tn(K.EPI_DMUL, dy, w2t, du, aux=u, dbias=db)
    du2, db2 = torch.empty_like(u), torch.empty(N, device="cuda")
    K.gemm_tn(K.EPI_DMUL, dy, w2t, du2, aux=ub, dbias=db2, blocked=True)
    assert torch.equal(du2, du) and torch.equal(db2, db)
    with pytest.raises(RuntimeError, match="blocked"):
        K.gemm_tn(K.EPI_BF16, x, w1, u, blocked=True)
