"""Race screen (GPU): every pipelined kernel is run many times on the same inputs, with a bandwidth hog on a second
stream to perturb LDS-DMA arrival order, and every result must be bit-identical to the first (the kernels are
deterministic by construction; an early LDS read or a mis-counted vmcnt wait shows up as a rare differing tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from get_model_amd import kernels as K

torch.manual_seed(0)
M, D, F, H, T, B = 51456, 768, 3072, 12, 201, 256
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(0)
def rb(*s): return (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(dev)
def rf(*s): return (torch.randn(*s, generator=g) * 0.5).to(dev)
xD, xF, x3 = rb(M, D), rb(M, F), rb(M, 3 * D)
wqkv, wproj, w1, w2, wqkvT = rb(3 * D, D), rb(D, D), rb(F, D), rb(D, F), rb(D, 3 * D)
bD, bF, b3, resid = rf(D), rf(F), rf(3 * D), rf(M, D)
scale = torch.rand(B, device=dev) + 0.5
hogA = torch.empty(64 * 1024 * 1024, device=dev); hogB = torch.empty_like(hogA)
side = torch.cuda.Stream()

def cases():
    o3 = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
    yield "qkv fwd", lambda: K.gemm_tn(K.EPI_BF16, xD, wqkv, o3, bias=b3), lambda: o3
    oF, oF2 = torch.empty(M, F, dtype=torch.bfloat16, device=dev), torch.empty(M, F, dtype=torch.bfloat16, device=dev)
    yield "fc1 fwd (gelu')", lambda: K.gemm_tn(K.EPI_BIAS_GELUG, xD, w1, oF, out1=oF2, bias=bF), lambda: torch.cat([oF, oF2])
    oDf = torch.empty(M, D, device=dev)
    yield "fc2 fwd (resid, 224)", lambda: K.gemm_tn(K.EPI_BIAS_RESID, xF, w2, oDf, bias=bD, resid=resid, rowscale=scale, rows_per_sample=T), lambda: oDf
    yield "proj fwd (resid, 224)", lambda: K.gemm_tn(K.EPI_BIAS_RESID, xD, wproj, oDf, bias=bD, resid=resid), lambda: oDf
    db = torch.empty(F, device=dev)
    yield "fc2 dgrad (dmul)", lambda: K.gemm_tn(K.EPI_DMUL, xD, w1, oF, aux=xF, dbias=db), lambda: torch.cat([oF.float().flatten(), db])
    oD = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    yield "fc1 dgrad (224)", lambda: K.gemm_tn(K.EPI_BF16, xF, w2, oD), lambda: oD
    yield "qkv dgrad (224)", lambda: K.gemm_tn(K.EPI_BF16, x3, wqkvT, oD), lambda: oD
    res = {}
    yield "wgrad fc1", lambda: res.__setitem__("w", K.gemm_wgrad(xF, xD)), lambda: res["w"]
    yield "wgrad qkv", lambda: res.__setitem__("w", K.gemm_wgrad(x3, xD)), lambda: res["w"]
    yield "wgrad proj", lambda: res.__setitem__("w", K.gemm_wgrad(xD, xD)), lambda: res["w"]
    out, lse = K.attention_fwd(x3, B, T, H, 0.125)
    yield "attention fwd", lambda: res.__setitem__("a", K.attention_fwd(x3, B, T, H, 0.125)), lambda: torch.cat([res["a"][0].float().flatten(), res["a"][1].flatten()])
    yield "attention bwd", lambda: res.__setitem__("b", K.attention_bwd(x3, out, xD, lse, B, T, H, 0.125, want_bias=True)), \
        lambda: torch.cat([res["b"][0].float().flatten(), res["b"][1], res["b"][2]])
    # as the training step calls it (rounding residual of the output), and a long context (key blocks + dQ slabs): the
    # streamed backward orders its LDS-DMA against its readers with counted vmcnt waits only
    out2, lse2, lo2 = K.attention_fwd(x3, B, T, H, 0.125, want_lo=True)
    yield "attention bwd (O_lo)", lambda: res.__setitem__("c", K.attention_bwd(x3, out2, xD, lse2, B, T, H, 0.125, want_bias=True, out_lo=lo2)), \
        lambda: torch.cat([res["c"][0].float().flatten(), res["c"][1], res["c"][2]])
    # round 3: the blocked hand-over of gelu'(u), the folded DropPath factors, the bf16-stream LayerNorm backward
    ub = torch.empty(K.blocked_rows(M), F, dtype=torch.bfloat16, device=dev)
    yield "fc1 fwd (blocked gelu', row scale)", lambda: K.gemm_tn(K.EPI_BIAS_GELUG, xD, w1, ub, out1=oF2, bias=bF, rowscale=scale,
                                                                   rows_per_sample=T, M=M, blocked=True), lambda: torch.cat([ub[:M], oF2])
    K.gemm_tn(K.EPI_BIAS_GELUG, xD, w1, ub, out1=oF2, bias=bF, rowscale=scale, rows_per_sample=T, M=M, blocked=True)
    ubk = ub.clone()
    yield "fc2 dgrad (dmul, blocked aux)", lambda: K.gemm_tn(K.EPI_DMUL, xD, w1, oF, aux=ubk, dbias=db, blocked=True), \
        lambda: torch.cat([oF.float().flatten(), db])
    yield "fc2 fwd (resid, prescaled)", lambda: K.gemm_tn(K.EPI_BIAS_RESID, xF, w2, oDf, bias=bD, resid=resid, rowscale=scale,
                                                          rows_per_sample=T, prescaled=True), lambda: oDf
    drop = scale.clone(); drop[::5] = 0.0
    out3, lse3, lo3 = K.attention_fwd(x3, B, T, H, 0.125, want_lo=True, sample_scale=drop)
    yield "attention fwd (sample scale)", lambda: res.__setitem__("e", K.attention_fwd(x3, B, T, H, 0.125, want_lo=True, sample_scale=drop)), \
        lambda: torch.cat([res["e"][0].float().flatten(), res["e"][1].flatten(), res["e"][2].float().flatten()])
    yield "attention bwd (sample scale)", lambda: res.__setitem__("f", K.attention_bwd(x3, out3, xD, lse3, B, T, H, 0.125, want_bias=True,
                                                                                       out_lo=lo3, sample_scale=drop)), \
        lambda: torch.cat([res["f"][0].float().flatten(), res["f"][1], res["f"][2]])
    xf32 = rf(M, D); gam = rf(D)
    _, mean, rstd = K.layernorm_fwd(xf32, gam, bD)
    yield "layernorm bwd (bf16 stream)", lambda: res.__setitem__("g", K.layernorm_bwd(xD, xf32, mean, rstd, gam, dres=oD, rowscale=scale,
                                                                                      rows_per_sample=T, want_dz=False, want_dbias=True,
                                                                                      dx_dtype=torch.bfloat16)), \
        lambda: torch.cat([res["g"][0].float().flatten(), res["g"][2], res["g"][3], res["g"][4]])
    BL, TL = 4, 2001
    xl, dl = rb(BL * TL, 3 * H * 64), rb(BL * TL, H * 64)
    outl, lsel, lol = K.attention_fwd(xl, BL, TL, H, 0.125, want_lo=True)
    yield "attention bwd (T=2001)", lambda: res.__setitem__("d", K.attention_bwd(xl, outl, dl, lsel, BL, TL, H, 0.125, want_bias=True, out_lo=lol)), \
        lambda: torch.cat([res["d"][0].float().flatten(), res["d"][1], res["d"][2]])

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for name, run, result in cases():
    run(); torch.cuda.synchronize()
    ref = result().clone()
    diff = 0
    for it in range(iters):
        with torch.cuda.stream(side):
            if it % 2: hogB.copy_(hogA)          # HBM traffic on a second stream for every other iteration
        run()
        torch.cuda.synchronize()
        if not torch.equal(result(), ref):
            diff += 1
    print(f"{name:24s} {iters} runs, {diff} differing", flush=True)
    bad += diff
print("RACE SCREEN", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
