"""GPU: product branches that the training-step fixtures do not reach (VERDICT r1 "What's missing" 1, "weak").

  * forward-only / inference path (BASELINE config 5; run_region.py:183-223 `predict_step`): under
    torch.inference_mode the fc1 GEMM takes its one-output GELU epilogue (out0 == NULL, csrc/gemm.hip);
  * the interpretation use of the path (run.py:351-434): forward hooks + retain_grad on `region_embed` and
    `encoder.blocks.i`, inputs with requires_grad (input-gradient GEMM of RegionEmbedTokens.backward);
  * `region_embed(x)` on its own and `concat_cls` on a foreign tensor;
  * a parameter used by two graph nodes of ONE backward with ddp.GradBuckets active;
  * the side-stream weight-gradient mode (GETMI_WGRAD_STREAM=1).
Tolerances as in test_model_parity_gpu.py (bf16 MFMA operands, fp32 accumulate): pred 2e-2 of max|pred| (PRED), loss 4.5e-3,
gradients / Jacobians 6 % of the tensor's scale on sampled elements, 4 % on norms, cosine >= 0.9995; printed as MEASURED.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.golden_util import golden_names, load_case

pytestmark = pytest.mark.gpu
PRED = 2e-2


def _perr(tag, got, ref, bar=PRED):
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print(f"MEASURED {tag}: pred err {err:.2e} of max")
    assert err <= bar, (tag, err)


EXTRA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_extra")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _model(kind, L, sd, train=False):
    from get_model_amd.config import build_model, region_model_yaml
    m = build_model(region_model_yaml(kind, num_layers=L))
    m.load_state_dict(sd, strict=True)
    return m.cuda().train(train)


@pytest.mark.parametrize("name", [n for n in golden_names() if n.endswith("_eval")])
@pytest.mark.parametrize("mode", ["inference_mode", "no_grad"])
def test_forward_only_matches_reference_golden_and_oracle(name, mode):
    """All three models, forward only: prediction vs the reference's golden prediction and vs the oracle."""
    _need_gpu()
    from oracle import get_cpu as O
    c = load_case(name)
    model = _model(c["kind"], c["L"], c["sd"])
    batch = {k: v.cuda() for k, v in c["batch"].items()}
    ctx = torch.inference_mode() if mode == "inference_mode" else torch.no_grad()
    with ctx:
        out = model(**model.get_input(batch))
        pred, obs = model.before_loss(out, batch)
        loss = model.after_loss(model.loss(pred, obs))
    key = list(pred)[0]
    got = pred[key].float().cpu()
    gold = torch.from_numpy(c["z"]["pred"])
    scale = gold.abs().max().item()
    assert got.shape == gold.shape
    _perr(f"forward-only {name} {mode} vs golden", got, gold)
    assert abs(loss.item() - float(c["z"]["loss"])) <= 4.5e-3 * abs(float(c["z"]["loss"]))
    torch.set_num_threads(16)
    if c["kind"] == "pretrain":
        _, opred = O.pretrain_loss(c["sd"], c["batch"], num_layers=c["L"])
    else:
        _, opred = O.finetune_loss(c["sd"], c["batch"], kind=c["kind"], num_layers=c["L"])
    _perr(f"forward-only {name} {mode} vs oracle", got, opred)
    # the forward-only epilogue must give the same numbers as the training forward (which also stores gelu')
    out_t = model(**model.get_input(batch))
    pred_t, _ = model.before_loss(out_t, batch)
    assert torch.equal(pred_t[key].detach().float().cpu(), got)


def test_c5_large_batch_inference_slice_against_the_oracle():
    """BASELINE config 5 shape on one GPU: B=1024 x N=200, 12 layers, torch.inference_mode (M = 205 824 rows: the
    persistent GEMMs with the one-output GELU epilogue at full size); samples 0-1 and the last two checked against the
    oracle run on those four samples alone (samples are independent on this path)."""
    _need_gpu()
    from oracle import get_cpu as O
    L, B, N = 12, 1024, 200
    sd = O.fill_state_dict("finetune", seed=31, num_layers=L)
    model = _model("finetune", L, sd)
    small = O.make_batch(8, N, 77)
    rm = small["region_motif"].repeat(B // 8, 1, 1)
    rm[-2:] = O.make_batch(2, N, 78)["region_motif"]
    with torch.inference_mode():
        pred = model(rm.cuda())
    assert pred.shape == (B, N, 2) and torch.isfinite(pred).all()
    idx = [0, 1, B - 2, B - 1]
    torch.set_num_threads(16)
    opred = O.finetune_forward(sd, rm[idx], num_layers=L)
    got = pred[idx].float().cpu()
    _perr("C5 B=1024 slice", got, opred)
    # repeated samples give repeated predictions (no cross-sample leakage through tile tails)
    assert torch.equal(pred[0], pred[8]) and torch.equal(pred[3], pred[B - 13])


def test_c2_full_batch_slice_against_the_oracle():
    """BASELINE config 2 shape (B=256 x N=200, 12 layers, training forward): four samples against the oracle."""
    _need_gpu()
    from oracle import get_cpu as O
    L, B, N = 12, 256, 200
    sd = O.fill_state_dict("finetune", seed=32, num_layers=L)
    model = _model("finetune", L, sd)
    batch = O.make_batch(B, N, 79)
    pred = model(batch["region_motif"].cuda())
    idx = [0, 100, 254, 255]
    torch.set_num_threads(16)
    opred = O.finetune_forward(sd, batch["region_motif"][idx], num_layers=L)
    got = pred[idx].detach().float().cpu()
    _perr("C2 B=256 slice", got, opred)


def test_c3_full_batch_pretrain_slice_against_the_oracle():
    """BASELINE config 3 shape (GETRegionPretrain, B=256 x N=200, 12 layers): the masked predictions of four samples
    against the oracle run on those samples alone (device-side row compaction, mask-token blend in the embed epilogue,
    masked head at full size)."""
    _need_gpu()
    from oracle import get_cpu as O
    L, B, N = 12, 256, 200
    sd = O.fill_state_dict("pretrain", seed=33, num_layers=L)
    model = _model("pretrain", L, sd)
    batch = O.make_batch(B, N, 81, kind="pretrain")
    cb = {k: v.cuda() for k, v in batch.items()}
    out = model(**model.get_input(cb))
    pred, obs = model.before_loss(out, cb)
    key = list(pred.keys())[0]
    got_all, obs_all = pred[key].detach().float().cpu(), obs[key].detach().float().cpu()
    cnt = batch["mask"].sum(1)
    start = torch.cumsum(cnt, 0) - cnt
    assert got_all.shape == (int(cnt.sum()), 283)
    idx = [0, 101, 254, 255]
    torch.set_num_threads(16)
    sub = {k: v[idx] for k, v in batch.items()}
    _, opred = O.pretrain_loss(sd, sub, num_layers=L)
    got = torch.cat([got_all[int(start[i]):int(start[i] + cnt[i])] for i in idx])
    assert got.shape == opred.shape
    _perr("C3 pretrain B=256 slice", got, opred)
    # the targets handed to the loss are the original rows of the masked regions, in the same order
    want = torch.cat([batch["region_motif"][i][batch["mask"][i].bool()] for i in idx])
    assert torch.equal(torch.cat([obs_all[int(start[i]):int(start[i] + cnt[i])] for i in idx]), want)


def test_c4_long_context_atac_slice_against_the_oracle():
    """BASELINE config 4 shape (GETRegionFinetuneATAC, B=16 x N=2000, 12 layers): two samples against the oracle, and one
    training step at that size (streamed attention backward over nine key blocks) gives finite gradients that do not
    depend on the other samples of the batch."""
    _need_gpu()
    from oracle import get_cpu as O
    L, B, N = 12, 16, 2000
    sd = O.fill_state_dict("atac", seed=34, num_layers=L)
    model = _model("atac", L, sd)
    batch = O.make_batch(B, N, 82, kind="atac")
    cb = {k: v.cuda() for k, v in batch.items()}
    out = model(**model.get_input(cb))
    pred, obs = model.before_loss(out, cb)
    key = list(pred.keys())[0]
    idx = [0, 15]
    torch.set_num_threads(16)
    _, opred = O.finetune_loss(sd, {k: v[idx] for k, v in batch.items()}, kind="atac", num_layers=L)
    got = pred[key][idx].detach().float().cpu()
    _perr("C4 atac N=2000 slice", got, opred)
    loss = model.after_loss(model.loss(pred, obs))
    loss.backward()
    g_full = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    assert all(torch.isfinite(g).all() for g in g_full.values())
    # the same two samples alone, loss re-weighted to their share of the batch mean: the gradient is a sum over samples,
    # so it must equal the full-batch gradient minus the other samples' part -- checked through linearity on a 2+14 split
    model.zero_grad(set_to_none=True)
    parts = []
    for sel in (idx, [i for i in range(B) if i not in idx]):
        sb = {k: v[sel] for k, v in cb.items()}
        o2 = model(**model.get_input(sb))
        p2, ob2 = model.before_loss(o2, sb)
        l2 = model.after_loss(model.loss(p2, ob2)) * (len(sel) / B)
        l2.backward()
    worst = (1.0, "")
    for n, p in model.named_parameters():
        ref = g_full[n].float().flatten()
        cos = F.cosine_similarity(p.grad.float().flatten(), ref, dim=0).item()
        worst = min(worst, (cos, n))
        # q_bias is a cancellation sum (rows of the softmax Jacobian sum to zero): same test, its own bar
        assert cos > (0.99 if n.endswith("attn.q_bias") else 0.999), (n, cos)
    print(f"MEASURED C4 linearity: min cosine {worst[0]:.6f} ({worst[1]})")


def _close(name, got, idx, vals, gsum, tol_val=6e-2, tol_norm=4e-2):
    g = got.detach().double().cpu().flatten()
    ref_scale = max(np.abs(vals).max(), gsum[1] / np.sqrt(g.numel()), 1e-20)
    err = np.abs(g[idx].numpy() - vals).max() / ref_scale
    assert err <= tol_val, f"{name}: sampled error {err:.3e} of scale"
    assert abs(g.norm().item() - gsum[1]) <= tol_norm * gsum[1], f"{name}: norm {g.norm().item()} vs {gsum[1]}"


def test_interpretation_hooks_and_input_jacobian_match_the_reference():
    """run.py:351-434 on the HIP path: the same hook code as the reference's `capture_target_tensor`, compared with
    what the reference produced (tests/golden_extra/interpret_ft_L2_B2_N200.npz) and, full tensor, with the oracle."""
    _need_gpu()
    from oracle import get_cpu as O
    z = np.load(os.path.join(EXTRA, "interpret_ft_L2_B2_N200.npz"), allow_pickle=False)
    L, B, N, seed, focus, strand = (int(v) for v in z["meta"])
    sd = O.fill_state_dict("finetune", seed=seed, num_layers=L)
    model = _model("finetune", L, sd)
    batch = {k: v.cuda() for k, v in O.make_batch(B, N, seed).items()}
    captured = {}

    def capture(name):
        def hook(module, inp, output):
            if isinstance(output, tuple):
                output = output[0]
            output.retain_grad()
            captured[name] = output
        return hook

    hooks = [model.get_submodule(n).register_forward_hook(capture(n)) for n in ("region_embed", "encoder.blocks.1")]
    inp = model.get_input(batch)
    for t in inp.values():
        t.requires_grad = True
    out = model(**inp)
    pred, _ = model.before_loss(out, batch)
    m = torch.zeros_like(pred["exp"])
    m[:, focus, strand] = 1
    pred["exp"].backward(m)
    for h in hooks:
        h.remove()
    tensors = {"input": inp["region_motif"], "region_embed": captured["region_embed"],
               "encoder.blocks.1": captured["encoder.blocks.1"]}
    for name, t in tensors.items():
        assert tuple(t.shape) == tuple(z["shape/" + name]), name          # the reference's hook shapes (no CLS row
        assert t.grad is not None and t.grad.shape == t.shape, name       # in region_embed's output)
        _close("d" + name, t.grad, z["gidx/" + name], z["gval/" + name], z["gsum/" + name])
        if name != "input":
            _close(name, t, z["vidx/" + name], z["vval/" + name], z["vsum/" + name], tol_val=2.5e-2, tol_norm=1e-2)
    fr = inp["region_motif"].grad[:, focus].float().cpu()
    ref = torch.from_numpy(z["input_focus_rows"])
    assert (fr - ref).abs().max().item() <= 6e-2 * ref.abs().max().item()
    # full tensors against the oracle
    torch.set_num_threads(16)
    leaf = batch["region_motif"].detach().cpu().clone().requires_grad_(True)
    x_emb = F.linear(leaf, sd["region_embed.embed.weight"], sd["region_embed.embed.bias"])
    x_emb.retain_grad()
    x = torch.cat((sd["cls_token"].expand(B, -1, -1), x_emb), dim=1)
    x = O.block(x, sd, 0, 12)
    x1 = O.block(x, sd, 1, 12)
    x1.retain_grad()
    h = F.layer_norm(x1, (768,), sd["encoder.norm.weight"], sd["encoder.norm.bias"], O.LN_EPS)[:, 1:]
    op = F.softplus(F.linear(h, sd["head_exp.head.weight"], sd["head_exp.head.bias"]))
    op.backward(m.cpu())
    for name, ref_t in (("input", leaf), ("region_embed", x_emb), ("encoder.blocks.1", x1)):
        cos = F.cosine_similarity(tensors[name].grad.float().cpu().flatten(), ref_t.grad.flatten(), dim=0).item()
        print(f"MEASURED interpretation d{name}: cosine vs oracle {cos:.6f}")
        assert cos >= 0.9995, f"d{name}: cosine vs oracle {cos}"


def test_region_embed_alone_and_concat_cls_of_a_foreign_tensor():
    """`RegionEmbed.forward(x)` (model/modules.py:101-103) without the models around it, and the CLS concat on a
    tensor that is NOT region_embed's output (a hook may replace it): it must be copied, never written around."""
    _need_gpu()
    from get_model_amd import functional as Fn
    from oracle import get_cpu as O
    c = load_case("atac_L2_B2_N37_eval")
    model = _model("atac", 2, c["sd"])
    x = c["batch"]["region_motif"].cuda().requires_grad_(True)
    e = model.region_embed(x)
    ref = F.linear(c["batch"]["region_motif"], c["sd"]["region_embed.embed.weight"], c["sd"]["region_embed.embed.bias"])
    assert e.shape == ref.shape == (2, 37, 768)
    assert (e.detach().cpu() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()
    w = torch.randn_like(ref)
    (e * w.cuda()).sum().backward()          # gradient arrives contiguous: the non-aliased branch of the backward
    gw = torch.einsum("bnd,bnf->df", w, c["batch"]["region_motif"])
    got = model.region_embed.embed.weight.grad.cpu()
    assert F.cosine_similarity(got.flatten(), gw.flatten(), dim=0).item() > 0.9995
    gx = w @ c["sd"]["region_embed.embed.weight"]
    assert F.cosine_similarity(x.grad.cpu().flatten(), gx.flatten(), dim=0).item() > 0.9995
    # foreign tensor: a slice of somebody else's buffer with the same strides as region_embed's output
    own = torch.randn(2, 38, 768, device="cuda")
    keep = own.clone()
    view = own[:, 1:]
    cls = torch.randn(1, 1, 768, device="cuda", requires_grad=True)
    full = Fn.concat_cls(view, cls)
    assert torch.equal(own, keep), "concat_cls wrote into a tensor it does not own"
    assert torch.equal(full[:, 1:], view) and torch.equal(full[:, 0], cls.detach().view(1, 768).expand(2, 768))
    full.sum().backward()
    assert torch.allclose(cls.grad, torch.full_like(cls, 2.0))


def test_two_forwards_one_backward_with_grad_buckets():
    """with GradBuckets active, a parameter used by two graph nodes of one backward pass must get g1 + g2
    (both weight-gradient GEMMs used to write the same bucket slot and autograd summed two aliases of it)."""
    _need_gpu()
    from get_model_amd.ddp import GradBuckets
    from oracle import get_cpu as O
    c = load_case("ft_L2_B2_N200_eval")
    model = _model("finetune", 2, c["sd"])
    ba = {k: v.cuda() for k, v in c["batch"].items()}
    bb = {k: v.cuda() for k, v in O.make_batch(2, 200, 55).items()}

    def loss_of(b):
        out = model(**model.get_input(b))
        pred, obs = model.before_loss(out, b)
        return model.after_loss(model.loss(pred, obs))

    ref = {}
    for b in (ba, bb):
        for p in model.parameters():
            p.grad = None
        loss_of(b).backward()
        for n, p in model.named_parameters():
            ref[n] = ref.get(n, 0) + p.grad.clone()
    buckets = GradBuckets(model)
    for step in range(2):                    # the second step also checks that finish() re-armed the bookkeeping
        model.zero_grad(set_to_none=True)    # NOT buckets.zero_grad(): the loop of a reference user
        (loss_of(ba) + loss_of(bb)).backward()
        buckets.finish()
        for n, p in model.named_parameters():
            torch.testing.assert_close(p.grad, ref[n], rtol=2e-5, atol=1e-7, msg=lambda s, n=n: f"{n} step {step}: {s}")
    buckets.remove()


def test_side_stream_weight_gradients_equal_the_default_order():
    """GETMI_WGRAD_STREAM=1 (functional._SideStream): same kernels on a second stream, joined before the block's
    backward returns -> bit-identical gradients."""
    _need_gpu()
    from get_model_amd import functional as Fn
    c = load_case("ft_L3_B4_N200_train")

    def grads(side):
        old = Fn._USE_SIDE
        Fn._USE_SIDE = side
        try:
            model = _model("finetune", c["L"], c["sd"], train=True)
            for i, blk in enumerate(model.encoder.blocks):
                if hasattr(blk.drop_path, "injected"):
                    blk.drop_path.injected = [c["keep"][i][0], c["keep"][i][1]]
            b = {k: v.cuda() for k, v in c["batch"].items()}
            out = model(**model.get_input(b))
            pred, obs = model.before_loss(out, b)
            model.after_loss(model.loss(pred, obs)).backward()
            torch.cuda.synchronize()
            return {n: p.grad.clone() for n, p in model.named_parameters()}
        finally:
            Fn._USE_SIDE = old

    a, b = grads(False), grads(True)
    for n in a:
        assert torch.equal(a[n], b[n]), n
