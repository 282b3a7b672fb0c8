"""GPU: the full HIP path (model classes -> C ABI) against (a) the golden vectors captured from the reference
and (b) the fp32 oracle run on the host, for every fixture: prediction, loss and ALL parameter gradients.

Tolerances (bf16 MFMA operands, fp32 accumulate / residual stream), set in round 3 to about twice
the worst value measured on MI355X over the seven fixtures (each test prints what it measured: `pytest -s`):
  prediction: max-abs error <= 2e-2 * max|pred|  (measured <= 9.4e-3, at 12 layers)
  loss      : 4.5e-3 relative                     (measured <= 2.1e-3)
  gradients : per-parameter L2 norm within 1.5 % (measured <= 6.3e-3), sampled elements within 6 % of the sample's max-abs (normalised
              max-abs; measured <= 3.9e-2, the q_bias cancellation sum of the 3-layer train case), cosine similarity of
              every full gradient tensor vs the oracle >= 0.9997 (measured >= 0.99989).
"""
RTOL_PRED, RTOL_LOSS, RTOL_GNORM, ATOL_GVAL, MIN_COS = 2e-2, 4.5e-3, 1.5e-2, 6e-2, 0.9997
import pytest
import torch
import torch.nn.functional as F

from tests.golden_util import check_against_golden, golden_names, grouped, load_case

pytestmark = pytest.mark.gpu


def run_hip(case, train):
    from get_model_amd.config import build_model, region_model_yaml
    model = build_model(region_model_yaml(case["kind"], num_layers=case["L"]))
    model.load_state_dict(case["sd"], strict=True)
    model = model.cuda()
    model.train(train)
    if train:
        for i, blk in enumerate(model.encoder.blocks):
            if hasattr(blk.drop_path, "injected"):
                blk.drop_path.injected = [case["keep"][i][0], case["keep"][i][1]]
    batch = {k: v.cuda() for k, v in case["batch"].items()}
    inp = model.get_input(batch)                       # the reference's training-step protocol, run.py:210-224
    out = model(**inp)
    pred, obs = model.before_loss(out, batch)
    loss = model.after_loss(model.loss(pred, obs))
    loss.backward()
    key = list(pred.keys())[0]
    grads = {n: p.grad for n, p in model.named_parameters()}
    return loss.detach(), pred[key].detach(), grads


@pytest.mark.parametrize("name", golden_names())
def test_hip_path_matches_reference_golden_and_oracle(name):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import get_cpu as O
    c = load_case(name)
    loss, pred, grads = run_hip(c, c["train"])
    assert all(g is not None for g in grads.values()), [n for n, g in grads.items() if g is None]
    worst = check_against_golden(c, loss.cpu(), pred.cpu(), grads, rtol_pred=RTOL_PRED, rtol_loss=RTOL_LOSS,
                                 rtol_gnorm=RTOL_GNORM, atol_gval_rel=ATOL_GVAL)
    # full-tensor comparison against the oracle (the golden file only samples gradients)
    torch.set_num_threads(16)
    oloss, opred, ograds = O.loss_and_grads(c["kind"], c["sd"], c["batch"], num_layers=c["L"], keep_masks=c["keep"])
    min_cos = 1.0
    gg, og = grouped(grads), grouped(ograds)
    for n, g in gg.items():
        ref = og[n]
        cos = F.cosine_similarity(g, ref, dim=0).item()
        min_cos = min(min_cos, cos)
        assert cos >= MIN_COS, f"{n}: cosine vs oracle {cos}"
    perr = (pred.cpu() - opred).abs().max().item() / opred.abs().max().item()
    print(f"{name}: pred err {perr:.2e} of max, loss {loss.item():.6f} vs {oloss.item():.6f}, "
          f"worst sampled grad err {worst:.2e}, min grad cosine {min_cos:.5f}")


def test_eval_forward_is_deterministic_and_state_dict_roundtrips():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from get_model_amd.config import build_model, region_model_yaml
    c = load_case("ft_L2_B2_N200_eval")
    model = build_model(region_model_yaml("finetune", num_layers=2))
    model.load_state_dict(c["sd"])
    model = model.cuda().eval()
    x = c["batch"]["region_motif"].cuda()
    with torch.no_grad():
        a = model(x)
        b = model(x)
    assert torch.equal(a, b)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    assert list(sd) == list(c["sd"]) and all(torch.equal(sd[k], c["sd"][k]) for k in sd)
    # Lightning checkpoints prefix every key with "model." (config/finetune/lightning.yaml:3-4)
    stripped = {k[len("model."):]: v for k, v in {"model." + k: v for k, v in sd.items()}.items()}
    model.load_state_dict(stripped, strict=True)
    # pretrain -> finetune with strict=False drops head_mask / mask_token (tutorials/predict_atac.ipynb)
    pre = build_model(region_model_yaml("pretrain", num_layers=2))
    missing, unexpected = model.load_state_dict(pre.state_dict(), strict=False)
    assert set(unexpected) == {"mask_token", "head_mask.weight", "head_mask.bias"}
    assert set(missing) == {"head_exp.head.weight", "head_exp.head.bias"}


def test_frozen_layers_get_no_grad():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from get_model_amd.config import build_model, region_model_yaml
    c = load_case("ft_L2_B2_N200_eval")
    model = build_model(region_model_yaml("finetune", num_layers=2))
    model.load_state_dict(c["sd"])
    model.freeze_layers(["encoder.blocks.0", "region_embed"])
    model = model.cuda().eval()
    batch = {k: v.cuda() for k, v in c["batch"].items()}
    out = model(**model.get_input(batch))
    pred, obs = model.before_loss(out, batch)
    model.after_loss(model.loss(pred, obs)).backward()
    for n, p in model.named_parameters():
        frozen = n.startswith("encoder.blocks.0") or n.startswith("region_embed")
        assert (p.grad is None) == frozen, n


def test_grad_buckets_hold_the_same_gradients_without_copies():
    """ddp.GradBuckets: weight gradients are written in place into the flat buckets (param.grad aliases its slot),
    bit-identical to the bucket-free run; a second backward without zero_grad accumulates (micro-batches)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from get_model_amd.config import build_model, region_model_yaml
    from get_model_amd.ddp import GradBuckets
    c = load_case("ft_L2_B2_N200_eval")
    _, _, ref = run_hip(c, False)
    model = build_model(region_model_yaml("finetune", num_layers=2))
    model.load_state_dict(c["sd"])
    model = model.cuda().eval()
    buckets = GradBuckets(model)
    batch = {k: v.cuda() for k, v in c["batch"].items()}

    def backward_once():
        out = model(**model.get_input(batch))
        pred, obs = model.before_loss(out, batch)
        model.after_loss(model.loss(pred, obs)).backward()

    for f in buckets.flat:
        f.fill_(float("nan"))           # stale contents must never leak into a gradient
    buckets.zero_grad()
    backward_once()
    buckets.finish()
    lo = {f.data_ptr(): f.data_ptr() + f.numel() * 4 for f in buckets.flat}
    for n, p in model.named_parameters():
        assert any(a <= p.grad.data_ptr() < b for a, b in lo.items()), f"{n}: grad is not a bucket view"
        assert torch.equal(p.grad, ref[n]), n
    backward_once()                     # no zero_grad: accumulate
    buckets.finish()
    for n, p in model.named_parameters():
        torch.testing.assert_close(p.grad, 2 * ref[n], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("fused", [False, True])
def test_optimizer_step_refreshes_the_bf16_weight_copies(fused):
    """After optimizer.step() (fused AdamW does not bump Parameter._version) the next forward must use the new
    weights: it has to equal, bit for bit, the forward of a fresh model loaded from the updated state_dict."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from get_model_amd.config import build_model, region_model_yaml
    c = load_case("ft_L2_B2_N200_eval")
    model = build_model(region_model_yaml("finetune", num_layers=2))
    model.load_state_dict(c["sd"])
    model = model.cuda().eval()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, fused=fused)
    batch = {k: v.cuda() for k, v in c["batch"].items()}

    def loss_of(m):
        out = m(**m.get_input(batch))
        pred, obs = m.before_loss(out, batch)
        return m.after_loss(m.loss(pred, obs))

    l0 = loss_of(model)
    l0.backward()
    opt.step()
    with torch.no_grad():
        l1 = loss_of(model)
    fresh = build_model(region_model_yaml("finetune", num_layers=2))
    fresh.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    fresh = fresh.cuda().eval()
    with torch.no_grad():
        l2 = loss_of(fresh)
    assert l1.item() != l0.item()
    assert torch.equal(l1, l2), (l0.item(), l1.item(), l2.item())


def test_full_size_batch_is_consistent_with_its_halves():
    """BASELINE.json configs[1] shape (B=256, N=200 -> M = 51 456 rows: the persistent / 224-row-tile / split-slab
    kernel paths the small golden cases never reach), size-independent property: predictions of the full batch equal
    those of its two halves, and the gradient of the mean loss is the mean of the halves' gradients."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from get_model_amd.config import build_model, region_model_yaml
    from oracle import get_cpu as O
    L, B, N = 2, 256, 200
    model = build_model(region_model_yaml("finetune", num_layers=L))
    model.load_state_dict(O.fill_state_dict("finetune", seed=5, num_layers=L))
    model = model.cuda().eval()
    batch = {k: v.cuda() for k, v in O.make_batch(B, N, 9).items()}

    def run(sl):
        for p in model.parameters():
            p.grad = None
        b = {k: v[sl] for k, v in batch.items()}
        out = model(**model.get_input(b))
        pred, obs = model.before_loss(out, b)
        loss = model.after_loss(model.loss(pred, obs))
        loss.backward()
        return loss.detach(), out.detach(), {n: p.grad.clone() for n, p in model.named_parameters()}

    lf, pf, gf = run(slice(0, B))
    la, pa, ga = run(slice(0, B // 2))
    lb, pb, gb = run(slice(B // 2, B))
    assert torch.isfinite(pf).all()
    scale = pf.abs().max().item()
    assert (pf - torch.cat([pa, pb])).abs().max().item() <= 2e-2 * scale
    assert abs(lf.item() - 0.5 * (la.item() + lb.item())) <= 1e-3 * abs(lf.item())
    # ... and equal those of 4 samples run alone (M = 804: the 128x128-tile kernels the golden cases pin)
    _, p4, _ = run(slice(0, 4))
    assert (pf[:4] - p4).abs().max().item() <= 2e-2 * scale
    gfull, ghalf = grouped(gf), grouped({n: 0.5 * (ga[n] + gb[n]) for n in ga})
    for n, g in gfull.items():
        cos = F.cosine_similarity(g.flatten(), ghalf[n].flatten(), dim=0).item()
        assert cos > 0.999, f"{n}: full batch vs halves cosine {cos}"
        assert abs(g.norm().item() / ghalf[n].norm().item() - 1) < 2e-2, n
