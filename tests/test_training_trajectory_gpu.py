"""GPU: several consecutive TRAINING STEPS of the HIP path -- get_input -> forward -> loss -> backward -> gradient buckets
-> FusedAdamW with the reference's parameter groups (run.py:210-234, 477-497) -- against the fp32 oracle trained on the host
with torch.optim.AdamW from the same initial weights on the same batches.  This is the loop bench.py times (SURVEY.md 8
row a14): the single-step tests pin loss and gradients; this one pins what the optimizer makes of them step after step
(bf16 weight-copy refresh, bucket re-arming, per-group hyper-parameters, moment state).

Tolerances (bf16 MFMA operands against fp32; Adam's early steps are sign-like, lr * g / (|g| + eps), so elements with a
tiny gradient may move the other way -- the update is compared by direction, the loss by value):
  loss of every step: 1e-2 relative;   per-tensor update (w_k - w_0) vs the oracle's: cosine >= 0.99 on every matrix,
  >= 0.995 over all parameters together (measured 0.998 / 0.9998);   final weights: max-abs difference <= 2.5 * lr * steps (no runaway element).
"""
from types import SimpleNamespace

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

STEPS, LR, WD = 6, 2e-4, 0.05


def _oracle_run(kind, L, sd0, batches, groups_by_name):
    from oracle import get_cpu as O
    torch.set_num_threads(16)
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd0.items()}
    by_wd = {}
    for n, wd in groups_by_name.items():
        by_wd.setdefault(wd, []).append(leaf[n])
    opt = torch.optim.AdamW([{"params": ps, "weight_decay": wd} for wd, ps in by_wd.items()], lr=LR, betas=(0.9, 0.999),
                            eps=1e-8)
    losses = []
    for batch in batches:
        opt.zero_grad(set_to_none=True)
        if kind == "pretrain":
            loss, _ = O.pretrain_loss(leaf, batch, num_layers=L, drop_path_rate=0.0)
        else:
            loss, _ = O.finetune_loss(leaf, batch, kind=kind, num_layers=L, drop_path_rate=0.0)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses, {k: v.detach() for k, v in leaf.items()}


@pytest.mark.parametrize("kind,N", [("finetune", 64), ("pretrain", 48), ("atac", 40)])
def test_training_steps_follow_the_oracle(kind, N):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import get_cpu as O
    from get_model_amd import optim as GO
    from get_model_amd.config import build_model, region_model_yaml
    from get_model_amd.ddp import GradBuckets

    L, B = 2, 6
    sd0 = O.fill_state_dict(kind, seed=11, num_layers=L)
    batches = [O.make_batch(B, N, 100 + i, kind=kind) for i in range(STEPS)]

    model = build_model(region_model_yaml(kind, num_layers=L))
    model.load_state_dict(sd0, strict=True)
    model = model.cuda()
    model.eval()                       # DropPath off on both sides (its masks are random); everything else trains
    buckets = GradBuckets(model)
    opt = GO.configure_optimizer(model, SimpleNamespace(opt="adamw", lr=LR, weight_decay=WD, opt_eps=1e-8,
                                                        opt_betas=[0.9, 0.999]))
    wd_by_param = {id(p): g["weight_decay"] for g in opt.param_groups for p in g["params"]}
    groups_by_name = {n: wd_by_param[id(p)] for n, p in model.named_parameters()}
    assert set(groups_by_name) == set(sd0), "oracle and model disagree on the parameter names"

    losses = []
    for batch in batches:
        dev = {k: v.cuda() for k, v in batch.items()}
        buckets.zero_grad()
        out = model(**model.get_input(dev))
        pred, obs = model.before_loss(out, dev)
        loss = model.after_loss(model.loss(pred, obs))
        loss.backward()
        buckets.finish()
        opt.step()
        losses.append(float(loss.detach()))
    final = {n: p.detach().float().cpu() for n, p in model.named_parameters()}

    olosses, ofinal = _oracle_run(kind, L, sd0, batches, groups_by_name)

    for s, (a, b) in enumerate(zip(losses, olosses)):
        assert abs(a - b) <= 1e-2 * abs(b), f"{kind}: loss of step {s}: {a} vs oracle {b}"
    du_all, do_all, worst_cos, worst_abs = [], [], 1.0, 0.0
    for n, w0 in sd0.items():
        du, do = (final[n] - w0).flatten(), (ofinal[n] - w0).flatten()
        if float(do.abs().max()) == 0.0:           # parameter without gradient (unused on this path): must not have moved
            assert float(du.abs().max()) == 0.0, f"{n} moved without a gradient"
            continue
        du_all.append(du); do_all.append(do)
        worst_abs = max(worst_abs, float((final[n] - ofinal[n]).abs().max()))
        if w0.ndim >= 2 and w0.numel() >= 4096:
            cos = float(F.cosine_similarity(du, do, dim=0))
            worst_cos = min(worst_cos, cos)
            assert cos >= 0.99, f"{kind}: update of {n}: cosine vs oracle {cos:.4f}"
    cos_all = float(F.cosine_similarity(torch.cat(du_all), torch.cat(do_all), dim=0))
    assert cos_all >= 0.995, f"{kind}: whole-model update cosine {cos_all:.4f}"
    assert worst_abs <= 2.5 * LR * STEPS, f"{kind}: a weight ended {worst_abs:.3e} away from the oracle's"
    print(f"{kind}: losses {['%.5f' % x for x in losses]} vs oracle {['%.5f' % x for x in olosses]}; update cosine "
          f"{cos_all:.4f} (worst matrix {worst_cos:.4f}); max |w - w_oracle| {worst_abs:.2e}")
