"""Optimizer / schedule row: host logic on CPU, the fused AdamW kernel on the GPU against
torch.optim.AdamW."""
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from get_model_amd import optim as O
from get_model_amd.config import build_model, region_model_yaml


def test_cosine_scheduler_matches_the_formula():
    # utils.py:314-345: linspace warm-up min_lr -> lr over warmup_epochs*niter, then half cosine to min_lr
    s = O.cosine_scheduler(1e-3, 1e-5, epochs=10, niter_per_ep=7, warmup_epochs=2, start_warmup_value=1e-5)
    assert len(s) == 70
    assert s[0] == pytest.approx(1e-5) and s[13] == pytest.approx(1e-3)
    n = 70 - 14
    for i in (0, 1, 20, 55):
        assert s[14 + i] == pytest.approx(1e-5 + 0.5 * (1e-3 - 1e-5) * (1 + math.cos(math.pi * i / n)))
    # warmup_steps sets the warm-up length, but warm-up values exist only if warmup_epochs > 0: with
    # warmup_epochs == 0 the reference's own length assertion fires (utils.py:343) -- same here
    with pytest.raises(AssertionError):
        O.cosine_scheduler(1.0, 0.0, epochs=2, niter_per_ep=5, warmup_epochs=0, warmup_steps=3)
    s2 = O.cosine_scheduler(1.0, 0.0, epochs=2, niter_per_ep=5, warmup_epochs=1, warmup_steps=3)
    assert len(s2) == 10 and s2[2] == pytest.approx(1.0) and s2[3] == pytest.approx(1.0)
    s3 = O.cosine_scheduler(1.0, 0.0, epochs=2, niter_per_ep=5)
    assert s3[0] == 1.0 and s3[-1] == pytest.approx(0.5 * (1 + math.cos(math.pi * 9 / 10)))


def test_parameter_groups_follow_the_reference_rules():
    model = build_model(region_model_yaml("finetune", num_layers=2))
    names = dict(model.named_parameters())
    assigner = O.LayerDecayValueAssigner([0.75 ** (2 + 1 - i) for i in range(2 + 2)])
    groups = O.get_parameter_groups(model, 0.05, model.no_weight_decay(), assigner.get_layer_id, assigner.get_scale)
    where = {}
    for g in groups:
        for p in g["params"]:
            where[id(p)] = g
    for n, p in names.items():
        g = where[id(p)]
        no_decay = p.ndim == 1 or n.endswith(".bias") or n in ("pos_embed", "cls_token")
        assert g["weight_decay"] == (0.0 if no_decay else 0.05), n
        lid = O.get_num_layer_for_vit(n, 4)
        assert g["lr_scale"] == pytest.approx(0.75 ** (3 - lid)), n
    assert O.get_num_layer_for_vit("cls_token", 4) == 0 and O.get_num_layer_for_vit("region_embed.embed.weight", 4) == 0
    assert O.get_num_layer_for_vit("encoder.blocks.1.attn.qkv.weight", 4) == 2
    assert O.get_num_layer_for_vit("head_exp.head.weight", 4) == 3
    assert sum(len(g["params"]) for g in groups) == len(names)
    cfg = SimpleNamespace(opt="adamw", lr=1e-4, weight_decay=0.05, opt_eps=1e-8, opt_betas=[0.9, 0.999])
    opt = O.configure_optimizer(model, cfg)
    assert isinstance(opt, O.FusedAdamW) and len(opt.param_groups) == len(groups)
    schedule = O.cosine_scheduler(1e-4, 1e-6, 3, 4, 1, 1e-6)
    sched = O.make_lr_scheduler(opt, schedule, 1e-4)
    assert opt.param_groups[0]["lr"] == pytest.approx(schedule[-1])   # step 0 reads schedule[-1] (run.py:517 quirk)
    with pytest.raises(RuntimeError):                             # CPU parameters: no fallback
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        opt.step()
    # a step that raises must not have advanced any step count (bias correction of the retry)
    assert all(len(opt.state[p]) == 0 or float(opt.state[p]["step"]) == 0.0 for p in model.parameters())
    del sched


@pytest.mark.parametrize("fixture", ["optim_groups_finetune_L12", "optim_groups_pretrain_L12", "optim_groups_atac_L3",
                                     "optim_groups_finetune_L2_frozen"])
def test_parameter_groups_equal_the_reference_optimizer(fixture):
    """Pinned by the reference itself: oracle/make_golden_optim.py imported /root/reference/get_model/optim.py
    (optim.py:16-144) and ran its `create_optimizer` the way run.py:477-497 does; this compares parameter by parameter
    the group index (= order of the groups in the optimizer), weight_decay, the stored lr_scale, lr and the layer id."""
    import json
    import os
    z = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_extra", fixture + ".json")))
    model = build_model(region_model_yaml(z["kind"], num_layers=z["num_layers"]))
    if z["frozen"]:
        model.freeze_layers(patterns_to_freeze=z["frozen"], invert_match=False)
    cfg = SimpleNamespace(opt="adamw", lr=1e-4, weight_decay=0.05, opt_eps=1e-8, opt_betas=[0.9, 0.999])
    opt = O.configure_optimizer(model, cfg)
    assert len(opt.param_groups) == z["n_groups"]
    assert z["optimizer_class"] == "AdamW" and isinstance(opt, O.FusedAdamW)      # AdamW semantics, fused launch
    for k, v in z["defaults"].items():
        got = opt.defaults[k]
        assert (list(got) if isinstance(got, (list, tuple)) else got) == v, k
    name_of = {id(p): n for n, p in model.named_parameters()}
    seen = {}
    for gi, g in enumerate(opt.param_groups):
        for p in g["params"]:
            seen[name_of[id(p)]] = [gi, float(g["weight_decay"]), float(g["lr_scale"]), float(g["lr"])]
    assert set(seen) == set(z["params"])
    for n, want in z["params"].items():
        assert seen[n][0] == want[0] and seen[n][1] == want[1], (n, seen[n], want)
        assert seen[n][2] == pytest.approx(want[2], rel=1e-12) and seen[n][3] == pytest.approx(want[3]), (n, seen[n], want)
        assert O.get_num_layer_for_vit(n, z["num_layers"] + 2) == z["layer_ids"][n], n
    # within each group the parameters come in the reference's order (named_parameters order)
    ref_order = {}
    for n, want in z["params"].items():
        ref_order.setdefault(want[0], []).append(n)
    got_order = {}
    for n, v in seen.items():
        got_order.setdefault(v[0], []).append(n)
    for gi in got_order:
        assert sorted(got_order[gi]) == sorted(ref_order[gi])


@pytest.mark.gpu
def test_fused_adamw_matches_torch_adamw_and_refreshes_bf16_copies():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from get_model_amd import functional as Fn
    from tests.golden_util import load_case
    c = load_case("ft_L2_B2_N200_eval")
    batch = {k: v.cuda() for k, v in c["batch"].items()}

    def make():
        m = build_model(region_model_yaml("finetune", num_layers=2))
        m.load_state_dict(c["sd"])
        return m.cuda().eval()

    def loss_of(m):
        out = m(**m.get_input(batch))
        pred, obs = m.before_loss(out, batch)
        return m.after_loss(m.loss(pred, obs))

    ma, mb = make(), make()
    cfg = SimpleNamespace(opt="adamw", lr=3e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=[0.9, 0.999])
    oa = O.configure_optimizer(ma, cfg)
    groups_b = O.get_parameter_groups(mb, 0.05, mb.no_weight_decay())
    ob = torch.optim.AdamW(groups_b, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    for step in range(3):
        for m, o in ((ma, oa), (mb, ob)):
            o.zero_grad(set_to_none=True)
            loss_of(m).backward()
        # same gradients into both optimizers (the two models differ by rounding after the first step)
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            pb.grad = pa.grad.clone()
            pb.data.copy_(pa.data) if step == 0 else None
        oa.step()
        ob.step()
        for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
            torch.testing.assert_close(pa, pb, rtol=2e-6, atol=2e-7, msg=lambda s, n=n: f"{n} step {step}: {s}")
            pb.data.copy_(pa.data)          # keep the two trajectories on the same point
        for pa_state, pb_state in zip(oa.state.values(), ob.state.values()):
            pass
    # the kernel rewrote the cached bf16 operand copies: they equal a cast of the updated masters
    w = ma.encoder.blocks[1].mlp.fc1.weight
    wn, wt = Fn.WEIGHTS.peek(w, "n"), Fn.WEIGHTS.peek(w, "t")
    assert wn is not None and wt is not None
    assert torch.equal(wn, w.detach().bfloat16()) and torch.equal(wt, w.detach().t().contiguous().bfloat16())
    # and the next forward uses them: equal to a fresh model loaded from the updated state
    with torch.no_grad():
        l1 = loss_of(ma)
    fresh = build_model(region_model_yaml("finetune", num_layers=2))
    fresh.load_state_dict({k: v.cpu() for k, v in ma.state_dict().items()})
    fresh = fresh.cuda().eval()
    with torch.no_grad():
        l2 = loss_of(fresh)
    assert torch.equal(l1, l2)
    # optimizer state uses torch's names
    st = next(iter(oa.state.values()))
    assert set(st) == {"step", "exp_avg", "exp_avg_sq"} and float(st["step"]) == 3.0


@pytest.mark.gpu
def test_fused_adamw_load_state_dict_and_uneven_step_counts():
    """(a) after load_state_dict() the cached job table must point at the NEW moment tensors (parameter
    and gradient addresses do not change, so they alone cannot key the cache); (b) bias correction follows each
    parameter's own step count (torch.optim.AdamW), e.g. for a parameter that had no gradient in earlier steps."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.manual_seed(3)
    shapes = [(64, 128), (128,), (192, 64), (7,)]
    pa = [torch.nn.Parameter(torch.randn(*s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = O.FusedAdamW([{"params": pa[:2], "weight_decay": 0.05}, {"params": pa[2:], "weight_decay": 0.0}], lr=1e-2)
    ob = torch.optim.AdamW([{"params": pb[:2], "weight_decay": 0.05}, {"params": pb[2:], "weight_decay": 0.0}], lr=1e-2)
    grads = [[torch.randn_like(p) for p in pa] for _ in range(5)]

    def step(i, skip=()):
        for k, (a, b) in enumerate(zip(pa, pb)):
            if k in skip:
                a.grad = b.grad = None
            else:
                # gradients at fixed addresses, as with ddp.GradBuckets: only the moment tensors change below
                if a.grad is None:
                    a.grad = grads[i][k].clone()
                else:
                    a.grad.copy_(grads[i][k])
                b.grad = grads[i][k].clone()
        oa.step()
        ob.step()
        for k, (a, b) in enumerate(zip(pa, pb)):
            torch.testing.assert_close(a, b, rtol=3e-6, atol=3e-7, msg=lambda s, k=k: f"param {k} step {i}: {s}")

    step(0, skip=(1, 3))        # parameters 1 and 3 start one step late: their bias correction lags by one
    step(1)
    snap_a, snap_b = oa.state_dict(), ob.state_dict()
    import copy
    snap_a, snap_b = copy.deepcopy(snap_a), copy.deepcopy(snap_b)
    weights = [p.detach().clone() for p in pa]
    step(2)
    step(3)
    # roll back: new moment tensors, same parameter / gradient addresses
    oa.load_state_dict(snap_a)
    ob.load_state_dict(snap_b)
    with torch.no_grad():
        for a, b, w in zip(pa, pb, weights):
            a.copy_(w)
            b.copy_(w)
    step(2)
    step(4)
    for a in pa:        # the restored moments are the ones being updated
        st = oa.state[a]
        assert st["exp_avg"].abs().sum().item() > 0
    # state dicts are interchangeable with torch.optim.AdamW's
    for sa, sb in zip(oa.state_dict()["state"].values(), ob.state_dict()["state"].values()):
        torch.testing.assert_close(sa["exp_avg"], sb["exp_avg"], rtol=3e-6, atol=1e-7)
        assert float(sa["step"]) == float(sb["step"])
