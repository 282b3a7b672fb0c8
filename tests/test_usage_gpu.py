"""GPU: unusual but legal uses of the model classes that a reference user may rely on (each checked against the oracle or
against the reference's documented behaviour): a pre-training batch with every / with no region masked (the latter gives the
reference's NaN of a mean over nothing, not an error), two forwards before one backward, gradient accumulation over two
backward passes, non-contiguous and float64 batch tensors, train mode under no_grad, and a CPU batch (refused loudly: the
product path has no CPU fallback)."""
import pytest
import torch

from oracle import get_cpu as O
from get_model_amd.config import build_model, region_model_yaml

pytestmark = pytest.mark.gpu


def mk(kind, L=2):
    sd = O.fill_state_dict(kind, seed=5, num_layers=L)
    m = build_model(region_model_yaml(kind, num_layers=L)); m.load_state_dict(sd); return m.cuda().eval(), sd

def step(m, batch):
    out = m(**m.get_input(batch)); pred, obs = m.before_loss(out, batch); return m.after_loss(m.loss(pred, obs)), pred


def p_mask_all(value):
    def f():
        m, sd = mk("pretrain"); b = O.make_batch(2, 30, 1, kind="pretrain"); b["mask"][:] = value
        dev = {k: v.cuda() for k, v in b.items()}
        loss, pred = step(m, dev)
        if value == 0:
            # the reference's mean-reduced MSELoss over no rows is NaN (model/model.py:527-536 with an all-zero mask);
            # the output has no rows, backward runs without raising and leaves zero (or no) gradients
            if not torch.isnan(loss):
                return f"loss over no rows is {float(loss.detach())}, expected NaN"
            if tuple(pred["masked"].shape) != (0, 283):
                return f"output shape {tuple(pred['masked'].shape)}"
            loss.backward()
            bad = [n for n, p in m.named_parameters() if p.grad is not None and bool((p.grad != 0).any())]
            return True if not bad else "non-zero gradients from an empty loss: " + str(bad[:4])
        loss.backward(); ol, _, _ = O.loss_and_grads("pretrain", sd, b, num_layers=2, drop_path_rate=0.0)
        return abs(float(loss.detach()) - float(ol)) <= 1e-2 * abs(float(ol)) or f"{float(loss.detach())} vs {float(ol)}"
    return f

def p_two_forwards_one_backward():
    m, sd = mk("finetune"); b1 = O.make_batch(2, 30, 1); b2 = O.make_batch(2, 30, 2)
    d1 = {k: v.cuda() for k, v in b1.items()}; d2 = {k: v.cuda() for k, v in b2.items()}
    (step(m, d1)[0] + step(m, d2)[0]).backward()
    _, _, g1 = O.loss_and_grads("finetune", sd, b1, num_layers=2, drop_path_rate=0.0)
    _, _, g2 = O.loss_and_grads("finetune", sd, b2, num_layers=2, drop_path_rate=0.0)
    bad = []; worst = 1.0
    for n, p in m.named_parameters():
        ref = g1[n] + g2[n]
        if float(ref.norm()) < 1e-9: continue
        cos = torch.nn.functional.cosine_similarity(p.grad.flatten().cpu(), ref.flatten(), dim=0).item()
        rel = float(p.grad.cpu().norm() / ref.norm())
        worst = min(worst, cos)
        if cos < 0.998 or abs(rel - 1) > 0.03: bad.append((n, round(cos, 4), round(rel, 3)))
    print(f"MEASURED two_forwards_one_backward: min cosine {worst:.5f}")
    return True if not bad else str(bad[:4])

def p_accumulate_two_backwards():
    m, sd = mk("finetune"); b1 = O.make_batch(2, 30, 1); d1 = {k: v.cuda() for k, v in b1.items()}
    step(m, d1)[0].backward(); g_once = {n: p.grad.clone() for n, p in m.named_parameters()}
    step(m, d1)[0].backward()
    bad = [n for n, p in m.named_parameters() if not torch.allclose(p.grad, 2 * g_once[n], rtol=1e-3, atol=1e-6 * float(g_once[n].abs().max()) + 1e-12)]
    return True if not bad else str(bad[:5])

def p_noncontiguous_input():
    m, sd = mk("finetune"); b = O.make_batch(2, 60, 3)
    rm = b["region_motif"].cuda()[:, ::2]; lab = b["exp_label"].cuda()[:, ::2]
    loss, _ = step(m, {"region_motif": rm, "exp_label": lab, "mask": b["mask"].cuda()[:, ::2]})
    b2 = {"region_motif": b["region_motif"][:, ::2].contiguous(), "exp_label": b["exp_label"][:, ::2].contiguous(), "mask": b["mask"][:, ::2].contiguous()}
    ol, _, _ = O.loss_and_grads("finetune", sd, b2, num_layers=2, drop_path_rate=0.0)
    return abs(float(loss.detach()) - float(ol)) <= 1e-2 * abs(float(ol)) or f"{float(loss.detach())} vs {float(ol)}"

def p_no_grad_train_mode():
    m, sd = mk("finetune"); m.train(); b = {k: v.cuda() for k, v in O.make_batch(2, 30, 1).items()}
    with torch.no_grad():
        loss, _ = step(m, b)
    return bool(torch.isfinite(loss))

def p_cpu_input_refused():
    m, sd = mk("finetune"); b = O.make_batch(2, 30, 1)
    try:
        step(m, b)
    except (RuntimeError, ValueError, TypeError):
        return True
    return "CPU batch was accepted"

def p_double_input():
    m, sd = mk("finetune"); b = O.make_batch(2, 30, 1); d = {k: (v.double() if v.is_floating_point() else v).cuda() for k, v in b.items()}
    loss, _ = step(m, d); ol, _, _ = O.loss_and_grads("finetune", sd, b, num_layers=2, drop_path_rate=0.0)
    return abs(float(loss.detach()) - float(ol)) <= 1e-2 * abs(float(ol)) or f"{float(loss.detach())} vs {float(ol)}"


@pytest.mark.parametrize("probe", [p_mask_all(1), p_mask_all(0), p_two_forwards_one_backward, p_accumulate_two_backwards,
                                   p_noncontiguous_input, p_no_grad_train_mode, p_cpu_input_refused, p_double_input],
                         ids=["all_masked", "none_masked", "two_forwards_one_backward", "accumulate_two_backwards",
                              "noncontiguous", "no_grad_train_mode", "cpu_batch_refused", "float64_batch"])
def test_usage(probe):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = probe()
    assert r is True or r is None, r


def test_autocast_context_changes_nothing():
    """A torch.autocast context around the step (the reference's own mixed-precision mode is fp16 autocast + GradScaler,
    utils.py:100-102: tests/test_round3_gpu.py; bf16 here is what a user of this build would pick) must change nothing:
    the fused path computes in bf16 with fp32 accumulation whatever the context says."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    m, sd = mk("finetune")
    d = {k: v.cuda() for k, v in O.make_batch(2, 30, 1).items()}
    l0, _ = step(m, d)
    l0.backward()
    g0 = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l1, _ = step(m, d)
    l1.backward()
    assert l1.dtype == torch.float32 and float(l1.detach()) == float(l0.detach())
    for n, p in m.named_parameters():
        assert torch.equal(p.grad, g0[n]), n


def test_torch_ddp_wrapper_on_rccl_world_of_one():
    """The reference's harness wraps the model in torch's DistributedDataParallel (utils.py:41-51, via Lightning): the fused
    autograd functions must work under it (its reducer hooks, its bucket views adopted as .grad) for two consecutive steps."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    import torch.distributed as dist
    import torch.nn.functional as F
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29541"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        m, sd = mk("finetune")
        b = O.make_batch(2, 30, 1)
        d = {k: v.cuda() for k, v in b.items()}
        ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True)
        torch.set_num_threads(16)
        _, _, og = O.loss_and_grads("finetune", sd, b, num_layers=2, drop_path_rate=0.0)
        for it in range(2):
            ddp.zero_grad(set_to_none=(it == 0))
            out = ddp(**m.get_input(d))
            pred, obs = m.before_loss(out, d)
            m.after_loss(m.loss(pred, obs)).backward()
            for n, p in m.named_parameters():
                if float(og[n].norm()) < 1e-9:
                    continue
                assert p.grad is not None, n
                cos = F.cosine_similarity(p.grad.flatten().cpu(), og[n].flatten(), dim=0).item()
                rel = float(p.grad.cpu().norm() / og[n].norm())
                assert cos >= 0.998 and abs(rel - 1) <= 0.03, (it, n, cos, rel)
    finally:
        dist.destroy_process_group()
