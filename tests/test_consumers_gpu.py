"""GPU: the validation / inference consumers driven by the real HIP model on device tensors, against the
reference's loops restated literally (run_region.py:121-141 validation_step, :183-210 predict_step "inference"), and the
multi-rank gradient exchange with the real model (two ranks sharing this box's GPU over gloo: the code path of
bench.py --gpus 2 minus RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch

from tests.golden_util import load_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(kind, L, sd):
    from get_model_amd.config import build_model, region_model_yaml
    m = build_model(region_model_yaml(kind, num_layers=L))
    m.load_state_dict(sd)
    return m.cuda().eval()


def test_validation_step_on_the_device_matches_the_reference_loop():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scipy.stats import pearsonr, spearmanr
    from get_model_amd import evaluate as E
    c = load_case("ft_L2_B2_N200_eval")
    model = _model("finetune", 2, c["sd"])
    batch = {k: v.cuda() for k, v in c["batch"].items()}
    g = torch.Generator().manual_seed(2)
    tss = (torch.rand(2, 200, 2, generator=g) < 0.15).int()          # dataset key 'mask' = TSS indicator per strand
    batch["mask"] = tss.cuda()
    loss, met = E.validation_step(model, batch)
    # run_region.py:121-141 literally, on the host, from the model's own prediction
    with torch.no_grad():
        out = model(**model.get_input(batch))
    pred = out.float().cpu()[tss > 0].flatten().numpy()
    obs = c["batch"]["exp_label"][tss > 0].flatten().numpy()
    assert abs(float(met["exp_pearson"]) - pearsonr(pred, obs).statistic) < 1e-4
    assert abs(float(met["exp_spearman"]) - spearmanr(pred, obs).statistic) < 1e-4
    r2 = 1 - ((obs - pred) ** 2).sum() / ((obs - obs.mean()) ** 2).sum()
    assert abs(float(met["exp_r2"]) - r2) < 1e-4
    assert abs(float(loss) - float(c["z"]["loss"])) <= 1e-2 * abs(float(c["z"]["loss"]))
    assert all(v.is_cuda for v in met.values())                      # metrics stay on the device (no per-step sync)
    batch["mask"] = torch.zeros_like(batch["mask"])
    assert E.validation_step(model, batch)[1] is None                # "no TSS in the batch" early-out


def test_predict_gene_expression_on_the_device_matches_the_per_sample_loop():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from get_model_amd import evaluate as E
    c = load_case("ft_L2_B2_N200_eval")
    model = _model("finetune", 2, c["sd"])
    B, N, P = 6, 200, 8
    from oracle import get_cpu as O
    hb = O.make_batch(B, N, 12)
    g = torch.Generator().manual_seed(3)
    peaks = torch.randint(1, N, (B, P), generator=g)
    peaks[:, 5:] = 0
    peaks[1, 1:] = 0
    hb["all_tss_peak"] = peaks
    hb["strand"] = torch.tensor([0, 1, 1, 0, 1, 0])
    batch = {k: v.cuda() for k, v in hb.items()}
    batch["gene_name"] = [f"gene{i}" for i in range(B)]
    with torch.inference_mode():
        _, preds, obs = E.shared_step(model, batch)
        rows = E.predict_gene_expression(preds, obs, batch)
    ref = []                                                   # run_region.py:187-208
    for b in range(B):
        goi = batch["all_tss_peak"][b]
        goi = goi[goi > 0]
        s = batch["strand"][b]
        atpm = batch["region_motif"][b][goi, -1].max().item()
        for key in preds:
            ref.append((batch["gene_name"][b], key, preds[key][b][:, s][goi].max().item(),
                        obs[key][b][:, s][goi].max().item(), atpm))
    assert len(rows) == len(ref)
    for a, r in zip(rows, ref):
        assert (a["gene_name"], a["key"]) == r[:2]
        assert a["pred"] == r[2] and a["obs"] == r[3] and a["atpm"] == r[4]      # same elements, max is exact


def _rank_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from get_model_amd.config import build_model, region_model_yaml
    from get_model_amd.ddp import GradBuckets, broadcast_parameters
    from oracle import get_cpu as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.manual_seed(50 + rank)                 # different init per rank: broadcast_parameters must fix it
    model = build_model(region_model_yaml("finetune", num_layers=2)).cuda().eval()
    broadcast_parameters(model)
    buckets = GradBuckets(model)
    full = O.make_batch(4, 64, 21)
    mine = {k: v[rank * 2:(rank + 1) * 2].cuda() for k, v in full.items()}
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        out = model(**model.get_input(mine))
        pred, obs = model.before_loss(out, mine)
        model.after_loss(model.loss(pred, obs)).backward()
        buckets.finish()
    torch.cuda.synchronize()
    torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()},
                "grads": {n: p.grad.cpu() for n, p in model.named_parameters()}}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_real_model_bucketed_exchange(tmp_path):
    """World size 2 on ONE GPU (gloo, host-staged buckets): both ranks end with the same weights and the same gradient,
    equal to the gradient of the mean of the two half-batch losses computed by one process."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=150)
        assert p.exitcode == 0
    r0 = torch.load(str(tmp_path / "r0.pt"), weights_only=True)
    r1 = torch.load(str(tmp_path / "r1.pt"), weights_only=True)
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
    for n in r0["grads"]:
        assert torch.equal(r0["grads"][n], r1["grads"][n]), n
    from oracle import get_cpu as O
    model = _model("finetune", 2, r0["sd"])
    full = {k: v.cuda() for k, v in O.make_batch(4, 64, 21).items()}
    total = 0
    for r in range(2):
        b = {k: v[r * 2:(r + 1) * 2] for k, v in full.items()}
        out = model(**model.get_input(b))
        pred, obs = model.before_loss(out, b)
        total = total + 0.5 * model.after_loss(model.loss(pred, obs))
    total.backward()
    for n, p in model.named_parameters():
        torch.testing.assert_close(p.grad.cpu(), r0["grads"][n], rtol=2e-5, atol=1e-7, msg=lambda s, n=n: f"{n}: {s}")
