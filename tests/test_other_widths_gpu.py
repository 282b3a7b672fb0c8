"""GPU: the model classes are configurable (model.cfg: embed_dim, num_heads, num_layers, num_motif); the shipped YAMLs use
768 / 12 / 283, but a drop-in must either run the other widths the kernels support (embed_dim 256 / 512 / 1024 with
64-wide heads, other motif counts, one layer) correctly or refuse loudly.  Forward, loss and all gradients against the
fp32 oracle at the tolerances of tests/test_model_parity_gpu.py."""
import pytest
import torch
import torch.nn.functional as F

from tests.golden_util import grouped

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,D,H,L,motif", [("finetune", 256, 4, 2, 283), ("finetune", 512, 8, 1, 283),
                                              ("finetune", 1024, 16, 2, 283), ("atac", 512, 8, 2, 283),
                                              ("pretrain", 256, 4, 2, 283), ("finetune", 768, 12, 1, 100),
                                              ("pretrain", 768, 12, 1, 640)])
def test_other_widths_match_oracle(kind, D, H, L, motif):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import get_cpu as O
    from get_model_amd.config import build_model, region_model_yaml
    B, N = 3, 50
    sd = O.fill_state_dict(kind, seed=9, num_layers=L, embed_dim=D, num_motif=motif)
    batch = O.make_batch(B, N, 31, num_motif=motif, kind=kind)
    model = build_model(region_model_yaml(kind, num_layers=L, embed_dim=D, num_heads=H, num_motif=motif))
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    dev = {k: v.cuda() for k, v in batch.items()}
    out = model(**model.get_input(dev))
    pred, obs = model.before_loss(out, dev)
    loss = model.after_loss(model.loss(pred, obs))
    loss.backward()
    p = pred[list(pred.keys())[0]].detach().float().cpu()
    grads = {n: q.grad for n, q in model.named_parameters() if q.grad is not None}
    torch.set_num_threads(16)
    oloss, opred, ograds = O.loss_and_grads(kind, sd, batch, num_layers=L, num_heads=H, drop_path_rate=0.0)
    perr = (p.reshape(opred.shape) - opred).abs().max().item() / opred.abs().max().item()
    print(f"MEASURED other width: pred err {perr:.2e}, loss err {abs(loss.item() - oloss.item()) / abs(oloss.item()):.2e}")
    assert perr <= 2e-2, f"pred err {perr}"
    assert abs(loss.item() - oloss.item()) <= 1e-2 * abs(oloss.item()), (loss.item(), oloss.item())
    gg, og = grouped(grads), grouped({k: v for k, v in ograds.items() if k in grads})
    for n, g in gg.items():
        cos = F.cosine_similarity(g.flatten().float().cpu(), og[n].flatten(), dim=0).item()
        assert cos >= 0.998, f"{n}: cosine {cos}"


def test_unsupported_head_width_is_refused():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import get_cpu as O
    from get_model_amd.config import build_model, region_model_yaml
    batch = {k: v.cuda() for k, v in O.make_batch(2, 20, 1).items()}
    with pytest.raises((RuntimeError, NotImplementedError, ValueError)):      # refused when the model is built
        model = build_model(region_model_yaml("finetune", num_layers=1, embed_dim=768, num_heads=8)).cuda().eval()   # 96-wide heads
        model(**model.get_input(batch))
        torch.cuda.synchronize()
