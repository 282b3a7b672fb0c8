"""Validation / inference consumers against a literal restatement of the reference's loops."""
import torch

from get_model_amd import evaluate as E


def test_predict_gene_expression_matches_the_per_sample_loop():
    g = torch.Generator().manual_seed(0)
    B, N, P = 5, 40, 6
    preds = {"exp": torch.rand(B, N, 2, generator=g)}
    obs = {"exp": torch.rand(B, N, 2, generator=g)}
    peaks = torch.randint(1, N, (B, P), generator=g)
    peaks[:, 4:] = 0                                    # padding
    peaks[2, 1:] = 0
    batch = {"region_motif": torch.rand(B, N, 283, generator=g), "all_tss_peak": peaks,
             "strand": torch.tensor([0, 1, 1, 0, 1]), "gene_name": [f"g{i}" for i in range(B)]}
    rows = E.predict_gene_expression(preds, obs, batch)
    # run_region.py:183-208, literally
    ref = []
    for b in range(B):
        goi = batch["all_tss_peak"][b]
        goi = goi[goi > 0]
        s = batch["strand"][b]
        atpm = batch["region_motif"][b][goi, -1].max().item()
        for key in preds:
            ref.append({"gene_name": batch["gene_name"][b], "key": key, "pred": preds[key][b][:, s][goi].max().item(),
                        "obs": obs[key][b][:, s][goi].max().item(), "atpm": atpm})
    assert len(rows) == len(ref)
    for a, r in zip(rows, ref):
        assert a["gene_name"] == r["gene_name"] and a["key"] == r["key"]
        for k in ("pred", "obs", "atpm"):
            assert abs(a[k] - r[k]) < 1e-7


class _Toy(torch.nn.Module):
    """stands for the region model on CPU: identity 'forward' on a stored prediction"""

    def __init__(self, pred):
        super().__init__()
        self.pred = pred
        from get_model_amd.model.model import RegressionMetrics
        from get_model_amd.config import AttrDict
        self.metrics = RegressionMetrics(AttrDict({"components": {"exp": ["pearson", "spearman", "r2"]}}))

    def get_input(self, batch):
        return {}

    def forward(self):
        return self.pred

    def before_loss(self, out, batch):
        return {"exp": out}, {"exp": batch["exp_label"]}

    def loss(self, pred, obs):
        return {"exp_loss": ((pred["exp"] - obs["exp"]) ** 2).mean()}

    def after_loss(self, loss):
        return sum(loss.values())


def test_validation_step_tss_masking_and_early_outs():
    g = torch.Generator().manual_seed(1)
    obs = torch.rand(3, 20, 2, generator=g)
    pred = obs + 0.05 * torch.randn(3, 20, 2, generator=g)
    mask = (torch.rand(3, 20, 2, generator=g) < 0.3).int()
    m = _Toy(pred)
    loss, met = E.validation_step(m, {"exp_label": obs, "mask": mask})
    sel_p, sel_o = pred[mask > 0], obs[mask > 0]
    pear = torch.corrcoef(torch.stack([sel_p, sel_o]))[0, 1]
    assert abs(float(met["exp_pearson"]) - float(pear)) < 1e-5 and set(met) == {"exp_pearson", "exp_spearman", "exp_r2"}
    assert E.validation_step(m, {"exp_label": obs, "mask": torch.zeros_like(mask)})[1] is None          # no TSS
    assert E.validation_step(m, {"exp_label": torch.zeros_like(obs), "mask": mask})[1] is None          # unexpressed
    assert E.validation_step(_Toy(pred * float("nan")), {"exp_label": obs, "mask": mask})[1] is None    # NaN
    _, met_all = E.validation_step(m, {"exp_label": obs, "mask": mask}, eval_tss=False)
    assert met_all is not None


def test_spearman_uses_average_ranks_for_ties():
    """torchmetrics SpearmanCorrCoef (model/model.py:173-174) ranks ties by the mean of their positions; expression
    labels are mostly tied zeros.  torchmetrics is not installed here: scipy.stats.spearmanr has the same definition."""
    import numpy as np
    from scipy.stats import rankdata, spearmanr
    from get_model_amd.model.model import RegressionMetrics
    g = torch.Generator().manual_seed(4)
    obs = torch.rand(500, generator=g) * 4
    obs[torch.rand(500, generator=g) < 0.7] = 0.0                 # 70 % tied zeros
    obs[10:20] = 1.5                                              # and a second run of ties
    pred = (obs + 0.3 * torch.randn(500, generator=g)).clamp_min(0).round(decimals=1)     # ties in pred too
    r = RegressionMetrics._rank(obs)
    assert np.array_equal(r.numpy(), rankdata(obs.numpy()).astype(np.float32))
    m = RegressionMetrics({"components": {"exp": ["spearman", "pearson"]}})
    got = float(m({"exp": pred}, {"exp": obs})["exp_spearman"])
    assert abs(got - spearmanr(pred.numpy(), obs.numpy()).statistic) < 1e-5
    # order independence (argsort-of-argsort ranks depended on the order of the tied elements)
    perm = torch.randperm(500, generator=g)
    assert abs(float(m({"exp": pred[perm]}, {"exp": obs[perm]})["exp_spearman"]) - got) < 1e-6
