"""Validation / inference consumers of the region models.

`validation_step` restates `RegionLitModel.validation_step` (run_region.py:121-141): optional TSS masking of
pred/obs (`batch['mask'] > 0`), the three early-outs, then `model.metrics`.
`predict_gene_expression` restates the "inference" branch of `predict_step` (run_region.py:179-210) as ONE batched
device reduction: the reference loops over the batch in Python and calls `.item()` five times per gene (a
host-device sync each), which cannot keep up with a ~4 M regions/s forward.  Plain tensor ops, any device.
"""
import torch


def shared_step(model, batch):
    """run.py:210-224 without logging: -> (loss, pred dict, obs dict)."""
    out = model(**model.get_input(batch))
    pred, obs = model.before_loss(out, batch)
    loss = model.after_loss(model.loss(pred, obs))
    return loss, pred, obs


@torch.no_grad()
def validation_step(model, batch, eval_tss=True):
    """-> (loss, metrics dict) or (loss, None) where the reference returns early (no TSS in the batch, all TSS
    unexpressed, NaNs)."""
    loss, pred, obs = shared_step(model, batch)
    pred, obs = dict(pred), dict(obs)
    if eval_tss and "exp" in pred:
        tss = batch["mask"]
        for key in pred:
            pred[key] = pred[key][tss > 0].flatten()
            obs[key] = obs[key][tss > 0].flatten()
        if pred["exp"].shape[0] == 0 or obs["exp"].sum() == 0:
            return loss, None
        if torch.isnan(pred["exp"]).any() or torch.isnan(obs["exp"]).any():
            return loss, None
    return loss, model.metrics(pred, obs)


@torch.no_grad()
def predict_gene_expression(preds, obs, batch):
    """Per batch element: max over the gene's TSS peaks (all_tss_peak > 0; 0 is padding) of pred / obs on the gene's
    strand column, and max aTPM (last input feature) over the same peaks.

    preds / obs: {key: [B, N, 2]}; batch: region_motif [B, N, F], all_tss_peak [B, P] int, strand [B] int (0/1),
    gene_name list.  -> list of dict rows in the reference's order (element-major, key-minor) with python floats;
    one device->host copy per key.  Elements without any TSS peak give -inf (the reference would raise there)."""
    peaks = batch["all_tss_peak"].long()
    B = peaks.shape[0]
    valid = peaks > 0
    idx = peaks.clamp_min(0)
    strand = torch.as_tensor(batch["strand"], device=peaks.device).long().view(B, 1, 1)
    ninf = float("-inf")

    def gene_max(x):                    # x [B, N, 2] -> [B]
        col = x.gather(2, strand.expand(B, x.shape[1], 1)).squeeze(2)        # [B, N] strand column
        return col.gather(1, idx).masked_fill(~valid, ninf).max(dim=1).values

    atpm = batch["region_motif"][:, :, -1].gather(1, idx).masked_fill(~valid, ninf).max(dim=1).values.cpu().tolist()
    cols = {key: (gene_max(preds[key]).cpu().tolist(), gene_max(obs[key]).cpu().tolist()) for key in preds}
    rows = []
    for b in range(B):
        for key in preds:
            rows.append({"gene_name": batch["gene_name"][b], "key": key, "pred": cols[key][0][b],
                         "obs": cols[key][1][b], "atpm": atpm[b]})
    return rows
