"""Region-batch producer: the dataset rules restated literally, and the device-side batch gather
against per-item stacking.  CPU (the gather is plain tensor indexing; on the GPU the tables live in HBM)."""
import numpy as np
import torch

from get_model_amd import data as D


def test_region_motif_rules():
    st = D.make_synthetic_store(num_peaks=800, celltypes=("a",), seed=1)
    rm = D.RegionMotif(st, "a")
    keep = np.nonzero(st["atpm/a"])[0]                       # drop_zero_atpm
    assert rm.num_peaks == len(keep) and np.array_equal(rm.atpm, st["atpm/a"][keep])
    raw = st["data"][keep]
    ref = raw / raw.max(axis=0)                              # per-column max over the KEPT peaks, clipped to 1
    ref[ref > 1] = 1
    assert np.allclose(rm.normalized_data, ref) and rm.normalized_data.max() <= 1.0
    low = rm.atpm < D.EXPRESSION_ATAC_CUTOFF                 # expression zeroed where aTPM < 0.05
    assert low.any() and (rm.expression_positive[low] == 0).all() and (rm.expression_negative[low] == 0).all()
    assert np.array_equal(rm.expression_positive[~low], st["expression_positive/a"][keep][~low])
    assert rm.chromosome[0] == "chr1" and rm.end[0] - rm.start[0] == 400
    assert D._chromosome_splitter(["chr1", "chr2", "chr3"], "chr2,chr3", True) == ["chr1"]
    assert D._chromosome_splitter(["chr1", "chr2", "chr3"], "chr2,chr3", False) == ["chr2", "chr3"]
    assert D._chromosome_splitter(["chr1", "chr2"], None, False) == ["chr1", "chr2"]


def test_windows_and_items():
    st = D.make_synthetic_store(num_peaks=3000, celltypes=("a", "b"), seed=2)
    ds = D.RegionMotifDataset(st, celltypes="a,b", sampling_step=50, num_region_per_sample=200, seed=3,
                              quantitative_atac=True)
    assert len(ds) > 40
    for ct, s, e in ds.sample_indices:
        rm = ds.region_motifs[ct]
        assert e - s == 200 and e < rm.num_peaks
        assert 0 <= rm.end[e] - rm.start[s] <= D.MAX_WINDOW_BP
    it = ds[5]
    ct, s, e = ds.sample_indices[5]
    rm = ds.region_motifs[ct]
    assert it["region_motif"].shape == (200, 283) and it["region_motif"].dtype == np.float32
    assert np.allclose(it["region_motif"][:, -1], rm.atpm[s:e] / rm.atpm[s:e].max())
    assert np.allclose(it["region_motif"][:, :282], rm.normalized_data[s:e])
    assert it["exp_label"].shape == (200, 2) and np.array_equal(it["mask"], rm.tss[s:e])
    # leave-out cell types: train drops them, eval keeps only them
    tr = D.RegionMotifDataset(st, leave_out_celltypes="b", is_train=True, num_region_per_sample=200, seed=0)
    ev = D.RegionMotifDataset(st, leave_out_celltypes="b", is_train=False, num_region_per_sample=200, seed=0)
    assert list(tr.region_motifs) == ["a"] and list(ev.region_motifs) == ["b"]
    ones = D.RegionMotifDataset(st, celltypes="a", num_region_per_sample=200, seed=0)[0]["region_motif"][:, -1]
    assert (ones == 1).all()


def test_batch_producer_matches_per_item_stacking():
    st = D.make_synthetic_store(num_peaks=3000, celltypes=("a", "b"), seed=4)
    for quant in (False, True):
        ds = D.RegionMotifDataset(st, sampling_step=50, num_region_per_sample=200, seed=5, quantitative_atac=quant)
        prod = D.RegionBatchProducer(ds, batch_size=8, device="cpu", shuffle=True, seed=7, rank=1, world=2)
        order = prod._order()
        assert len(order) == len(ds) // 2          # rank 1 of 2 takes the odd positions of the shuffled list
        nb = 0
        for b, batch in enumerate(prod):
            ids = order[b * 8:(b + 1) * 8]
            items = [ds[int(i)] for i in ids]
            ref_motif = np.stack([it["region_motif"] for it in items])
            assert batch["region_motif"].shape == (8, 200, 283)
            assert np.allclose(batch["region_motif"].numpy(), ref_motif, atol=1e-6)
            assert np.array_equal(batch["exp_label"].numpy(), np.stack([it["exp_label"] for it in items]))
            assert np.array_equal(batch["mask"].numpy(), np.stack([it["mask"] for it in items]))
            assert np.allclose(batch["atpm"].numpy(), np.stack([it["atpm"] for it in items]))
            nb += 1
        assert nb == len(prod) and nb >= 2
    # pretrain masking: Bernoulli(mask_ratio) per region
    ds = D.RegionMotifDataset(st, num_region_per_sample=200, seed=5, mask_ratio=0.5)
    b = next(iter(D.RegionBatchProducer(ds, 16, "cpu", seed=1)))
    assert b["mask"].shape == (16, 200) and 0.4 < b["mask"].float().mean() < 0.6
    # different epochs reshuffle; ranks see disjoint samples
    p0 = D.RegionBatchProducer(ds, 4, "cpu", seed=1, rank=0, world=2)
    p1 = D.RegionBatchProducer(ds, 4, "cpu", seed=1, rank=1, world=2)
    assert not set(p0._order()) & set(p1._order())
    o0 = p0._order().copy(); list(p0); assert not np.array_equal(o0, p0._order())


import pytest  # noqa: E402


@pytest.mark.gpu
def test_producer_feeds_the_model_on_the_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from get_model_amd.config import build_model, region_model_yaml
    st = D.make_synthetic_store(num_peaks=3000, celltypes=("a",), seed=6)
    ds = D.RegionMotifDataset(st, num_region_per_sample=200, seed=1, quantitative_atac=True)
    prod = D.RegionBatchProducer(ds, batch_size=4, device="cuda", seed=2)
    cpu = D.RegionBatchProducer(ds, batch_size=4, device="cpu", seed=2)
    model = build_model(region_model_yaml("finetune", num_layers=2)).cuda().train()
    batch, ref = next(iter(prod)), next(iter(cpu))
    assert batch["region_motif"].is_cuda and torch.allclose(batch["region_motif"].cpu(), ref["region_motif"], atol=1e-6)
    out = model(**model.get_input(batch))
    pred, obs = model.before_loss(out, batch)
    loss = model.after_loss(model.loss(pred, obs))
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in model.parameters())


import glob
import os

import pytest

EXTRA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_extra")
_STORE_KW = {
    "data_two_celltypes": (dict(num_peaks=3000, celltypes=("a", "b"), seed=2),
                           dict(celltypes="a,b", sampling_step=50, num_region_per_sample=200, quantitative_atac=True)),
    "data_leave_out_chr": (dict(num_peaks=2400, celltypes=("a",), seed=6, chromosomes=("chr1", "chr2", "chr3")),
                           dict(celltypes="a", sampling_step=40, num_region_per_sample=100, leave_out_chromosomes="chr2",
                                is_train=True)),
    "data_eval_split_pretrain_mask": (dict(num_peaks=2400, celltypes=("a",), seed=6, chromosomes=("chr1", "chr2", "chr3")),
                                      dict(celltypes="a", sampling_step=40, num_region_per_sample=100,
                                           leave_out_chromosomes="chr2", is_train=False, mask_ratio=0.5)),
}


@pytest.mark.parametrize("name", sorted(_STORE_KW))
def test_dataset_equals_the_reference_dataset(name):
    """Fixtures = the reference's own RegionMotif / RegionMotifDataset (zarr_dataset.py:880-1185) run on the synthetic
    store (oracle/make_golden_data.py: zarr replaced by an in-memory group, everything else reference code): the same
    windows in the same order, the same normalisation, the same items, draw for draw under the same numpy seed."""
    z = np.load(os.path.join(EXTRA, name + ".npz"), allow_pickle=False)
    skw, dkw = _STORE_KW[name]
    st = D.make_synthetic_store(**skw)
    np.random.seed(int(z["seed"]))
    ds = D.RegionMotifDataset(st, seed="numpy-global", **dkw)
    cts = [str(c) for c in z["celltypes"]]
    assert list(ds.region_motifs) == cts
    assert len(ds) == int(z["n_samples"])
    got = [(cts.index(c), s, e) for c, s, e in ds.sample_indices]
    assert got == list(zip(z["sample_ct"].tolist(), z["sample_start"].tolist(), z["sample_end"].tolist()))
    for ct, rm in ds.region_motifs.items():
        assert np.array_equal(np.asarray(rm.normalizing_factor, dtype=np.float64), z[f"norm_factor/{ct}"])
        nd = rm.normalized_data
        chk = np.array([nd.shape[0], nd.sum(dtype=np.float64), (nd.astype(np.float64) ** 2).sum()])
        assert np.array_equal(chk, z[f"norm_chk/{ct}"])
        assert np.array_equal(np.array([rm.expression_positive.sum(dtype=np.float64),
                                        rm.expression_negative.sum(dtype=np.float64)]), z[f"exp_chk/{ct}"])
    for j, i in enumerate(z["item_ids"].tolist()):
        it = ds[i]
        for key in ("region_motif", "exp_label", "mask", "atpm", "peak_coord"):
            assert np.array_equal(np.asarray(it[key]), z[f"item{j}/{key}"]), (i, key)
        assert it["region_motif"].dtype == np.float32 and it["exp_label"].dtype == np.float32
        assert str(it["chromosome"]) == str(z[f"item{j}/chromosome"]) and it["celltype"] == str(z[f"item{j}/celltype"])
    # ... and the device-side batch gather gives exactly these items (deterministic keys)
    if dkw.get("mask_ratio", 0) == 0:
        prod = D.RegionBatchProducer(ds, batch_size=3, device="cpu", shuffle=False)
        batch = prod.gather(z["item_ids"].tolist())
        for j in range(3):
            assert np.allclose(batch["region_motif"][j].numpy(), z[f"item{j}/region_motif"], rtol=0, atol=1e-6)
            assert np.array_equal(batch["exp_label"][j].numpy(), z[f"item{j}/exp_label"])
            assert np.array_equal(batch["mask"][j].numpy(), z[f"item{j}/mask"])
