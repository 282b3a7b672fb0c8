"""Region-batch producer for the region models.

Restates the reference's `RegionMotif` / `RegionMotifDataset` (`get_model/dataset/zarr_dataset.py:880-1185`): per
cell type a peaks x 282 motif matrix, max-normalised per column and clipped to 1, peaks with zero aTPM dropped,
expression zeroed where aTPM < 0.05, samples = windows of N consecutive peaks every `sampling_step` peaks with a
random shift, at most 5 Mb long, never across the end of the table; each sample appends the ATAC column (ones, or
aTPM / max aTPM of the window when `quantitative_atac`).

zarr is not installed on the training hosts, so the on-disk form here is the same arrays under the same names in a
plain mapping (`np.load`-able .npz, or a directory of .npy): `data`, `peak_names`, `motif_names`, `atpm/<ct>`,
`expression_positive/<ct>`, `expression_negative/<ct>`, `tss/<ct>`.

MI355X-first batching (`RegionBatchProducer`): the whole normalised table of every cell type lives in HBM (1 M peaks
x 282 x 4 B = 1.1 GB of 288 GB), a batch is a device-side gather by window start -- the host sends B integers per
step instead of 58 MB of features, so the input side cannot starve a 1.3 M regions/s consumer.  The per-item
`__getitem__` path is kept for parity with the reference and for the tests.
"""
import numpy as np
import torch

EXPRESSION_ATAC_CUTOFF = 0.05        # zarr_dataset.py:21
MAX_WINDOW_BP = 5000000              # zarr_dataset.py:1109


def _chromosome_splitter(all_chromosomes, leave_out_chromosomes, is_train=True):
    """zarr_dataset.py:29-53: train = everything not left out; eval = exactly the left-out chromosomes."""
    if leave_out_chromosomes is None:
        leave_out = []
    elif isinstance(leave_out_chromosomes, str):
        leave_out = leave_out_chromosomes.split(",") if "," in leave_out_chromosomes else [leave_out_chromosomes]
    else:
        leave_out = list(leave_out_chromosomes)
    if is_train or leave_out in ([""], []):
        return [c for c in all_chromosomes if c not in leave_out]
    return leave_out


def make_synthetic_store(num_peaks=5000, celltypes=("ct0",), num_motifs=282, seed=0, chromosomes=("chr1", "chr2")):
    """A store with the reference's array names and value ranges: for tests and benchmarks."""
    rng = np.random.default_rng(seed)
    per = num_peaks // len(chromosomes)
    names = []
    for c in chromosomes:
        starts = np.cumsum(rng.integers(500, 20000, size=per))
        names += [f"{c}:{s}-{s + 400}" for s in starts]
    n = len(names)
    store = {"data": (rng.random((n, num_motifs)) * (rng.random((n, num_motifs)) > 0.6) * 30).astype(np.float32),
             "peak_names": np.array(names), "motif_names": np.array([f"m{i}" for i in range(num_motifs)])}
    for ct in celltypes:
        atpm = rng.random(n).astype(np.float32) * (rng.random(n) > 0.1)
        tss = (rng.random((n, 2)) < 0.1).astype(np.int8)
        store[f"atpm/{ct}"] = atpm
        store[f"expression_positive/{ct}"] = (rng.random(n) * 4 * tss[:, 0]).astype(np.float32)
        store[f"expression_negative/{ct}"] = (rng.random(n) * 4 * tss[:, 1]).astype(np.float32)
        store[f"tss/{ct}"] = tss
    return store


class RegionMotif:
    """One cell type of one store (zarr_dataset.py:880-989)."""

    def __init__(self, store, celltype, motif_scaler=1.0, normalize=True, drop_zero_atpm=True):
        self.celltype = celltype
        self.motif_scaler = motif_scaler
        self.normalize = normalize
        self.data = np.asarray(store["data"])
        self.peak_names = np.asarray(store["peak_names"])
        self.motif_names = np.asarray(store["motif_names"])
        chrom, start, end = [], [], []
        for name in self.peak_names:                      # "chr:start-end"
            c, span = str(name).split(":")
            s, e = span.split("-")
            chrom.append(c)
            start.append(int(s))
            end.append(int(e))
        self.chromosome, self.start, self.end = np.array(chrom), np.array(start), np.array(end)
        self.atpm = np.asarray(store[f"atpm/{celltype}"])
        keep = np.nonzero(self.atpm)[0] if drop_zero_atpm else np.arange(len(self.atpm))
        self.kept_index = keep
        self.atpm, self.data, self.peak_names = self.atpm[keep], self.data[keep], self.peak_names[keep]
        self.chromosome, self.start, self.end = self.chromosome[keep], self.start[keep], self.end[keep]
        self.has_expression = f"expression_positive/{celltype}" in store
        if self.has_expression:
            self.expression_positive = np.array(store[f"expression_positive/{celltype}"])[keep]
            self.expression_negative = np.array(store[f"expression_negative/{celltype}"])[keep]
            self.tss = np.asarray(store[f"tss/{celltype}"])[keep]
            low = self.atpm < EXPRESSION_ATAC_CUTOFF
            self.expression_positive[low] = 0
            self.expression_negative[low] = 0
        self._normalized = None

    @property
    def num_peaks(self):
        return len(self.peak_names)

    @property
    def normalizing_factor(self):
        return self.data.max(axis=0) * self.motif_scaler if self.normalize else 1

    @property
    def normalized_data(self):
        if not self.normalize:
            return self.data
        if self._normalized is None:
            d = self.data / self.normalizing_factor
            d[d > 1] = 1
            self._normalized = d
        return self._normalized


class RegionMotifDataset:
    """zarr_dataset.py:997-1185 over in-memory stores.  `stores`: one mapping or a list of them."""

    def __init__(self, stores, celltypes=None, quantitative_atac=False, sampling_step=50, num_region_per_sample=1000,
                 leave_out_chromosomes=None, leave_out_celltypes=None, is_train=True, mask_ratio=0.0,
                 drop_zero_atpm=True, seed=None):
        """seed: None / int -> a private numpy Generator (reproducible per data set); "numpy-global" -> numpy's global
        stream through the reference's own calls (np.random.randint in setup, np.random.choice per item), which
        reproduces the reference's windows and masks draw for draw under the same np.random.seed (the parity tests)."""
        stores = stores if isinstance(stores, (list, tuple)) else [stores]
        available = []
        for st in stores:
            available += [k.split("/", 1)[1] for k in st if k.startswith("atpm/")]
        wanted = celltypes.split(",") if isinstance(celltypes, str) else (list(celltypes) if celltypes else available)
        self.celltypes = [c for c in wanted if c in available]
        self.quantitative_atac = quantitative_atac
        self.sampling_step = sampling_step
        self.num_region_per_sample = num_region_per_sample
        self.leave_out_chromosomes = leave_out_chromosomes if leave_out_chromosomes else []
        leave_ct = leave_out_celltypes.split(",") if leave_out_celltypes else []
        self.is_train = is_train
        self.mask_ratio = mask_ratio
        self.rng = None if seed == "numpy-global" else np.random.default_rng(seed)
        self.region_motifs = {}
        for ct in self.celltypes:
            if leave_ct and ((is_train and ct in leave_ct) or (not is_train and ct not in leave_ct)):
                continue
            st = next(s for s in stores if f"atpm/{ct}" in s)
            self.region_motifs[ct] = RegionMotif(st, ct, drop_zero_atpm=drop_zero_atpm)
        self.sample_indices = []
        self.setup()

    def setup(self):
        n, step = self.num_region_per_sample, self.sampling_step
        for ct, rm in self.region_motifs.items():
            chroms = list(dict.fromkeys(rm.chromosome.tolist()))
            for chrom in _chromosome_splitter(chroms, self.leave_out_chromosomes, self.is_train):
                idx = np.nonzero(rm.chromosome == chrom)[0]
                if len(idx) == 0:
                    continue
                # the reference walks the ORIGINAL row labels of the chromosome's first / last kept peak (its DataFrame keeps
                # the pre-drop index, zarr_dataset.py:912,1095-1099) and then uses them as positions in the kept table
                for i in range(int(rm.kept_index[idx[0]]), int(rm.kept_index[idx[-1]]), step):
                    shift = int(np.random.randint(-step // 2, step // 2) if self.rng is None
                                else self.rng.integers(-step // 2, step // 2))
                    start = max(0, i + shift)
                    end = start + n
                    if end >= rm.num_peaks:
                        continue
                    span = rm.end[end] - rm.start[start]
                    if span > MAX_WINDOW_BP or span < 0:
                        continue
                    self.sample_indices.append((ct, start, end))

    def __len__(self):
        return len(self.sample_indices)

    def __getitem__(self, index):
        ct, s, e = self.sample_indices[index]
        rm = self.region_motifs[ct]
        motif = rm.normalized_data[s:e]
        atpm = rm.atpm[s:e]
        if rm.has_expression:
            pos, neg, tss = rm.expression_positive[s:e], rm.expression_negative[s:e], rm.tss[s:e]
        else:
            pos = neg = tss = np.zeros(e - s)
        col = atpm.reshape(-1, 1) / atpm.max() if self.quantitative_atac else np.ones((e - s, 1))
        if self.mask_ratio > 0:
            draw = np.random.choice if self.rng is None else self.rng.choice
            mask = draw([0, 1], size=len(tss), p=[1 - self.mask_ratio, self.mask_ratio])
        else:
            mask = tss
        return {"region_motif": np.concatenate([motif, col], axis=1).astype(np.float32), "mask": mask,
                "atpm": atpm.reshape(-1, 1), "chromosome": rm.chromosome[s],
                "peak_coord": np.stack([rm.start[s:e], rm.end[s:e]], 1),
                "exp_label": np.column_stack((pos, neg)).astype(np.float32), "celltype": ct}


class RegionBatchProducer:
    """Batches of the dataset as device tensors, gathered ON the device from HBM-resident tables.

    Yields {"region_motif" f32 [B,N,F+1], "exp_label" f32 [B,N,2], "mask" [B,N] or [B,N,2] (tss), "atpm" [B,N,1]}
    -- the keys the models' get_input / before_loss read (zarr_dataset.py:1161-1169).  `rank` / `world` shard the
    sample list (one process per GPU); `shuffle` reshuffles every epoch with `seed + epoch`.  Incomplete last
    batches are dropped."""

    def __init__(self, dataset, batch_size, device, shuffle=True, seed=0, rank=0, world=1):
        self.ds, self.B, self.device = dataset, batch_size, torch.device(device)
        self.shuffle, self.seed, self.rank, self.world = shuffle, seed, rank, world
        self.N = dataset.num_region_per_sample
        self.epoch = 0
        # ONE table for all cell types (rows of cell type c start at row0[c]): [motif columns | aTPM], expression pair and
        # TSS flag per peak.  A batch -- whatever mix of cell types its windows come from -- is then one row gather per
        # output (the first version gathered per cell type and scattered by batch position: 0.7-1.6 ms per step in the
        # training loop against 0.1 ms now; tools/ab_producer.sh).
        full, exp, tss, row0, r0 = [], [], [], {}, 0
        tss_like = next((rm.tss for rm in dataset.region_motifs.values() if rm.has_expression), None)
        for ct, rm in dataset.region_motifs.items():
            raw = torch.from_numpy(np.ascontiguousarray(rm.data)).to(self.device)
            if rm.normalize:                                   # normalised on the device, as normalized_data does
                table = (raw / (raw.max(dim=0).values * rm.motif_scaler)).clamp_(max=1.0)
            else:
                table = raw
            atpm = torch.from_numpy(np.ascontiguousarray(rm.atpm)).float().to(self.device)
            full.append(torch.cat([table.float(), atpm[:, None]], dim=1))
            n = table.shape[0]
            if tss_like is not None:
                if rm.has_expression:
                    exp.append(torch.from_numpy(np.column_stack((rm.expression_positive, rm.expression_negative))
                                                .astype(np.float32)).to(self.device))
                    tss.append(torch.from_numpy(np.ascontiguousarray(rm.tss)).to(self.device))
                else:
                    exp.append(torch.zeros(n, 2, device=self.device))
                    tss.append(torch.zeros((n,) + tuple(tss_like.shape[1:]), dtype=torch.from_numpy(tss_like[:1]).dtype,
                                           device=self.device))
            row0[ct] = r0
            r0 += n
        self.full = torch.cat(full, dim=0).contiguous()
        self.exp = torch.cat(exp, dim=0) if exp else None
        self.tss = torch.cat(tss, dim=0) if tss else None
        self.F = self.full.shape[1] - 1
        self.win = torch.arange(self.N, device=self.device)
        # first table row of every sample of the dataset
        self._row_of = np.array([row0[ct] + s for ct, s, _ in dataset.sample_indices], dtype=np.int64)

    def __len__(self):
        return (len(self.ds) // self.world) // self.B

    def _order(self):
        idx = np.arange(len(self.ds))
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(idx)
        return idx[self.rank::self.world]

    def gather(self, sample_ids):
        """One batch from explicit sample ids (windows of one batch may come from different cell types)."""
        ids = np.asarray(sample_ids, dtype=np.int64)
        B, N, F = len(ids), self.N, self.F
        st = torch.from_numpy(self._row_of[ids]).to(self.device, non_blocking=True)
        rows = st[:, None] + self.win[None, :]                                            # [B, N] table rows
        rm = self.full[rows]                                                              # [B, N, F + 1]
        atpm = rm[:, :, F].clone()
        if self.ds.quantitative_atac:
            rm[:, :, F] = atpm / atpm.max(dim=1, keepdim=True).values
        else:
            rm[:, :, F] = 1.0
        out = {"region_motif": rm, "atpm": atpm[:, :, None]}
        out["exp_label"] = self.exp[rows] if self.exp is not None else torch.zeros(B, N, 2, device=self.device)
        if self.ds.mask_ratio > 0:                              # pretrain: Bernoulli(mask_ratio) per region
            out["mask"] = (torch.rand(B, N, device=self.device) < self.ds.mask_ratio).to(torch.int64)
        elif self.tss is not None:
            out["mask"] = self.tss[rows]
        else:
            out["mask"] = torch.zeros(B, N, dtype=torch.int64, device=self.device)
        return out

    def __iter__(self):
        order = self._order()
        self.epoch += 1
        for b in range(len(order) // self.B):
            yield self.gather(order[b * self.B:(b + 1) * self.B])
