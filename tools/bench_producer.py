"""GPU: throughput of data.RegionBatchProducer at the bench shape, next to the model's consumption rate:
batches of 256 windows x 200 regions gathered on the device from HBM-resident tables (synthetic store, 4 cell types x
200 k peaks).  Prints regions/s of the producer alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from get_model_amd import data as D

st = D.make_synthetic_store(num_peaks=200000, celltypes=("a", "b", "c", "d"), seed=0, chromosomes=tuple(f"chr{i}" for i in range(1, 9)))
for quant, ratio in ((False, 0.0), (True, 0.0), (False, 0.5)):
    ds = D.RegionMotifDataset(st, sampling_step=100, num_region_per_sample=200, seed=1, quantitative_atac=quant, mask_ratio=ratio)
    prod = D.RegionBatchProducer(ds, batch_size=256, device="cuda", shuffle=True, seed=2)
    it = iter(prod)
    for _ in range(3):
        next(it)
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    for batch in it:
        n += 1
        if n == 20:
            break
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"quantitative_atac={quant} mask_ratio={ratio}: {len(ds)} windows, {dt * 1e3:.2f} ms per batch of 256 x 200 = "
          f"{256 * 200 / dt / 1e6:.1f} M regions/s (the train step consumes 1.4 M regions/s, inference 4.4 M)")
