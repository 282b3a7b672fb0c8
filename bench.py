"""Benchmark of the GET region-transformer hot path on MI355X (contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One "step" = one training step of GETRegionFinetune 12L x 768 on a synthetic batch resident in HBM:
get_input -> forward -> before_loss -> PoissonNLL -> backward -> bucketed gradient all-reduce (N>1) -> AdamW.
Per-GPU batch is fixed (weak scaling): 256 samples x 200 regions (BASELINE.json configs[1]).
Prints ONE JSON line on rank 0.

The other BASELINE.json configurations (not the contract metric; each prints the same kind of line):
    C3  --model pretrain                                   masked-region objective (mask ratio 0.5, MSE)
    C4  --model atac --regions 2000 --batch 16             Motif->ATAC model, long context
    C5  --model finetune --inference --batch 1024          forward only under torch.inference_mode
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0         # HBM3E peak (same guide; 6.3 TB/s is what a streaming copy reaches)
FLOP_PER_REGION_N200 = 5.358e8   # fwd+bwd algorithmic FLOPs per region at N=200 (SURVEY.md 8d / BASELINE.md 4)


def flop_per_region(kind, N, layers, forward_only, D=768, n_feat=283, masked_frac=0.5):
    """Algorithmic FLOPs per region (2mnk per GEMM; per layer 24 T D^2 + 4 T^2 D with T = N + 1 tokens;
    embed 2 N 283 D; head 2 N D out, or for pre-training 2 n_masked D 283); backward = 2 x forward.
    N=200, 12 layers, finetune: 3.572e10 per sample forward -> 5.358e8 per region fwd+bwd; N=2000: 7.33e8."""
    T = N + 1
    fwd = layers * (24.0 * T * D * D + 4.0 * T * T * D) + 2.0 * N * n_feat * D
    fwd += 2.0 * masked_frac * N * D * n_feat if kind == "pretrain" else 2.0 * N * D * (2 if kind == "finetune" else 1)
    return fwd * (1.0 if forward_only else 3.0) / N


def cpu_model_string():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def synthetic_batch(B, N, seed, device):
    """motif columns U[0,1) with ~60 % zeros, ATAC column in (0,1], labels 90 % zeros else U[0,4)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    u = torch.rand(B, N, 283, generator=g)
    z = torch.rand(B, N, 283, generator=g)
    rm = torch.where(z < 0.6, torch.zeros_like(u), u)
    rm[:, :, -1] = torch.rand(B, N, generator=g) * 0.99 + 0.01
    lab = torch.rand(B, N, 2, generator=g) * 4
    lz = torch.rand(B, N, 2, generator=g)
    exp = torch.where(lz < 0.9, torch.zeros_like(lab), lab)
    mask = (torch.rand(B, N, generator=g) < 0.5).to(torch.int64)     # pretrain: Bernoulli(0.5), zarr_dataset.py:1149-1152
    return {"region_motif": rm.to(device), "exp_label": exp.to(device), "mask": mask.to(device)}


# bench kernel group -> (primary device kernels, helper kernels launched with them): name PREFIXES as rocprofv3 prints
# them (template arguments after the first vary with the tile height)
def _tn(*epis):
    return [f"{k}<{e}" for e in epis for k in ("gemm_tn256p_kernel", "gemm_tn256_kernel", "gemm_tn_kernel")]


_TRAFFIC_GROUPS = {
    "gemm_wgrad_kernel(+reduce_slabs)": (["gemm_wgrad256_kernel", "gemm_wgrad_kernel"], ["reduce_slabs_kernel"]),
    "gemm_tn_kernel<bf16>": (_tn(0), []),
    "gemm_tn_kernel<bias_gelu>": (_tn(1, 6), []),
    "gemm_tn_kernel<bias_resid>": (_tn(2), []),
    "gemm_tn_kernel<dgelu>": (_tn(3, 7), []),
    "attn_bwd_kernel(+delta)": (["attn_bwd_kernel"], []),
    "attn_fwd_kernel": (["attn_fwd_kernel"], []),
}


PMC_FILES = ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")


def pmc_traffic(group):
    """(bytes per launch, provenance) of a kernel group: HBM-side bytes (FETCH_SIZE x2 + WRITE_SIZE, the corrections of
    MI355X_MICROARCH.md) from the committed rocprofv3 --pmc passes of this same workload (tools/pmc_traffic.sh ->
    profiles/).  PMC counters cannot be read from inside the process, so this number is NOT measured in this run: the
    provenance names the file and the commit / command it was taken at ("_meta" entry of the file).  (None, None) when the
    file or the group is missing."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f) for f in PMC_FILES) if os.path.exists(q)), None)
    if group not in _TRAFFIC_GROUPS or path is None:
        return None, None
    with open(path) as f:
        d = json.load(f)
    meta = d.pop("_meta", {}) if isinstance(d.get("_meta"), dict) else {}
    prim, helpers = _TRAFFIC_GROUPS[group]

    def rows(prefixes):
        return [v for k, v in d.items() if any(k == p or k.startswith(p + ",") or k.startswith(p + ">") or
                                               k.startswith(p + "<") or k.startswith(p + "(") for p in prefixes)]

    calls = sum(v["calls"] for v in rows(prim))
    if calls == 0:
        return None, None
    tot = sum(v["calls"] * (v["fetch_bytes"] + v["write_bytes"]) for v in rows(prim + helpers))
    src = {"file": "profiles/" + os.path.basename(path), "commit": meta.get("commit"), "command": meta.get("command"),
           "measured_in_this_run": False}
    return tot / calls, src


def cpu_baseline(kind="finetune", N=200, forward_only=False, budget_s=25.0):
    """The oracle (CPU restatement of the reference path, fp32) timed on this box's host cores on a bounded sample of the
    same workload: B=4 samples of the same model / context length (C1 shape B=4 x N=200, C4 shape
    B=4 x N=2000), 12 layers, fwd + loss + bwd (forward only for the inference workload)."""
    from oracle import get_cpu as O
    # the GPU box gives one GPU's share of the host: 16 threads (OMP_NUM_THREADS there); os.cpu_count() is the machine
    cores = min(int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1)
    torch.set_num_threads(cores)
    B = 4
    sd = O.fill_state_dict(kind, seed=3, num_layers=12)
    batch = O.make_batch(B, N, 3, kind=kind)

    def once():
        if forward_only:
            with torch.no_grad():
                if kind == "pretrain":
                    O.pretrain_loss(sd, batch, num_layers=12)
                else:
                    O.finetune_loss(sd, batch, kind=kind, num_layers=12)
        else:
            O.loss_and_grads(kind, sd, batch, num_layers=12)

    t0 = time.time()
    once()        # warm-up
    warm = time.time() - t0
    times = []
    t_end = time.time() + max(budget_s - warm, 0.0)
    while len(times) < 5 and (time.time() < t_end or not times):
        t0 = time.time()
        once()
        times.append(time.time() - t0)
    best = min(times)
    what = "fwd" if forward_only else "fwd+loss+bwd"
    return {"value": B * N / best, "unit": "regions/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model_string(), "host_cpus": os.cpu_count(),
            "sample": f"oracle/get_cpu.py fp32 {what}, {MODEL_NAMES[kind]} 12Lx768, B={B} x N={N}, best of {len(times)} "
                      f"steps ({best:.3f} s/step) after 1 warm-up, {cores} threads"}


DOMINANT_GROUP = "gemm_wgrad_kernel(+reduce_slabs)"
MODEL_NAMES = {"finetune": "GETRegionFinetune", "pretrain": "GETRegionPretrain", "atac": "GETRegionFinetuneATAC"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (samples)")
    ap.add_argument("--regions", type=int, default=200)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--all-kernel-events", action="store_true",
                    help="bracket EVERY GEMM / attention launch with HIP events (the per-kernel table; costs ~0.9 ms per C2 "
                         "step).  Default: only the dominant kernel group (weight-gradient GEMMs) is bracketed")
    ap.add_argument("--torch-adamw", action="store_true", help="torch.optim.AdamW(fused=True) instead of FusedAdamW")
    ap.add_argument("--cu-limit", type=int, default=0, help="experiment: getmi_set_cu_limit before the run")
    ap.add_argument("--model", choices=["finetune", "pretrain", "atac"], default="finetune",
                    help="finetune = the contract workload (BASELINE configs[1]); pretrain = C3; atac = C4")
    ap.add_argument("--producer", action="store_true",
                    help="draw every batch of the timed loop through data.RegionBatchProducer (device-side gather from "
                         "HBM-resident peak tables, SURVEY 8 f3) instead of re-using one resident batch")
    ap.add_argument("--no-extra-pass", action="store_true",
                    help="skip the un-timed extra steps after the timed region that bracket the LayerNorm-backward kernel "
                         "(roofline_hbm) and every GEMM / attention launch (kernels table) with HIP events")
    ap.add_argument("--inference", action="store_true",
                    help="forward only, eval mode, torch.inference_mode (SURVEY 8d C5); not the contract metric")
    args = ap.parse_args()

    if os.environ.get("GETMI_BENCH_TRACE"):          # debugging aid: dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["GETMI_BENCH_TRACE"]), exit=True)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    # one rank per GPU; GETMI_BENCH_BACKEND=gloo lets several ranks share the GPUs of a smaller box (rehearsal of the
    # multi-rank code path only -- its numbers mean nothing)
    backend = os.environ.get("GETMI_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from get_model_amd import kernels as K
    from get_model_amd.config import build_model, region_model_yaml
    from get_model_amd.ddp import GradBuckets, broadcast_parameters

    torch.manual_seed(0)
    model = build_model(region_model_yaml(args.model, num_layers=args.layers)).to(device)
    broadcast_parameters(model)
    model.train(not args.inference)
    buckets = None if args.inference else GradBuckets(model)
    # AdamW as the reference configures it (config/finetune_tutorial.yaml:29-35, optim.py:44-130, run.py:477-497):
    # layer-wise parameter groups, weight decay 0.05 on >=2-D non-bias parameters; get_model_amd.optim.FusedAdamW
    # (one HIP launch, also refreshes the bf16 operand copies); --torch-adamw times torch.optim.AdamW(fused) instead
    from types import SimpleNamespace
    from get_model_amd import optim as GO
    ocfg = SimpleNamespace(opt="adamw", lr=1e-4, weight_decay=0.05, opt_eps=1e-8, opt_betas=[0.9, 0.999])
    if args.torch_adamw:
        opt = torch.optim.AdamW(GO.get_parameter_groups(model, 0.05, model.no_weight_decay()), lr=1e-4,
                                betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, fused=True)
    else:
        opt = GO.configure_optimizer(model, ocfg)
    B, N = args.batch, args.regions
    batch = synthetic_batch(B, N, 1234 + rank, device)
    batches = None
    if args.producer:
        # f3: a synthetic zarr-shaped store (4 cell types x 60 k peaks), tables resident in HBM; the host sends B window
        # starts per step and the batch is gathered on the device
        from get_model_amd import data as GD
        store = GD.make_synthetic_store(num_peaks=60000, celltypes=("a", "b", "c", "d"), seed=rank,
                                        chromosomes=tuple(f"chr{i}" for i in range(1, 9)))
        ds = GD.RegionMotifDataset(store, sampling_step=100, num_region_per_sample=N, seed=1, quantitative_atac=False,
                                   mask_ratio=0.5 if args.model == "pretrain" else 0.0)
        producer = GD.RegionBatchProducer(ds, batch_size=B, device=device, shuffle=True, seed=2 + rank)

        def endless():
            while True:
                for b in producer:
                    yield b

        batches = endless()

    batch0 = batch

    def infer_step():
        batch = next(batches) if batches is not None else batch0
        with torch.inference_mode():
            out = model(**model.get_input(batch))
            return (out[0] if isinstance(out, tuple) else out).float().mean()

    opt_events = []      # (the optimizer step is in ms_per_step and also reported on its own)

    def step():
        if args.inference:
            return infer_step()
        batch = next(batches) if batches is not None else batch0
        buckets.zero_grad()
        out = model(**model.get_input(batch))
        pred, obs = model.before_loss(out, batch)
        loss = model.after_loss(model.loss(pred, obs))
        loss.backward()
        buckets.finish()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        opt.step()          # on torch's current stream, the one the events are recorded on
        e1.record()
        opt_events.append((e0, e1))
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.cu_limit:
        from get_model_amd.ddp import set_cu_limit
        set_cu_limit(args.cu_limit)
    # (N > 1) pick the gradient-exchange mode by measurement; these steps come before the warm-up and are not timed
    from get_model_amd.ddp import autotune
    exchange = autotune(step, buckets) if (world > 1 and not args.inference) else None
    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_kernel_events:
        K.PROFILE = []
        # The dominant kernel group of a training step is known from the committed rocprofv3 kernel statistics of this
        # same command (profiles/r03_bench_kernel_stats.csv: weight-gradient GEMMs 20.4 % of the step, next the plain
        # bf16 GEMMs 19.0 %), so only its launches carry events by default; forward-only runs have no wgrad: all groups.
        K.PROFILE_ONLY = None if (args.all_kernel_events or args.inference) else {DOMINANT_GROUP}
    opt_events.clear()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    prof, K.PROFILE = K.PROFILE, None
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    # ---- un-timed extra passes (after the timed region, so their events cannot perturb it) ----
    extra_prof = ln_prof = None
    if not args.no_kernel_events and not args.no_extra_pass and rank == 0 and world == 1:
        K.PROFILE, K.PROFILE_ONLY = [], {"ln_bwd_kernel"}
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ln_prof, K.PROFILE, K.PROFILE_ONLY = K.PROFILE, [], None
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        extra_prof, K.PROFILE = K.PROFILE, None
    if world > 1:
        t = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms = dt / args.steps * 1e3
    value = world * B * N / (dt / args.steps)

    result = None
    if rank == 0:
        # ---- per-kernel roofline from HIP events recorded on the launch stream during the timed region ----
        roof = None
        kernels = {}
        if prof:
            agg = {}
            for name, flops, e0, e1 in prof:
                a = agg.setdefault(name, [0.0, 0.0, 0])
                a[0] += flops
                a[1] += e0.elapsed_time(e1) * 1e-3
                a[2] += 1
            for name, (fl, sec, n) in agg.items():
                kernels[name] = {"launches": n, "avg_us": sec / n * 1e6, "tflops": fl / sec / 1e12,
                                 "time_share": sec / dt}
            dom = max(agg, key=lambda k: agg[k][1])
            fl, sec, n = agg[dom]
            traffic, traffic_src = pmc_traffic(dom)
            roof = {"bound": "mfma", "kernel": dom, "achieved": fl / sec / 1e12, "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": fl / sec / 1e12 / PEAK_BF16_TFLOPS, "traffic": traffic,
                    "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                    "launches": n, "avg_launch_us": sec / n * 1e6,
                    "events_on": "every GEMM / attention launch" if K.PROFILE_ONLY is None else
                                 "this kernel group only (dominant by profiles/r03_bench_kernel_stats.csv; "
                                 "--all-kernel-events for the table)"}
        # HBM-bound sibling (north_star: HBM GB/s of the norm kernels): LayerNorm backward, algorithmic bytes of each
        # launch (kernels.layernorm_bwd) over HIP-event durations, from two extra un-timed steps
        roof_hbm = None
        if ln_prof:
            by = sum(x[1] for x in ln_prof)
            sec = sum(x[2].elapsed_time(x[3]) for x in ln_prof) * 1e-3
            roof_hbm = {"bound": "hbm", "kernel": "ln_bwd_kernel", "achieved": by / sec / 1e9, "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": by / sec / 1e9 / PEAK_HBM_GBS, "launches": len(ln_prof),
                        "avg_launch_us": sec / len(ln_prof) * 1e6, "bytes_per_launch": by / len(ln_prof),
                        "measured": "HIP events on the launch stream, 2 un-timed steps after the timed region"}
        if extra_prof:
            agg = {}
            for name, flops, e0, e1 in extra_prof:
                a = agg.setdefault(name, [0.0, 0.0, 0])
                a[0] += flops
                a[1] += e0.elapsed_time(e1) * 1e-3
                a[2] += 1
            for name, (fl, sec, n) in agg.items():
                if name in kernels:
                    continue
                rec = {"launches_per_step": n / 2, "avg_us": sec / n * 1e6, "from": "2 un-timed steps with events on every launch"}
                if name == "ln_bwd_kernel" or name == "ln_fwd_kernel":
                    rec["gb_per_s"] = fl / sec / 1e9
                else:
                    rec["tflops"] = fl / sec / 1e12
                kernels[name] = rec
        fpr = flop_per_region(args.model, N, args.layers, args.inference)
        loss_name = "MSE" if args.model == "pretrain" else "PoissonNLL"
        if args.inference:
            work = (f"{MODEL_NAMES[args.model]} {args.layers}Lx768 forward only (eval mode, torch.inference_mode), "
                    f"per-GPU batch {B} x {N} regions x 283 features")
        else:
            work = (f"{MODEL_NAMES[args.model]} {args.layers}Lx768 train step (fwd+{loss_name}+bwd+grad all-reduce+AdamW), "
                    f"per-GPU batch {B} x {N} regions x 283 features")
        result = {
            "metric": ("inference" if args.inference else "train") + f" regions/sec/node, GET 12Lx768 @{N}-region ctx",
            "value": value, "unit": "regions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic region x motif tensors (SURVEY.md 8d), random-init weights",
            "config": {"workload": work, "global_batch": world * B, "regions_per_sample": N, "parallelism": f"dp{world}",
                       "grad_exchange": exchange},
            "loss": float(loss.item()),
            "flop_per_region": fpr,
            "e2e_mfma_frac": value * fpr / (world * PEAK_BF16_TFLOPS * 1e12),
            "optimizer_ms_per_step": (sum(a.elapsed_time(b) for a, b in opt_events) / len(opt_events)) if opt_events else None,
            "ms_per_step_median": median_ms,
            "batch_source": "RegionBatchProducer (device-side gather per step)" if args.producer else "one batch resident in HBM",
            "roofline": roof, "roofline_hbm": roof_hbm, "kernels": kernels,
        }
    if world > 1:
        dist.barrier()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.model, 2000 if N >= 1000 else 200, args.inference)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
