"""Benchmark of the GET region-transformer hot path on MI355X (contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One "step" = one training step of GETRegionFinetune 12L x 768 on a synthetic batch resident in HBM:
get_input -> forward -> before_loss -> PoissonNLL -> backward -> bucketed gradient all-reduce (N>1) -> AdamW.
Per-GPU batch is fixed (weak scaling): 256 samples x 200 regions (BASELINE.json configs[1]).
Prints ONE JSON line on rank 0.
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
FLOP_PER_REGION_N200 = 5.358e8   # fwd+bwd algorithmic FLOPs per region at N=200 (SURVEY.md 8d / BASELINE.md 4)


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
    return {"region_motif": rm.to(device), "exp_label": exp.to(device)}


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


def pmc_traffic(group):
    """HBM-side bytes per launch of a kernel group (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md corrections),
    from the committed rocprofv3 --pmc passes of this same workload (tools/pmc_traffic.sh -> profiles/); PMC
    counters cannot be read from inside the process, so this is None when the file or the group is missing."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if group not in _TRAFFIC_GROUPS or not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    prim, helpers = _TRAFFIC_GROUPS[group]

    def rows(prefixes):
        return [v for k, v in d.items() if any(k == p or k.startswith(p + ",") or k.startswith(p + ">") or
                                               k.startswith(p + "<") for p in prefixes)]

    calls = sum(v["calls"] for v in rows(prim))
    if calls == 0:
        return None
    tot = sum(v["calls"] * (v["fetch_bytes"] + v["write_bytes"]) for v in rows(prim + helpers))
    return tot / calls


def cpu_baseline(budget_s=20.0):
    """The oracle (CPU restatement of the reference path, fp32) timed on this box's host cores:
    BASELINE.json configs[0] shape (B=4, N=200, 12 layers), train-mode-equivalent fwd + loss + bwd."""
    from oracle import get_cpu as O
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sd = O.fill_state_dict("finetune", seed=3, num_layers=12)
    batch = O.make_batch(4, 200, 3)
    O.loss_and_grads("finetune", sd, batch, num_layers=12)        # warm-up
    times = []
    t_end = time.time() + budget_s
    while len(times) < 5 and (time.time() < t_end or not times):
        t0 = time.time()
        O.loss_and_grads("finetune", sd, batch, num_layers=12)
        times.append(time.time() - t0)
    best = min(times)
    return {"value": 4 * 200 / best, "unit": "regions/s", "cores": cores, "kind": "port",
            "sample": f"oracle/get_cpu.py fp32 fwd+loss+bwd, GETRegionFinetune 12Lx768, B=4 x N=200, best of {len(times)} "
                      f"steps ({best:.3f} s/step) after 1 warm-up"}


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
    ap.add_argument("--torch-adamw", action="store_true", help="torch.optim.AdamW(fused=True) instead of FusedAdamW")
    ap.add_argument("--cu-limit", type=int, default=0, help="experiment: getmi_set_cu_limit before the run")
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
    model = build_model(region_model_yaml("finetune", num_layers=args.layers)).to(device)
    broadcast_parameters(model)
    model.train(not args.inference)
    buckets = GradBuckets(model)
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

    def infer_step():
        with torch.inference_mode():
            return model(**model.get_input(batch)).float().mean()

    def step():
        if args.inference:
            return infer_step()
        buckets.zero_grad()
        out = model(**model.get_input(batch))
        pred, obs = model.before_loss(out, batch)
        loss = model.after_loss(model.loss(pred, obs))
        loss.backward()
        buckets.finish()
        opt.step()
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
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    prof, K.PROFILE = K.PROFILE, None
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
            roof = {"bound": "mfma", "kernel": dom, "achieved": fl / sec / 1e12, "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": fl / sec / 1e12 / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(dom),
                    "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_traffic.json)",
                    "launches": n, "avg_launch_us": sec / n * 1e6}
        result = {
            "metric": ("inference" if args.inference else "train") + f" regions/sec/node, GET 12Lx768 @{N}-region ctx",
            "value": value, "unit": "regions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic region x motif tensors (SURVEY.md 8d), random-init weights",
            "config": {"workload": f"GETRegionFinetune {args.layers}Lx768 train step (fwd+PoissonNLL+bwd+grad all-reduce+AdamW), "
                                   f"per-GPU batch {B} x {N} regions x 283 features", "global_batch": world * B,
                       "regions_per_sample": N, "parallelism": f"dp{world}", "grad_exchange": exchange},
            "loss": float(loss.item()),
            "e2e_mfma_frac": value * FLOP_PER_REGION_N200 / (world * PEAK_BF16_TFLOPS * 1e12) if N == 200 and args.layers == 12 else None,
            "roofline": roof, "kernels": kernels,
        }
    if world > 1:
        dist.barrier()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
