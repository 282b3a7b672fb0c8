"""Data-parallel gradient exchange for the GET region models: one process per GPU, bucketed all-reduce on
RCCL (torch.distributed backend "nccl") overlapped with backward.

Counterpart of the reference's Lightning `strategy="ddp_find_unused_parameters_true"` (utils.py:41-51): samples
are split over ranks, weights are replicated, and the only exchange step of the path is the mean of the
gradients.  Differences by design:
  * gradients live in flat fp32 buckets laid out in REVERSE forward order (head -> block L-1 -> ... -> embed),
    one bucket per encoder block (~28 MB at 12Lx768; the 30 MB cap merges only the small head / embed groups), so a bucket completes as soon as its block's fused
    backward returns and its all-reduce runs on RCCL's stream under the remaining backward;
  * `param.grad` are views into the buckets: no flatten/unflatten copies, and the optimizer reads the reduced
    gradients in place; the weight-gradient GEMMs write their result directly into the bucket slot
    (`param.getmi_grad_buffer`), which autograd then adopts as `param.grad` without a copy or an add;
  * no unused-parameter graph walk: parameters that received no gradient are simply reduced as zeros.
  * how the exchange shares the GPU with backward is MEASURED per node (`autotune`): overlapped on all CUs,
    overlapped with some CUs left to RCCL (the persistent GEMM / wgrad kernels take whole CUs), or not overlapped.
Works with any backend (`gloo` on CPU for the tests).
"""
import re

import torch
import torch.distributed as dist


def _bucket_key(name):
    m = re.match(r"encoder\.blocks\.(\d+)\.", name)
    if m:
        return ("block", int(m.group(1)))
    if name.startswith("encoder.norm") or name.startswith("head"):
        return ("head", 0)
    return ("embed", 0)


class GradBuckets:
    """Flat gradient buckets + async all-reduce from post-accumulate-grad hooks."""

    def __init__(self, module, process_group=None, bucket_cap_mb=30.0):
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        groups = {}
        for n, p in named:
            groups.setdefault(_bucket_key(n), []).append(p)
        order = sorted(groups, key=lambda k: {"head": 0, "block": 1, "embed": 2}[k[0]] * 100000 - (k[1] if k[0] == "block" else 0))
        # merge small neighbours up to the cap so tiny buckets (head, embed) do not cost a launch each
        cap = int(bucket_cap_mb * 1024 * 1024 / 4)
        self.buckets = []
        cur, cur_n = [], 0
        for k in order:
            n = sum(p.numel() for p in groups[k])
            if cur and cur_n + n > cap:
                self.buckets.append(cur)
                cur, cur_n = [], 0
            cur += groups[k]
            cur_n += n
        if cur:
            self.buckets.append(cur)
        self.flat = []
        self._bucket_of = {}
        self._view = {}
        for bi, params in enumerate(self.buckets):
            pad = lambda n: (n + 63) // 64 * 64     # every slot starts on a 256-byte boundary (kernel stores need 16 B)
            total = sum(pad(p.numel()) for p in params)
            dev, dt = params[0].device, params[0].dtype
            flat = torch.zeros(total, dtype=dt, device=dev)
            off = 0
            for p in params:
                view = flat[off:off + p.numel()].view_as(p)
                off += pad(p.numel())
                self._bucket_of[p] = bi
                self._view[p] = view
                # the fused backward (functional.grad_buffer) writes weight gradients straight into this slot and
                # autograd adopts the returned alias as p.grad: no zero fill, no `grad += new` pass per step.
                p.getmi_grad_buffer = view
                p.grad = None
            self.flat.append(flat)
        self.sync = True      # False: accumulate locally (micro-batches); the exchange happens in finish()
        self.overlap = True   # False: no all-reduce is launched during backward, all of them start in finish()
        # CUs the one-workgroup-per-CU kernels may assume WHILE an all-reduce overlaps backward (0: all).  Applied from
        # the first bucket launch of a step until finish(): forward, and backward before the first bucket, keep all CUs.
        self.cu_limit_during_exchange = 0
        self._cu_limited = False
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for b in self.buckets for p in b]
        self._use_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.zero_grad()

    # ---- per step ----
    def zero_grad(self):
        """Start a new step: gradients are dropped (set to None), not zero-filled; slots of parameters that end
        the step without a gradient are zeroed in finish().  Also closes a step that never reached finish() (an
        exception in backward): pending exchanges are waited for and the CU limit is lifted."""
        self._drain()
        for b in self.buckets:
            for p in b:
                p.grad = None
        self._begin_step()

    def _begin_step(self):
        for b in self.buckets:
            for p in b:
                p._getmi_slot_taken = False         # see functional.grad_buffer
        self._remaining = [len(b) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._stray = [[] for _ in self.buckets]
        self._cb_queued = False
        self._handles = []

    def _drain(self):
        """Wait for every exchange in flight and give the compute kernels all CUs back."""
        for h, flat in getattr(self, "_handles", []):
            h.wait()
            if flat is not None:
                flat.div_(self.world)
        self._handles = []
        if self._cu_limited:        # the exchange is ordered before everything launched from here on
            set_cu_limit(0)
            self._cu_limited = False

    def _launch(self, bi):
        self._launched[bi] = True
        if self.world == 1:
            return
        flat = self.flat[bi]
        if self.cu_limit_during_exchange > 0 and not self._cu_limited and self.overlap:
            set_cu_limit(self.cu_limit_during_exchange)     # host-side grid sizes of the launches that follow
            self._cu_limited = True
        if self._use_avg:
            self._handles.append((dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.pg, async_op=True), None))
        elif flat.is_cuda:
            # a host backend (gloo) with GPU gradients: staged through host memory, synchronously (rehearsals of the
            # multi-rank path on a box without RCCL peers; gloo's own CUDA path hangs on this stack)
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.pg)
            flat.copy_(host.div_(self.world))
        else:
            self._handles.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), flat))

    def _on_grad(self, p):
        view = self._view[p]
        g = p.grad
        bi = self._bucket_of[p]
        p._getmi_slot_taken = False      # this backward is done with the parameter: the slot may be handed out again
        if not self._cb_queued:          # once per backward pass: flush / re-arm when the pass ends (_end_of_backward)
            self._cb_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
        # (world 1: nothing is exchanged, so a second backward before finish() is plain gradient accumulation)
        if self._remaining[bi] <= 0 and self.sync and self._launched[bi] and self.world > 1:
            raise RuntimeError("ddp.GradBuckets: a gradient arrived for a bucket that was already exchanged in this step "
                               "(a second backward without finish()? use sync=False for micro-batches)")
        if g.data_ptr() != view.data_ptr() or g.stride() != view.stride():
            # small parameters (biases, norms, tokens) and anything not written in place: gathered into the
            # bucket by ONE multi-tensor copy when the bucket completes, instead of a copy launch per parameter
            self._stray[bi].append((view, g))
            p.grad = view
        self._remaining[bi] -= 1
        if self._remaining[bi] == 0:
            self._flush(bi)
            if self.sync and self.overlap:
                self._launch(bi)

    def _end_of_backward(self):
        """End of one backward pass: buckets that stayed incomplete (unused parameters) are flushed, so a following
        accumulate-backward sees real data; after a sync=False micro-batch the per-bucket countdown is re-armed, so the
        LAST micro-batch (sync=True) launches each bucket's exchange as it completes instead of deferring all of them
        to finish()."""
        self._cb_queued = False
        for bi in range(len(self.buckets)):
            self._flush(bi)
        if not any(self._launched):
            self._remaining = [len(b) for b in self.buckets]

    def _flush(self, bi):
        stray = self._stray[bi]
        if stray:
            torch._foreach_copy_([v for v, _ in stray], [g for _, g in stray])
            self._stray[bi] = []

    def finish(self):
        """Call after backward: zero-fills the slots of parameters that got no gradient (unused), launches the
        buckets that are still pending, then waits for every exchange."""
        for b in self.buckets:
            for p in b:
                if p.grad is None:
                    view = self._view[p]
                    view.zero_()
                    p.grad = view
        for bi in range(len(self.buckets)):
            self._flush(bi)
            if not self._launched[bi]:
                self._launch(bi)
        self._drain()
        # the step's bookkeeping ends here, whichever zero_grad the training loop calls next (optimizer.zero_grad()
        # leaves this object alone; with stale counters no bucket would ever be exchanged again and ranks would diverge)
        self._begin_step()

    def remove(self):
        for h in self._hooks:
            h.remove()


def set_cu_limit(n):
    """Grid size the one-workgroup-per-CU kernels assume (include/getmi355.h: getmi_set_cu_limit); 0 = all CUs.
    No-op without a GPU (the gloo tests)."""
    if not torch.cuda.is_available():
        return 0
    from . import lib
    return lib.query("getmi_set_cu_limit", int(n))


def autotune(step_fn, buckets, steps=2, reserve_options=(16, 32, 64), total_cus=None):
    """Measure, on this node, how the gradient exchange should share the GPU with backward, and leave the winner set.

    Candidates: all-reduces overlapped with backward on all CUs; overlapped with `r` CUs left to RCCL's channel
    workgroups (the persistent GEMM / wgrad kernels take whole CUs and would otherwise run a second round for the
    displaced workgroups); and no overlap (every all-reduce starts after backward).  Each candidate runs one
    settling step and `steps` timed steps of `step_fn` (a full training step incl. buckets.finish()); ranks agree on
    the maximum time.  The CU limit applies only while an exchange is in flight (GradBuckets.cu_limit_during_exchange).
    Returns {"overlap": bool, "cu_limit": int, "ms": {candidate: ms_per_step}}.
    world_size 1: nothing to tune.  total_cus: CU count to derive the limited candidates from (default: asked of the
    device; the gloo tests pass it to rehearse those candidates without a GPU)."""
    import time
    result = {"overlap": True, "cu_limit": 0, "ms": {}}
    if buckets.world == 1:
        return result
    gpu = torch.cuda.is_available()
    total = 0
    if gpu:
        from . import lib
        set_cu_limit(0)
        total = lib.query("getmi_cu_count")
    if total_cus is not None:
        total = int(total_cus)
    cands = [(True, 0)] + [(True, total - r) for r in reserve_options if total - r >= 64] + [(False, 0)]

    def fence():
        if gpu:
            torch.cuda.synchronize()
        dist.barrier(group=buckets.pg)

    best = None
    for overlap, limit in cands:
        buckets.overlap = overlap
        buckets.cu_limit_during_exchange = limit
        step_fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        fence()
        t = torch.tensor([(time.perf_counter() - t0) / steps * 1e3], dtype=torch.float64,
                         device=buckets.flat[0].device if buckets._use_avg else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=buckets.pg)
        ms = float(t.item())
        result["ms"]["%s/%d" % ("overlap" if overlap else "serial", limit)] = ms
        if best is None or ms < best[0]:
            best = (ms, overlap, limit)
    buckets.overlap = best[1]
    buckets.cu_limit_during_exchange = best[2]
    result["overlap"], result["cu_limit"] = best[1], best[2]
    return result


def broadcast_parameters(module, src=0, process_group=None):
    """Make every rank start from rank `src`'s weights (what DDP does at construction)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    host_staged = dist.get_backend(process_group) != "nccl"
    with torch.no_grad():       # on the tensors themselves (not .data): bumps _version, so cached bf16 copies refresh
        for t in list(module.parameters()) + list(module.buffers()):
            if host_staged and t.is_cuda:
                h = t.cpu()
                dist.broadcast(h, src=src, group=process_group)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src, group=process_group)
