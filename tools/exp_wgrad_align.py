"""GPU experiment (round 3): does the weight-gradient GEMM run faster when every XCD holds exactly one M-split?

gemm_wgrad256_kernel cuts M into splits of whole 256 x 256 output-tile sets and gives the tiles of one split consecutive
logical workgroup ids (= one XCD's L2 under xcd_remap).  The training shapes have 36 tiles per split -> 7 splits = 252
workgroups, 31.5 per XCD: every split straddles two XCDs, so its X / Y rows are fetched into two L2s (rocprofv3: 607 MB
fetched per launch for ~400 MB of operands, L2 hit 0.67).  A shape with 32 tiles per split gets 8 splits = 256 workgroups,
one split per XCD.  Same M, same FLOPs per tile: TFLOP/s of the two layouts, interleaved, operands far larger than the
Infinity Cache.  Prints one line per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from get_model_amd import kernels as K

M = 51456
g = torch.Generator(device="cpu").manual_seed(0)
def rb(*s): return (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).cuda()
shapes = [(3072, 768), (2048, 1024), (4096, 512), (2304, 1024), (1024, 2048), (768, 3072)]
ops = {}
for n1, n2 in shapes:
    for n in (n1, n2):
        if n not in ops:
            ops[n] = rb(M, n)
res = {s: [] for s in shapes}
for rnd in range(4):
    for (n1, n2) in shapes:
        x, y = ops[n1], ops[n2]
        for _ in range(2):
            K.gemm_wgrad(x, y)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.gemm_wgrad(x, y)
        e1.record(); torch.cuda.synchronize()
        res[(n1, n2)].append(e0.elapsed_time(e1) / 10 * 1e3)
from get_model_amd import lib
for (n1, n2), us in res.items():
    tiles = (n1 // 256) * (n2 // 256)
    ns = lib.query("getmi_gemm_wgrad_workspace_bytes", M, n1, n2) // (n1 * n2 * 4)
    med = sorted(us)[len(us) // 2]
    print(f"wgrad {n1:5d} x {n2:5d}: {tiles:3d} tiles x {ns} splits = {tiles * ns:3d} workgroups  {med:8.1f} us (incl. slab reduce)  "
          f"{2.0 * M * n1 * n2 / med / 1e6:8.1f} TFLOP/s   all rounds: {[round(u, 1) for u in us]}")
