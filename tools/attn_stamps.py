"""Diagnostic (GPU): where the attention backward spends its cycles.  Builds a private copy of the library with
-DGETMI_ATTN_STAMPS (s_memtime stamps at the phase boundaries of every wave), runs the kernel at the bench shape and
prints the median cycles per phase for a few waves.  The product library is not touched."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.environ.get("GETMI_STAMPS_LIB", os.path.join(ROOT, "get_model_amd", "build", "libgetmi355_stamps.so"))


def build():
    csrc = os.path.join(ROOT, "get_model_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in ("error.hip", "gemm.hip", "attention.hip", "norm.hip", "misc.hip", "optim.hip", "capi.hip")]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value",
                           "-ffp-contract=fast", "-DGETMI_ATTN_STAMPS", "-shared", "-o", out] + srcs)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
        sys.exit(0)
    os.environ["GETMI_LIB"] = out
    import numpy as np
    import torch
    from get_model_amd import kernels as K
    B, T, H = 256, 201, 12
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.randn(B * T, 3 * H * 64, generator=g) * 0.5).bfloat16().cuda()
    do = (torch.randn(B * T, H * 64, generator=g) * 0.5).bfloat16().cuda()
    step_like = len(sys.argv) > 1 and sys.argv[1] == "lo"       # as the training step calls it: O_lo + fused bias sums
    if step_like:
        o, lse, olo = K.attention_fwd(x, B, T, H, 0.125, want_lo=True)
    else:
        (o, lse), olo = K.attention_fwd(x, B, T, H, 0.125), None
    for _ in range(5):
        K.attention_bwd(x, o, do, lse, B, T, H, 0.125, want_bias=step_like, out_lo=olo)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(out)
    n = B * H * 8 * 32
    buf = np.zeros(n, dtype=np.uint64)
    rc = lib.getmi_debug_attn_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n * 8))
    assert rc == 0, rc
    st = buf.reshape(B * H, 8, 32).astype(np.int64)
    names = ["entry->setup", "setup->staged-issue", "issue->landed(barrier)", "barrier->K frags"]
    for w in range(8):
        s = st[256:2816, w, :]                     # steady-state workgroups
        d = np.diff(s[:, :5], axis=1)
        line = [f"{names[i]} {int(np.median(d[:, i]))}" for i in range(4)]
        main = np.median([s[:, 5 + 3 * k] - (s[:, 4] if k == 0 else s[:, 7 + 3 * (k - 1)]) for k in range(7)])
        barr = np.median([s[:, 6 + 3 * k] - s[:, 5 + 3 * k] for k in range(7)])
        dq = np.median([s[:, 7 + 3 * k] - s[:, 6 + 3 * k] for k in range(7)])
        tail = [int(np.median(s[:, 26] - s[:, 25])), int(np.median(s[:, 27] - s[:, 26])), int(np.median(s[:, 28] - s[:, 27]))]
        total = int(np.median(s[:, 28] - s[:, 0]))
        print(f"wave {w}: " + " | ".join(line))
        print(f"        per slice: main {int(main)}  barrier-wait {int(barr)}  dQ {int(dq)}   after-loop {tail}   total {total} cycles")
    wg = st[256:2816, :, 28].max(axis=1) - st[256:2816, :, 0].min(axis=1)
    print("workgroup lifetime median", int(np.median(wg)), "cycles (s_memtime ticks)")
