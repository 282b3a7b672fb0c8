"""A/B of GEMM builds / generations in separate processes is noisy (cdna guide rule 24), so this runs one process per
variant but interleaves rounds: usage  python tools/bench_gemm_ab.py <label>:<ENV=val>[,<ENV=val>...] ...
(e.g. old:GETMI_GEMM_GEN=1,GETMI_GEMM_GROUP=0 new:GETMI_GEMM_GEN=1).  Prints per-shape medians (us) per variant."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = []
for a in sys.argv[1:]:
    label, _, spec = a.partition(":")
    variants.append((label, dict(kv.split("=", 1) for kv in spec.split(",") if kv)))
res = {}
for rnd in range(3):
    for label, envs in variants:
        env = dict(os.environ)
        env.update(envs)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_gemm.py")], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if " us " in line:
                name = line[:34].strip(); us = float(line[34:].split()[0])
                res.setdefault(name, {}).setdefault(label, []).append(us)
print(f"{'shape':34s} " + " ".join(f"{l:>10s}" for l, _ in variants))
for name, d in res.items():
    print(f"{name:34s} " + " ".join(f"{sorted(d[l])[len(d[l]) // 2]:10.1f}" for l, _ in variants))
