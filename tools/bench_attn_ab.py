"""Interleaved A/B of attention kernels under different environment settings (one child process per variant and round).
usage: python tools/bench_attn_ab.py label:ENV=val,... ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = [(a.partition(":")[0], dict(kv.split("=", 1) for kv in a.partition(":")[2].split(",") if kv)) for a in sys.argv[1:]]
res = {}
for rnd in range(3):
    for label, envs in variants:
        env = dict(os.environ); env.update(envs)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_attn.py")], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if "T=201" in line or "T=2001" in line:
                res.setdefault(line.split(":")[0], {}).setdefault(label, []).append(line.split(":", 1)[1].strip())
for k, d in res.items():
    for label, _ in variants:
        print(k, label, " | ".join(d.get(label, [])))
