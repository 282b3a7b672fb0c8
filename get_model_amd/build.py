"""Build libgetmi355.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

    python -m get_model_amd.build            # or __graft_entry__.build()

One object per .hip file (compiled in parallel), linked into get_model_amd/libgetmi355.so.  hipcc
cross-compiles for gfx950 without a GPU present.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libgetmi355.so")
SOURCES = ["error.hip", "gemm.hip", "attention.hip", "norm.hip", "misc.hip", "optim.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-ffp-contract=fast"]
FLAGS += os.environ.get("GETMI_EXTRA_HIPCC_FLAGS", "").split()      # experiments (A/B builds via GETMI_LIB)


def _stamp(src):
    h = hashlib.sha1()
    for f in [src, os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "getmi355.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(name):
    src = os.path.join(CSRC, name)
    obj = os.path.join(OBJ, name.replace(".hip", ".o"))
    stamp_file = obj + ".stamp"
    stamp = _stamp(src)
    if os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj, False
    cmd = ["hipcc"] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {name}:\n{r.stdout}\n{r.stderr}")
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return obj, True


def build(verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=6) as ex:
        res = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in res]
    if any(changed for _, changed in res) or not os.path.exists(LIB):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"built {LIB}")
    elif verbose:
        print(f"{LIB} is up to date")
    return LIB


if __name__ == "__main__":
    build()
    sys.exit(0)
