"""GPU: determinism / race screen of the pipelined kernels at the bench shape (tools/race_screen.py)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_pipelined_kernels_are_bit_reproducible_under_memory_contention():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    # a child process: the screen holds ~6 GB of operands and must not share the allocator state of the other tests
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "race_screen.py"), "20"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "RACE SCREEN clean" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
