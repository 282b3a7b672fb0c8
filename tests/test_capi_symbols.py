"""CPU: the C-ABI library builds, loads, and exports exactly the symbols include/getmi355.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "getmi355.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(getmi_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from get_model_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        from get_model_amd.build import build
        build(verbose=False)
    handle = lib.load()
    declared = header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in getmi355.h but not exported"
    assert sorted(lib.EXPORTED_SYMBOLS) == declared, "ctypes signature table and header disagree"


def test_version_and_error_channel_without_gpu():
    from get_model_amd import lib
    h = lib.load()
    assert h.getmi_version() == 100
    assert h.getmi_arch() == b"gfx950"
    assert lib.query("getmi_gemm_wgrad_workspace_bytes", 51456, 768, 768) >= 768 * 768 * 4
    assert lib.query("getmi_layernorm_bwd_workspace_bytes", 51456, 768) == 3 * 512 * 768 * 4      # 2 workgroups per CU


def test_product_path_has_no_cpu_fallback():
    import torch
    from get_model_amd import kernels
    with pytest.raises(RuntimeError):
        kernels.cast_pad(torch.zeros(4, 8))
