"""GPU: getmi_scale_cast_noalloc (include/getmi355.h) -- the entry point that allocates nothing -- against
getmi_scale_cast, through the C ABI: bit-identical outputs (same kernels, same partial-row order) on two calls in a
row, the second with a smaller input (the shared scratch buffer keeps the first call's partial rows), plus the
size limit.  Collected last on purpose (file name): see the tier rules."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _call_noalloc(lib, dx, rowscale, T):
    """Straight through ctypes: no torch allocation between the argument set-up and the call."""
    M, D = dx.shape
    dz = torch.empty(M, D, dtype=torch.bfloat16, device=dx.device)
    dbias = torch.empty(D, dtype=torch.float32, device=dx.device)
    stream = torch.cuda.current_stream().cuda_stream
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    rc = lib.getmi_scale_cast_noalloc(dx.data_ptr(), None if rowscale is None else rowscale.data_ptr(), T, dz.data_ptr(),
                                      dbias.data_ptr(), 0, M, D, stream)
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == before
    assert rc == 0, lib.getmi_last_error().decode()
    return dz, dbias


def test_noalloc_entry_point_equals_the_workspace_one_on_two_calls_in_a_row():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from get_model_amd import kernels as K
    from get_model_amd import lib as L
    lib = L.load()
    assert lib.getmi_scale_cast_noalloc_max_bytes() == 3 << 20
    g = torch.Generator(device="cpu").manual_seed(5)
    for (B, T, D) in ((64, 201, 768), (3, 37, 768), (8, 201, 1024), (1, 5, 256)):    # large, then smaller; two widths
        M = B * T
        dx = torch.randn(M, D, generator=g).cuda()
        scale = (torch.rand(B, generator=g) < 0.8).float().div(0.8).cuda()
        for rs in (scale, None):
            dz_ref, db_ref = K.scale_cast(dx, rs, T)
            dz, db = _call_noalloc(lib, dx, rs, T)
            assert torch.equal(dz, dz_ref), (B, T, D)
            assert torch.equal(db, db_ref), (B, T, D)
            # and against plain torch on the same inputs
            ref = dx if rs is None else dx * rs.repeat_interleave(T).unsqueeze(1)
            assert torch.equal(dz, ref.bfloat16())
            torch.testing.assert_close(db, ref.sum(0), rtol=1e-4, atol=1e-3)
    # the Python wrapper takes the same path
    dz2, db2 = K.scale_cast_noalloc(dx, None, 0)
    assert torch.equal(dz2, dz) and torch.equal(db2, db)


def test_noalloc_entry_point_rejects_what_does_not_fit():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from get_model_amd import lib as L
    lib = L.load()
    M, D = 4100, 1024          # 1024 partial rows x 1024 columns x 4 B = 4 MiB > 3 MiB
    assert lib.getmi_scale_cast_workspace_bytes(M, D) > lib.getmi_scale_cast_noalloc_max_bytes()
    dx = torch.zeros(M, D, device="cuda")
    dz = torch.full((M, D), 7.0, dtype=torch.bfloat16, device="cuda")
    db = torch.full((D,), 7.0, device="cuda")
    rc = lib.getmi_scale_cast_noalloc(dx.data_ptr(), None, 0, dz.data_ptr(), db.data_ptr(), 0, M, D,
                                      torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"scratch" in lib.getmi_last_error()
    torch.cuda.synchronize()
    assert bool((dz == 7).all()) and bool((db == 7).all())       # nothing was launched
    with pytest.raises(RuntimeError, match="scratch"):
        L.call("getmi_scale_cast_noalloc", dx, None, 0, dz, db, 0, M, D)


def test_noalloc_entry_point_inside_a_hip_graph_capture():
    """the header names HIP graph capture as a caller.  After the per-device set-up query (outside capture)
    a call made while a stream is capturing must enqueue only kernel nodes: captured once, replayed on new inputs,
    bit-identical to the workspace entry point each time."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from get_model_amd import kernels as K
    from get_model_amd import lib as L
    lib = L.load()
    assert lib.getmi_scale_cast_noalloc_max_bytes() == 3 << 20          # the set-up call: resolves the scratch address
    B, T, D = 16, 201, 768
    M = B * T
    g = torch.Generator(device="cpu").manual_seed(9)
    dx = torch.randn(M, D, generator=g).cuda()
    scale = (torch.rand(B, generator=g) < 0.8).float().div(0.8).cuda()
    dz = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    dbias = torch.empty(D, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        rc = lib.getmi_scale_cast_noalloc(dx.data_ptr(), scale.data_ptr(), T, dz.data_ptr(), dbias.data_ptr(), 0, M, D,
                                          torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.getmi_last_error().decode()
    for seed in (1, 2):
        dx.copy_(torch.randn(M, D, generator=g))
        dz.zero_()
        dbias.zero_()
        graph.replay()
        torch.cuda.synchronize()
        dz_ref, db_ref = K.scale_cast(dx, scale, T)
        assert torch.equal(dz, dz_ref) and torch.equal(dbias, db_ref), seed
