"""CPU tests of the C-ABI library: it loads without a GPU, exports exactly the
symbols include/dmdx.h declares, and rejects bad arguments before touching HIP."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "dmd_era5_amd", "libdmdx.so")
    if not os.path.exists(so):
        import __graft_entry__

        __graft_entry__.build()
    from dmd_era5_amd import _lib

    return _lib.load()


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "dmdx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmdx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from dmd_era5_amd import _lib

    declared = _header_symbols()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in dmdx.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"


def test_version_and_error_string(lib):
    assert lib.dmdx_version() == 110
    assert isinstance(lib.dmdx_last_error(), bytes)


def test_argument_errors_are_reported_without_a_gpu(lib):
    rc = lib.dmdx_syrk_f32(None, 10, 10, 10, None, 10, None, 10, 0, None, 0, None)
    assert rc == -1000  # DMDX_E_INVALID
    assert b"null" in lib.dmdx_last_error()
    rc = lib.dmdx_gemm_nn_skinny_f32(None, 1, 1, 1, None, 1, 1, None, 1, None)
    assert rc == -1000
    rc = lib.dmdx_delay_shift_sum_f64(None, 4, 4, 2, None, 3, None, 0, None)
    assert rc == -1000
    rc = lib.dmdx_eigh_small_f64(None, 4, 4, None, None, 4, None, None)
    assert rc == -1000
    rc = lib.dmdx_syrk_blocks_f32(None, None, None, 0, 8, None, 8, None, 0, 0, None, 0, None)
    assert rc == -1000 and b"syrk_blocks" in lib.dmdx_last_error()
    assert lib.dmdx_eigh_small_max_n() == 96


def test_workspace_queries(lib):
    # 1 tile, >= 1 split; a Gram of <= 96 columns runs as one 64- / 96-row tile (64*128 / 96*128
    # doubles per partial tile), wider ones as 128*128 tiles
    assert lib.dmdx_syrk_workspace_bytes(1000, 64) % (64 * 128 * 8) == 0
    assert lib.dmdx_syrk_workspace_bytes(1000, 64) > 0
    assert lib.dmdx_syrk_workspace_bytes(1000, 90) % (96 * 128 * 8) == 0
    assert lib.dmdx_syrk_workspace_bytes(1000, 200) % (128 * 128 * 8) == 0
    big = lib.dmdx_syrk_workspace_bytes(129780, 8760)
    assert 0 < big < 8 << 30
    assert lib.dmdx_gemm_tn_workspace_bytes(100000, 8760, 70) > 0
    # batched K1: the slabs of all blocks of a launch (16 blocks at most) at once
    import ctypes as C

    ms = (C.c_int64 * 8)(*([129780] * 8))
    batched = lib.dmdx_syrk_blocks_workspace_bytes(ms, 8, 8760)
    assert batched % (128 * 128 * 8) == 0 and big <= batched <= 8 * big
    many = (C.c_int64 * 40)(*([5000] * 40))
    assert lib.dmdx_syrk_blocks_workspace_bytes(many, 40, 70) > 0
    assert lib.dmdx_syrk_blocks_workspace_bytes(None, 0, 70) == 0


def test_workspace_planners_survive_degenerate_shapes():
    """The planners are host code and run for every shape a caller hands in -- including row blocks
    shorter than one 64-row chunk of the small-l X^T Y body (a fuzz run died of SIGFPE there:
    0 chunks -> 0 chunks per split -> division by zero in dmdx_gemm_tn_blocks_workspace_bytes)."""
    import ctypes as C
    import itertools

    from dmd_era5_amd import _lib

    lib = _lib.load()
    for sizes, na, nb in itertools.product([[1], [63, 1], [5, 64, 7000], [64] * 17, [1] * 40, [3000, 2]],
                                           [1, 127, 128, 129, 500, 8760], [1, 2, 20, 32, 33, 96, 97, 130, 220]):
        ks = (C.c_int64 * len(sizes))(*sizes)
        w = lib.dmdx_gemm_tn_blocks_workspace_bytes(ks, len(sizes), na, nb)
        assert w > 0, (sizes, na, nb)
        assert lib.dmdx_syrk_blocks_workspace_bytes(ks, len(sizes), na) > 0
    for K, na, nb in itertools.product([1, 31, 32, 33, 64], [1, 33, 129], [1, 32, 97]):
        assert lib.dmdx_gemm_tn_workspace_bytes(K, na, nb) > 0 and lib.dmdx_syrk_workspace_bytes(K, na) > 0
    for m, l in itertools.product([1, 3, 64, 1000], [1, 2, 31, 96]):
        assert lib.dmdx_gemm_nn_skinny_gram_workspace_bytes(m, l) >= 0
    for n, b in itertools.product([2, 3, 8760], [2, 62, 124, 4096]):
        assert lib.dmdx_symm_skinny_workspace_bytes(n, b) >= 0
    for n, b1, b2 in itertools.product([1, 2, 8760], [2, 78], [2, 124]):
        assert lib.dmdx_gemm_tn_f64_workspace_bytes(n, b1, b2) >= 0
    for n in (2, 3, 78, 1024, 1025):
        assert lib.dmdx_svd_jacobi_workspace_bytes(n) >= 0


def test_product_has_no_cpu_fallback():
    """Without a GPU the kernel provider must refuse to construct."""
    import torch

    from dmd_era5_amd import _lib, kernels

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DmdxError, match="no CPU fallback"):
        kernels.HipKernels()
