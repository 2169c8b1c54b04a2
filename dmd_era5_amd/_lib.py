"""ctypes loader for libdmdx.so (the C ABI declared in include/dmdx.h).

The library is built in-tree (``dmd_era5_amd/libdmdx.so``) by
``__graft_entry__.build()`` / ``make -C dmd_era5_amd/csrc``.  There is no CPU
fallback: if the shared object is missing, loading raises and every kernel
entry point of :mod:`dmd_era5_amd.kernels` fails loudly.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DMDX_LIB_PATH: A/B experiments load an alternative build of the same ABI
LIB_PATH = os.environ.get("DMDX_LIB_PATH") or os.path.join(_HERE, "libdmdx.so")

_i64 = C.c_int64
_p = C.c_void_p
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/dmdx.h one to one
SIGNATURES = {
    "dmdx_version": (C.c_int, []),
    "dmdx_last_error": (C.c_char_p, []),
    "dmdx_syrk_workspace_bytes": (_sz, [_i64, _i64]),
    "dmdx_syrk_f32": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64, _p, _i64, C.c_int, _p, _sz, _p]),
    "dmdx_syrk_blocks_workspace_bytes": (_sz, [_p, C.c_int, _i64]),
    "dmdx_syrk_blocks_f32": (C.c_int, [_p, _p, _p, C.c_int, _i64, _p, _i64, _p, _i64, C.c_int, _p, _sz, _p]),
    "dmdx_gemm_tn_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "dmdx_gemm_tn_f32": (
        C.c_int,
        [_p, _i64, _p, _i64, _i64, _i64, _i64, _p, _i64, _p, _i64, C.c_int, _p, _sz, _p],
    ),
    "dmdx_gemm_tn_blocks_workspace_bytes": (_sz, [_p, C.c_int, _i64, _i64]),
    "dmdx_gemm_tn_blocks_f32": (
        C.c_int,
        [_p, _p, _p, _p, _p, C.c_int, _i64, _i64, _p, _i64, _p, _i64, C.c_int, _p, _sz, _p],
    ),
    "dmdx_gemm_nn_skinny_f32": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64, _i64, _p, _i64, _p]),
    "dmdx_gemm_nn_skinny_gram_max_l": (C.c_int, []),
    "dmdx_gemm_nn_skinny_gram_workspace_bytes": (_sz, [_i64, _i64]),
    "dmdx_gemm_nn_skinny_gram_f32": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, C.c_int,
                                               _p, _sz, _p]),
    "dmdx_row_center_scale_f32": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, C.c_int, _p]),
    "dmdx_delay_shift_sum_f64": (C.c_int, [_p, _i64, _i64, C.c_int, _p, _i64, _p, _i64, _p]),
    "dmdx_scale_columns_f32": (C.c_int, [_p, _i64, _i64, _i64, _p, _p]),
    "dmdx_eigh_small_max_n": (C.c_int, []),
    "dmdx_eigh_small_f64": (C.c_int, [_p, _i64, _i64, _p, _p, _i64, _p, _p]),
    "dmdx_svd_jacobi_max_n": (C.c_int, []),
    "dmdx_svd_jacobi_workspace_bytes": (_sz, [_i64]),
    "dmdx_svd_jacobi_f64": (C.c_int, [_p, _i64, _i64, _p, _p, _i64, _p, _p, _sz, _p]),
    "dmdx_symm_skinny_workspace_bytes": (_sz, [_i64, _i64]),
    "dmdx_symm_skinny_f64": (C.c_int, [_p, _i64, _i64, _p, _i64, _i64, C.c_double, _p, _i64, _p, _sz, _p]),
    "dmdx_gemm_tn_f64_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "dmdx_gemm_tn_f64": (C.c_int, [_p, _i64, _p, _i64, _i64, _i64, _i64, _p, _i64, _p, _sz, _p]),
    "dmdx_potrf_trtri_max_n": (C.c_int, []),
    "dmdx_potrf_trtri_workspace_bytes": (_sz, [_i64]),
    "dmdx_potrf_trtri_f64": (C.c_int, [_p, _i64, _i64, C.c_double, _p, _i64, _p, _i64, _p, _p, _sz, _p]),
    "dmdx_gemm_nt_f64": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64, _i64, _p, _i64, _p]),
    "dmdx_pack_triu_f64": (C.c_int, [_p, _i64, _i64, _p, _p]),
    "dmdx_unpack_triu_f64": (C.c_int, [_p, _i64, _p, _i64, _p]),
    "dmdx_exp_basis": (C.c_int, [_p, _p, _i64, _i64, _p, _p, C.c_int, _p]),
    "dmdx_set_clock_probe": (C.c_int, [_p]),
    "dmdx_calib_mfma_f32": (C.c_int, [C.c_int, C.c_int, _p, C.POINTER(C.c_double), _p]),
}

_lib = None


class DmdxError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libdmdx.so once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DmdxError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C dmd_era5_amd/csrc` (there is no CPU fallback)."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().dmdx_last_error().decode("utf-8", "replace")
        raise DmdxError(f"{what} failed (rc={rc}): {msg}")
