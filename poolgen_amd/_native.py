"""ctypes binding of libpoolgen_hip.so (the C ABI declared in include/poolgen_hip.h)."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
_LIB = None


class NativeError(RuntimeError):
    """Raised when the HIP library is missing, fails to load, or a call returns an error."""


def library_path() -> Path:
    return Path(os.environ.get("POOLGEN_HIP_LIB", _HERE / "csrc" / "libpoolgen_hip.so"))


class PgFilter(C.Structure):
    _fields_ = [
        ("remove_ns", C.c_int32),
        ("reserved", C.c_int32),
        ("min_coverage_depth", C.c_uint64),
        ("min_allele_frequency", C.c_double),
        ("max_missingness_rate", C.c_double),
    ]


_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_u64 = C.c_uint64
_pi, _pd, _pi64 = C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int64)
_pf = C.POINTER(PgFilter)
_batch = [_vp, _vp, _i64, _i, _vp, _pf, _vp, _i, _vp, _vp, _vp, _vp, _vp]

SIGNATURES = {
    "pg_create": (_i, [C.POINTER(_vp), _i, _vp]),
    "pg_destroy": (None, [_vp]),
    "pg_last_error": (C.c_char_p, [_vp]),
    "pg_version": (C.c_char_p, []),
    "pg_synchronize": (_i, [_vp]),
    "pg_profile_enable": (_i, [_vp, _i]),
    "pg_profile_reset": (_i, [_vp]),
    "pg_profile_get": (_i, [_vp, _i, _pd, _pi64]),
    "pg_locus_op_stats": (_i, [_vp, _pi64, _pi64]),
    "pg_set_phenotypes": (_i, [_vp, _i, _vp, _i]),
    "pg_kinship_partial_dev": (_i, [_vp, _vp, _i64, _i, _i64, _vp]),
    "pg_kinship_set": (_i, [_vp, _vp, _i64, _i, _vp, _i, _d, _i, _pi, _vp, _vp]),
    "pg_covariates_set": (_i, [_vp, _i, _vp, _i, _vp, _i]),
    "pg_ols_sweep_dev": (_i, [_vp, _vp, _i64, _i, _i64, _vp, _vp, _vp]),
    "pg_ols_kinship_dev": (_i, [_vp, _vp, _i64, _i, _i64, _vp, _i, _d, _i, _pi, _vp, _vp, _vp, _vp]),
    "pg_ols_kinship": (_i, [_vp, _vp, _i64, _i, _i64, _vp, _i, _d, _i, _pi, _vp, _vp, _vp, _vp]),
    "pg_mle_kinship_dev": (_i, [_vp, _vp, _i64, _i, _i64, _vp, _i, _d, _i, _pi, _vp, _vp, _vp, _vp]),
    "pg_comm_unique_id": (_i, [_vp]),
    "pg_comm_init_rank": (_i, [_vp, _vp, _i, _i]),
    "pg_comm_destroy": (_i, [_vp]),
    "pg_comm_size": (_i, [_vp]),
    "pg_comm_rank": (_i, [_vp]),
    "pg_comm_version": (_i, [_pi]),
    "pg_allreduce_sum_dev": (_i, [_vp, _vp, _i64]),
    "pg_ols_kinship_sharded_dev": (_i, [_vp, _vp, _i64, _i64, _i, _i64, _vp, _i, _d, _i, _pi, _vp, _vp, _vp, _vp]),
    "pg_ols_iter_batch_dev": (_i, _batch),
    "pg_pearson_batch_dev": (_i, _batch),
    "pg_chisq_batch_dev": (_i, [_vp, _vp, _i64, _i, _vp, _pf, _vp, _vp, _vp, _vp]),
    "pg_load_plan_dev": (_i, [_vp, _vp, _i64, _i, _vp, _pf, _i, _vp, _vp]),
    "pg_load_emit_dev": (_i, [_vp, _vp, _i, _vp, _i64, _vp, _vp]),
    "pg_ols_iter_batch": (_i, _batch),
    "pg_pearson_batch": (_i, _batch),
    "pg_chisq_batch": (_i, [_vp, _vp, _i64, _i, _vp, _pf, _vp, _vp, _vp, _vp]),
    "pg_gp_xxt_dev": (_i, [_vp, _vp, _i64, _i, _i64, _vp]),
    "pg_gp_ols_dev": (_i, [_vp, _vp, _i64, _i, _i64, _vp, _i, _vp, _i, _vp, _vp]),
    "pg_gp_ridge_dev": (_i, [_vp, _vp, _i64, _i, _i64, _vp, _i, _vp, _i, _vp, _i, _i, _d, _d, _vp, _vp, _vp]),
    "pg_gp_penalised_dev": (_i, [_vp, _vp, _i64, _i, _i64, _vp, _i, _vp, _i, _vp, _i, _i, _d, _i, _d, _vp, _vp, _vp, _vp, _vp]),
    "pg_gp_proxy_dev": (_i, [_vp, _vp, _i64, _i, _i64, _vp, _i, _vp, _i, _vp, _vp]),
    "pg_expand_counts_u16_dev": (_i, [_vp, _vp, _i64, _vp]),
    "pg_load_emit_cov_dev": (_i, [_vp, _vp, _i, _vp, _i64, _vp, _vp, _vp]),
    "pg_host_sliding_windows": (_i64, [_vp, _vp, _i64, _u64, _u64, _u64, _vp, _vp]),
    "pg_pi_dev": (_i, [_vp, _vp, _vp, _i64, _i, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "pg_fst_dev": (_i, [_vp, _vp, _vp, _i64, _i, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "pg_gp_predict_dev": (_i, [_vp, _vp, _i64, _i, _i64, _vp, _i, _vp]),
    "pg_host_sym_eig": (_i, [_vp, _i, _vp, _vp]),
    "pg_host_sym_eig_top": (_i, [_vp, _i, _i, _vp, _vp]),
    "pg_host_n_eigenvecs": (_i, [_vp, _i, _d]),
    "pg_host_pinv_sym": (_i, [_vp, _i, _vp]),
    "pg_host_t_two_sided_p": (_d, [_d, _i]),
}

KERNEL_IDS = {"kinship": 0, "kinship_reduce": 1, "sweep": 2, "ols_iter": 3, "pearson": 4,
              "chisq": 5, "gp_xxt": 6, "gp_beta": 7, "sweep_finish": 8, "allreduce": 9, "gp_predict": 10}


def load_library():
    """Load libpoolgen_hip.so; fail loudly if it has not been built (no fallback exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not path.exists():
        raise NativeError(
            f"{path} not found: build it with `make -C {path.parent}` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). poolgen_amd has no CPU fallback.")
    try:
        lib = C.CDLL(str(path))
    except OSError as e:  # pragma: no cover
        raise NativeError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch, let it propagate
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib
