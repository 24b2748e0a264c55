"""Engine: torch-facing wrapper over one pg_ctx (one GPU, one stream).

Method names follow the reference operators they stand in for:
  ols_with_covariate   gwas::ols_with_covariate      (gwas/ols.rs:278-436)
  ols_iterate          gwas::ols_iterate             (gwas/ols.rs:201-276)
  correlation          gwas::correlation             (gwas/correlation_test.rs:73-129)
  chisq                tables::chisq                 (tables/chisq_test.rs:5-47)
  gp_ols               gp::ols                       (gp/ols.rs:8-101)
All heavy arguments are torch CUDA tensors (device memory owned by torch); results are
torch CUDA tensors.  Everything is computed by libpoolgen_hip.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from ._native import KERNEL_IDS, NativeError, PgFilter, load_library

ALLELES = "ATCGND"  # base/sync.rs:134 reader order


@dataclass
class Filter:
    """FilterStats subset (base/structs_and_traits.rs:68-78) with the CLI defaults (main.rs:44-58)."""
    remove_ns: bool = True
    min_coverage_depth: int = 1
    min_allele_frequency: float = 0.001
    max_missingness_rate: float = 0.0

    def to_c(self) -> PgFilter:
        return PgFilter(int(self.remove_ns), 0, int(self.min_coverage_depth),
                        float(self.min_allele_frequency), float(self.max_missingness_rate))


def _host_f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class Engine:
    def __init__(self, device: int | None = None, use_torch_stream: bool = True):
        self._lib = load_library()
        if not torch.cuda.is_available():
            raise NativeError("no GPU visible to torch: poolgen_amd needs an MI355X (gfx950) device")
        self.device = torch.cuda.current_device() if device is None else int(device)
        torch.cuda.set_device(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream if use_torch_stream else 0
        ctx = C.c_void_p()
        rc = self._lib.pg_create(C.byref(ctx), self.device, C.c_void_p(stream))
        if rc != 0:
            raise NativeError(f"pg_create failed ({rc}): {self._lib.pg_last_error(None).decode()}")
        self._ctx = ctx
        self._comm_ready = False   # True while this context holds an RCCL communicator (comm_init .. comm_destroy)

    def close(self):
        self._comm_ready = False
        if getattr(self, "_ctx", None):
            self._lib.pg_destroy(self._ctx)
            self._ctx = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise NativeError(f"{what} failed ({rc}): {self._lib.pg_last_error(self._ctx).decode()}")

    def _dev(self, t: torch.Tensor, dtype) -> int:
        if not (t.is_cuda and t.dtype == dtype and t.is_contiguous() and t.device.index == self.device):
            raise ValueError(f"expected a contiguous {dtype} tensor on cuda:{self.device}")
        return t.data_ptr()

    # ---- profiling ---------------------------------------------------------------------
    def profile(self, on: bool = True):
        self._check(self._lib.pg_profile_enable(self._ctx, int(on)), "pg_profile_enable")

    def profile_reset(self):
        self._check(self._lib.pg_profile_reset(self._ctx), "pg_profile_reset")

    def profile_get(self, kernel: str):
        ms, n = C.c_double(), C.c_int64()
        self._check(self._lib.pg_profile_get(self._ctx, KERNEL_IDS[kernel], C.byref(ms), C.byref(n)),
                    "pg_profile_get")
        return ms.value, n.value

    def synchronize(self):
        self._check(self._lib.pg_synchronize(self._ctx), "pg_synchronize")

    # ---- multi-GPU: RCCL inside the library (pg_comm.cpp) ------------------------------------
    def comm_unique_id(self) -> bytes:
        """ncclGetUniqueId through the library; one rank calls it and hands the bytes to the others."""
        buf = C.create_string_buffer(128)
        self._check(self._lib.pg_comm_unique_id(buf), "pg_comm_unique_id")
        return buf.raw

    def comm_init(self, unique_id: bytes, nranks: int, rank: int):
        """Collective: every rank of the node calls it with the same id (ncclCommInitRank on this engine's GPU)."""
        assert len(unique_id) == 128
        self._check(self._lib.pg_comm_init_rank(self._ctx, C.c_char_p(unique_id), int(nranks), int(rank)), "pg_comm_init_rank")
        self._comm_ready = True

    def comm_destroy(self):
        self._comm_ready = False   # allreduce_sum would be the identity from here on: callers must fall back
        self._check(self._lib.pg_comm_destroy(self._ctx), "pg_comm_destroy")

    @property
    def comm_size(self) -> int:
        return int(self._lib.pg_comm_size(self._ctx))

    @property
    def comm_rank(self) -> int:
        return int(self._lib.pg_comm_rank(self._ctx))

    def comm_version(self) -> int:
        v = C.c_int()
        self._check(self._lib.pg_comm_version(C.byref(v)), "pg_comm_version")
        return v.value

    def allreduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks of this engine's communicator (identity without one), on the engine's stream."""
        self._check(self._lib.pg_allreduce_sum_dev(self._ctx, self._dev(t, torch.float64), t.numel()), "pg_allreduce_sum_dev")
        return t

    def ols_with_covariate_sharded(self, G: torch.Tensor, p_total: int, Y, var_explained: float = 0.75, force_m: int = -1,
                                   n: int | None = None, out=None, want_K: bool = False):
        """One rank's share of ols_iter_with_kinship in ONE library call (pg_ols_kinship_sharded_dev): partial kinship,
        RCCL all-reduce, n x n step with p_total, sweep of the own slab.  Returns (m, K or None, beta, var, pval)."""
        p, ld, n = self._g_dims(G, n)
        Yh = _host_f64(Y).reshape(n, -1)
        k = Yh.shape[1]
        if out is None:
            out = torch.empty((3, p, k), dtype=torch.float64, device=G.device)
        K = np.empty((n, n)) if want_K else None
        m = C.c_int()
        self._check(self._lib.pg_ols_kinship_sharded_dev(self._ctx, self._dev(G, torch.float64), p, int(p_total), n, ld,
                                                         Yh.ctypes.data, k, float(var_explained), int(force_m), C.byref(m),
                                                         K.ctypes.data if want_K else None, out[0].data_ptr(),
                                                         out[1].data_ptr(), out[2].data_ptr()),
                    "pg_ols_kinship_sharded_dev")
        return m.value, K, out[0], out[1], out[2]

    # ---- kinship path ---------------------------------------------------------------------
    @staticmethod
    def _g_dims(G: torch.Tensor, n: int | None):
        p, ld = G.shape
        return int(p), int(ld), int(ld if n is None else n)

    def set_phenotypes(self, Y, n: int | None = None):
        """Announce Y before kinship_partial so the kinship pass can pre-compute the m = 0 fits."""
        if Y is None:
            self._check(self._lib.pg_set_phenotypes(self._ctx, 0, None, 0), "pg_set_phenotypes")
            return
        Yh = _host_f64(Y)
        Yh = Yh.reshape(len(Yh) if n is None else n, -1)
        self._check(self._lib.pg_set_phenotypes(self._ctx, Yh.shape[0], Yh.ctypes.data, Yh.shape[1]),
                    "pg_set_phenotypes")

    def kinship_partial(self, G: torch.Tensor, n: int | None = None) -> torch.Tensor:
        """Unscaled sum_l g_l g_l^T (n x n) over the loci of G (p x ld, locus-major)."""
        p, ld, n = self._g_dims(G, n)
        S = torch.empty((n, n), dtype=torch.float64, device=G.device)
        self._check(self._lib.pg_kinship_partial_dev(self._ctx, self._dev(G, torch.float64), p, n, ld,
                                                     self._dev(S, torch.float64)), "pg_kinship_partial_dev")
        return S

    def kinship_set(self, S: torch.Tensor, p_total: int, Y, var_explained: float = 0.75,
                    force_m: int = -1, want_evals: bool = False, want_K: bool = True):
        """K = S / p_total, eigen rule, covariates, projected phenotypes.  Returns (m, K, evals);
        evals is None unless want_evals (asking for them forces the full eigen-decomposition); K is None
        unless want_K."""
        n = S.shape[0]
        Yh = _host_f64(Y).reshape(n, -1)
        K = np.empty((n, n)) if want_K else None; m = C.c_int()
        ev = np.empty(n) if want_evals else None
        self._check(self._lib.pg_kinship_set(self._ctx, self._dev(S, torch.float64), int(p_total), n,
                                             Yh.ctypes.data, Yh.shape[1], float(var_explained),
                                             int(force_m), C.byref(m), K.ctypes.data if want_K else None,
                                             ev.ctypes.data if want_evals else None),
                    "pg_kinship_set")
        return m.value, K, ev

    def covariates_set(self, n: int, Cmat, Y):
        Yh = _host_f64(Y).reshape(n, -1)
        if Cmat is None or np.size(Cmat) == 0:
            m, cptr = 0, None
        else:
            Ch = _host_f64(Cmat).reshape(n, -1)
            m, cptr = Ch.shape[1], Ch.ctypes.data
        self._check(self._lib.pg_covariates_set(self._ctx, n, cptr, m, Yh.ctypes.data, Yh.shape[1]),
                    "pg_covariates_set")
        self._k = Yh.shape[1]

    def ols_sweep(self, G: torch.Tensor, k: int, n: int | None = None, out=None):
        p, ld, n = self._g_dims(G, n)
        if out is None:
            out = torch.empty((3, p, k), dtype=torch.float64, device=G.device)
        self._check(self._lib.pg_ols_sweep_dev(self._ctx, self._dev(G, torch.float64), p, n, ld,
                                               out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()),
                    "pg_ols_sweep_dev")
        return out[0], out[1], out[2]

    def ols_with_covariate(self, G: torch.Tensor, Y, var_explained: float = 0.75, force_m: int = -1,
                           n: int | None = None, out=None, want_K: bool = True):
        """Single-GPU ols_iter_with_kinship numeric core.  Returns (m, K, beta, var, pval).  want_K=False passes K_out = NULL:
        the library may then decide m = 0 from a bound that needs no kinship matrix (the lazy route, include/poolgen_hip.h)."""
        p, ld, n = self._g_dims(G, n)
        Yh = _host_f64(Y).reshape(n, -1)
        k = Yh.shape[1]
        if out is None:
            out = torch.empty((3, p, k), dtype=torch.float64, device=G.device)
        K = np.empty((n, n)) if want_K else None
        m = C.c_int()
        self._check(self._lib.pg_ols_kinship_dev(self._ctx, self._dev(G, torch.float64), p, n, ld,
                                                 Yh.ctypes.data, k, float(var_explained), int(force_m),
                                                 C.byref(m), K.ctypes.data if want_K else None, out[0].data_ptr(),
                                                 out[1].data_ptr(), out[2].data_ptr()),
                    "pg_ols_kinship_dev")
        return m.value, K, out[0], out[1], out[2]

    def mle_with_covariate(self, G: torch.Tensor, Y, var_explained: float = 0.75, force_m: int = -1, n: int | None = None):
        """gwas::mle_with_covariate (gwas/mle.rs:307-463), numeric core; parity unpinned (see include/poolgen_hip.h).
        Returns (m, K, beta, var, pval)."""
        p, ld, n = self._g_dims(G, n)
        Yh = _host_f64(Y).reshape(n, -1)
        k = Yh.shape[1]
        out = torch.empty((3, p, k), dtype=torch.float64, device=G.device)
        K = np.empty((n, n)); m = C.c_int()
        self._check(self._lib.pg_mle_kinship_dev(self._ctx, self._dev(G, torch.float64), p, n, ld, Yh.ctypes.data, k,
                                                 float(var_explained), int(force_m), C.byref(m), K.ctypes.data,
                                                 out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()), "pg_mle_kinship_dev")
        return m.value, K, out[0], out[1], out[2]

    # ---- sync-derived batch operators -------------------------------------------------------
    # The library's outputs are SLOT-MAJOR (include/poolgen_hip.h): element (slot r, locus l) of allele_ids / mean_freq at
    # r * L + l, of stat / pval at (r * L + l) * k + trait; slots r >= n_out[l] are unspecified.  raw=True hands those arrays
    # out as they are ((5, L) / (5, L, k): what bench.py times); the default returns locus-major (L, 5[, k]) copies with the
    # unspecified slots filled with -1 / NaN, which is what the parity tests index.
    def _batch(self, fn, name, counts: torch.Tensor, pool_sizes, flt: Filter, Y, raw: bool = False):
        L, n, six = counts.shape
        assert six == 6
        ps = _host_f64(pool_sizes)
        Yh = _host_f64(Y).reshape(n, -1)
        k = Yh.shape[1]
        dev = counts.device
        n_out = torch.empty(L, dtype=torch.int32, device=dev)
        ids = torch.empty((5, L), dtype=torch.int32, device=dev)
        mf = torch.empty((5, L), dtype=torch.float64, device=dev)
        stat = torch.empty((5, L, k), dtype=torch.float64, device=dev)
        pv = torch.empty((5, L, k), dtype=torch.float64, device=dev)
        f = flt.to_c()
        self._check(fn(self._ctx, self._dev(counts, torch.int32), L, n, ps.ctypes.data, C.byref(f),
                       Yh.ctypes.data, k, n_out.data_ptr(), ids.data_ptr(), mf.data_ptr(),
                       stat.data_ptr(), pv.data_ptr()), name)
        if raw:
            return n_out, ids, mf, stat, pv
        live = torch.arange(5, device=dev)[None, :] < n_out[:, None]
        nan = torch.tensor(float("nan"), dtype=torch.float64, device=dev)
        return (n_out, torch.where(live, ids.T, torch.tensor(-1, dtype=torch.int32, device=dev)), torch.where(live, mf.T, nan),
                torch.where(live[:, :, None], stat.permute(1, 0, 2), nan), torch.where(live[:, :, None], pv.permute(1, 0, 2), nan))

    def ols_iterate(self, counts, pool_sizes, flt: Filter, Y, raw: bool = False):
        return self._batch(self._lib.pg_ols_iter_batch_dev, "pg_ols_iter_batch_dev", counts,
                           pool_sizes, flt, Y, raw)

    def correlation(self, counts, pool_sizes, flt: Filter, Y, raw: bool = False):
        return self._batch(self._lib.pg_pearson_batch_dev, "pg_pearson_batch_dev", counts,
                           pool_sizes, flt, Y, raw)

    def chisq(self, counts, pool_sizes, flt: Filter, raw: bool = False):
        L, n, _ = counts.shape
        ps = _host_f64(pool_sizes)
        dev = counts.device
        n_out = torch.empty(L, dtype=torch.int32, device=dev)
        ids = torch.empty((5, L), dtype=torch.int32, device=dev)   # the surviving alleles of the row in the slots below n_out
        chi2 = torch.empty(L, dtype=torch.float64, device=dev)
        pv = torch.empty(L, dtype=torch.float64, device=dev)
        f = flt.to_c()
        self._check(self._lib.pg_chisq_batch_dev(self._ctx, self._dev(counts, torch.int32), L, n, ps.ctypes.data,
                                                 C.byref(f), n_out.data_ptr(), ids.data_ptr(),
                                                 chi2.data_ptr(), pv.data_ptr()), "pg_chisq_batch_dev")
        if raw:
            return n_out, ids, chi2, pv
        live = torch.arange(5, device=dev)[None, :] < n_out[:, None]
        return n_out, torch.where(live, ids.T, torch.tensor(-1, dtype=torch.int32, device=dev)), chi2, pv

    def last_listed(self):
        """(loci, listed) of the last batch operator call: how many loci its streaming pass handed to the second pass."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self._lib.pg_locus_op_stats(self._ctx, C.byref(a), C.byref(b)), "pg_locus_op_stats")
        return a.value, b.value

    def last_listed_fraction(self) -> float:
        loci, listed = self.last_listed()
        return listed / loci if loci else 0.0

    def load_frequencies(self, counts, pool_sizes, flt: Filter, keep_p_minus_1: bool = False, order=None,
                         pool_keep=None, ld: int | None = None, coverages: bool = False):
        """The reference loader (base/sync.rs:972-1180) on a counts batch in HBM: filter, frequencies over the
        surviving alleles, optionally drop the major allele; one column of G per surviving allele.  Returns
        (G [p x ld], col_locus [p], col_allele [p]); `pool_keep` (bool per pool) selects the rows written.
        coverages=True appends cov [p x ld]: per column row the pools' depths over the locus' surviving alleles."""
        L, n, _ = counts.shape
        ps = _host_f64(pool_sizes)
        dev = counts.device
        f = flt.to_c()
        p = C.c_int64()
        optr = None
        if order is not None:
            order = order.to(device=dev, dtype=torch.int64).contiguous()
            optr = order.data_ptr()
        self._check(self._lib.pg_load_plan_dev(self._ctx, self._dev(counts, torch.int32), L, n, ps.ctypes.data,
                                               C.byref(f), int(keep_p_minus_1), optr, C.byref(p)), "pg_load_plan_dev")
        if pool_keep is None:
            pmap, n_out = None, n
        else:
            keep = np.asarray(pool_keep, dtype=bool)
            pmap = np.where(keep, np.cumsum(keep) - 1, -1).astype(np.int32)
            n_out = int(keep.sum())
        ld = (n_out + (n_out & 1)) if ld is None else int(ld)
        G = torch.empty((p.value, ld), dtype=torch.float64, device=dev)
        col_locus = torch.empty(p.value, dtype=torch.int64, device=dev)
        col_allele = torch.empty(p.value, dtype=torch.int32, device=dev)
        if coverages:
            cov = torch.empty((p.value, ld), dtype=torch.float64, device=dev)
            self._check(self._lib.pg_load_emit_cov_dev(self._ctx, pmap.ctypes.data if pmap is not None else None, n_out,
                                                       G.data_ptr(), ld, col_locus.data_ptr(), col_allele.data_ptr(),
                                                       cov.data_ptr()), "pg_load_emit_cov_dev")
            return G, col_locus, col_allele, cov
        self._check(self._lib.pg_load_emit_dev(self._ctx, pmap.ctypes.data if pmap is not None else None, n_out,
                                               G.data_ptr(), ld, col_locus.data_ptr(), col_allele.data_ptr()),
                    "pg_load_emit_dev")
        return G, col_locus, col_allele

    # ---- popgen -------------------------------------------------------------------------------
    def sliding_windows(self, chrom_ids, pos, window_size_bp: int, window_slide_size_bp: int, min_loci_per_window: int):
        """define_sliding_windows (base/helpers.rs:294-403) -> (head, tail) inclusive locus index ranges."""
        ch = np.ascontiguousarray(chrom_ids, dtype=np.int32)
        po = np.ascontiguousarray(pos, dtype=np.uint64)
        head = np.empty(max(len(ch), 1), dtype=np.int64); tail = np.empty(max(len(ch), 1), dtype=np.int64)
        nw = self._lib.pg_host_sliding_windows(ch.ctypes.data, po.ctypes.data, len(ch), int(window_size_bp),
                                               int(window_slide_size_bp), int(min_loci_per_window), head.ctypes.data,
                                               tail.ctypes.data)
        return head[:nw].copy(), tail[:nw].copy()

    def _popgen_args(self, G, cov, locus_col, win_head, win_tail, n):
        p, ld, n = self._g_dims(G, n)
        assert cov.shape == G.shape
        lc = np.ascontiguousarray(locus_col, dtype=np.int64)
        wh = np.ascontiguousarray(win_head, dtype=np.int64); wt = np.ascontiguousarray(win_tail, dtype=np.int64)
        return p, ld, n, lc, wh, wt

    def theta_pi(self, G, cov, locus_col, win_head, win_tail, n: int | None = None):
        """popgen::theta_pi (popgen/pi.rs:10-113): (pi per window [n_windows x n], mean across windows [n])."""
        p, ld, n, lc, wh, wt = self._popgen_args(G, cov, locus_col, win_head, win_tail, n)
        win = np.empty((len(wh), n)); mean = np.empty(n)
        self._check(self._lib.pg_pi_dev(self._ctx, self._dev(G, torch.float64), self._dev(cov, torch.float64), p, n, ld,
                                        lc.ctypes.data, len(lc) - 1, wh.ctypes.data, wt.ctypes.data, len(wh),
                                        win.ctypes.data, mean.ctypes.data), "pg_pi_dev")
        return win, mean

    def fst(self, G, cov, locus_col, win_head, win_tail, n: int | None = None):
        """popgen::fst (popgen/fst.rs:10-115, :158-200): (genome-wide mean [n x n], per window [n_windows x n*n])."""
        p, ld, n, lc, wh, wt = self._popgen_args(G, cov, locus_col, win_head, win_tail, n)
        mean = np.empty((n, n)); win = np.empty((len(wh), n * n))
        self._check(self._lib.pg_fst_dev(self._ctx, self._dev(G, torch.float64), self._dev(cov, torch.float64), p, n, ld,
                                         lc.ctypes.data, len(lc) - 1, wh.ctypes.data, wt.ctypes.data, len(wh),
                                         mean.ctypes.data, win.ctypes.data), "pg_fst_dev")
        return mean, win

    # ---- genomic prediction -------------------------------------------------------------------
    def gp_xxt(self, G: torch.Tensor, n: int | None = None) -> torch.Tensor:
        p, ld, n = self._g_dims(G, n)
        S = torch.empty((n, n), dtype=torch.float64, device=G.device)
        self._check(self._lib.pg_gp_xxt_dev(self._ctx, self._dev(G, torch.float64), p, n, ld,
                                            S.data_ptr()), "pg_gp_xxt_dev")
        return S

    def gp_ols(self, G: torch.Tensor, Y, row_idx, XXt=None, n: int | None = None):
        p, ld, n = self._g_dims(G, n)
        Yh = _host_f64(Y).reshape(n, -1)
        k = Yh.shape[1]
        ri = np.ascontiguousarray(np.asarray(row_idx, dtype=np.int64))
        beta = torch.empty((p + 1, k), dtype=torch.float64, device=G.device)
        xptr = None
        if XXt is not None:
            XXt = _host_f64(XXt)
            xptr = XXt.ctypes.data
        self._check(self._lib.pg_gp_ols_dev(self._ctx, self._dev(G, torch.float64), p, n, ld,
                                            Yh.ctypes.data, k, ri.ctypes.data, len(ri), xptr,
                                            beta.data_ptr()), "pg_gp_ols_dev")
        return beta

    def gp_predict(self, G: torch.Tensor, beta: torch.Tensor, n: int | None = None) -> np.ndarray:
        """yhat = [1 | G^T] beta for every pool (n x k on the host)."""
        p, ld, n = self._g_dims(G, n)
        k = beta.shape[1]
        yhat = np.empty((n, k))
        self._check(self._lib.pg_gp_predict_dev(self._ctx, self._dev(G, torch.float64), p, n, ld,
                                                self._dev(beta, torch.float64), k, yhat.ctypes.data), "pg_gp_predict_dev")
        return yhat

    def gp_penalised(self, G: torch.Tensor, Y, row_idx, fold_of, n_folds: int, alpha: float, iterative_proxy: bool = False,
                     lambda_step: float = 0.1, n: int | None = None, XXt=None):
        """Every model behind penalised_lambda_path_with_k_fold_cross_validation (gp/penalise.rs:461-669): alpha in
        [0, 1] one lambda path, alpha < 0 the alpha x lambda grid of penalise_glmnet, iterative_proxy the
        *_with_iterative_proxy_norms variants.  Returns (beta (1+p) x k on the device, alphas[k], lambdas[k],
        perf[n_reps, n_folds, A, L, k])."""
        p, ld, n = self._g_dims(G, n)
        Yh = _host_f64(Y).reshape(n, -1)
        k = Yh.shape[1]
        ri = np.ascontiguousarray(np.asarray(row_idx, dtype=np.int64))
        fo = np.ascontiguousarray(np.asarray(fold_of, dtype=np.int32)).reshape(-1, len(ri))
        L = int(round(1.0 / lambda_step)) + 1
        A = 1 if alpha >= 0 else L
        beta = torch.empty((p + 1, k), dtype=torch.float64, device=G.device)
        al, lam = np.empty(k), np.empty(k)
        perf = np.empty((fo.shape[0], n_folds, A, L, k))
        xh = None if XXt is None else _host_f64(XXt)   # kept alive across the call
        self._check(self._lib.pg_gp_penalised_dev(self._ctx, self._dev(G, torch.float64), p, n, ld, Yh.ctypes.data, k,
                                                  ri.ctypes.data, len(ri), fo.ctypes.data, fo.shape[0], int(n_folds),
                                                  float(alpha), int(bool(iterative_proxy)), float(lambda_step),
                                                  beta.data_ptr(), al.ctypes.data, lam.ctypes.data, perf.ctypes.data,
                                                  None if xh is None else xh.ctypes.data),
                    "pg_gp_penalised_dev")
        return beta, al, lam, perf

    def gp_proxy(self, G: torch.Tensor, Y, row_idx, n: int | None = None) -> torch.Tensor:
        """ols_iterative_with_kinship_pca_covariate (gp/ols.rs:104-199): (1+p) x k on the device."""
        p, ld, n = self._g_dims(G, n)
        Yh = _host_f64(Y).reshape(n, -1)
        k = Yh.shape[1]
        ri = np.ascontiguousarray(np.asarray(row_idx, dtype=np.int64))
        out = torch.empty((p + 1, k), dtype=torch.float64, device=G.device)
        self._check(self._lib.pg_gp_proxy_dev(self._ctx, self._dev(G, torch.float64), p, n, ld, Yh.ctypes.data, k,
                                              ri.ctypes.data, len(ri), None, out.data_ptr()), "pg_gp_proxy_dev")
        return out

    def gp_ridge(self, G: torch.Tensor, Y, row_idx, fold_of, n_folds: int, alpha: float = 0.0,
                 lambda_step: float = 0.1, n: int | None = None):
        """penalise_ridge_like (gp/penalise.rs:133-159) with explicit folds: fold_of is (n_reps, len(row_idx)).
        Returns (beta (1+p) x k on the device, lambdas[k], perf[n_reps, n_folds, L, k])."""
        p, ld, n = self._g_dims(G, n)
        Yh = _host_f64(Y).reshape(n, -1)
        k = Yh.shape[1]
        ri = np.ascontiguousarray(np.asarray(row_idx, dtype=np.int64))
        fo = np.ascontiguousarray(np.asarray(fold_of, dtype=np.int32)).reshape(-1, len(ri))
        L = int(round(1.0 / lambda_step)) + 1
        beta = torch.empty((p + 1, k), dtype=torch.float64, device=G.device)
        lam = np.empty(k)
        perf = np.empty((fo.shape[0], n_folds, L, k))
        self._check(self._lib.pg_gp_ridge_dev(self._ctx, self._dev(G, torch.float64), p, n, ld, Yh.ctypes.data, k,
                                              ri.ctypes.data, len(ri), fo.ctypes.data, fo.shape[0], int(n_folds),
                                              float(alpha), float(lambda_step), beta.data_ptr(), lam.ctypes.data,
                                              perf.ctypes.data), "pg_gp_ridge_dev")
        return beta, lam, perf
