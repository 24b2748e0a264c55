"""ctypes wrapper over oracle/liboracle.so -- the CPU restatement used ONLY as the checker."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ODIR = ROOT / "oracle"
ALLELES = "ATCGND"


class OrcFilter(C.Structure):
    _fields_ = [("remove_ns", C.c_int), ("min_coverage_depth", C.c_uint64),
                ("min_allele_frequency", C.c_double), ("max_missingness_rate", C.c_double)]


class OrcHdr(C.Structure):
    _fields_ = [("n_alleles", C.c_int), ("allele_ids", C.c_int * 5), ("mean_freq", C.c_double * 5)]


def build(name="liboracle.so"):
    so = ODIR / name
    src = [ODIR / "poolgen_oracle.c", ODIR / "poolgen_oracle.h", ODIR / "poolgen_exact.c"]
    if not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in src):
        subprocess.check_call(["make", "-C", str(ODIR), "-s"])
    return so


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        d, i, i64, vp = C.c_double, C.c_int, C.c_int64, C.c_void_p
        lib.orc_students_t_cdf.restype = d; lib.orc_students_t_cdf.argtypes = [d, d]
        lib.orc_chisq_cdf.restype = d; lib.orc_chisq_cdf.argtypes = [d, d]
        lib.orc_beta_reg.restype = d; lib.orc_beta_reg.argtypes = [d, d, d]
        lib.orc_ln_gamma.restype = d; lib.orc_ln_gamma.argtypes = [d]
        lib.orc_gamma_lr.restype = d; lib.orc_gamma_lr.argtypes = [d, d]
        lib.orc_sensible_round.restype = d; lib.orc_sensible_round.argtypes = [d, i]
        lib.orc_fmt_display.argtypes = [d, C.c_char_p, i]
        lib.orc_parse_f64_roundup_and_own.argtypes = [d, i, C.c_char_p, i]
        lib.orc_ndarray_sum.restype = d; lib.orc_ndarray_sum.argtypes = [vp, i64]
        lib.orc_mean_ignore_nan.restype = d; lib.orc_mean_ignore_nan.argtypes = [vp, i64, i64]
        lib.orc_lu_inverse.argtypes = [vp, i, vp]
        lib.orc_lu_det.restype = d; lib.orc_lu_det.argtypes = [vp, i]
        lib.orc_sym_eig.argtypes = [vp, i, vp, vp]
        lib.orc_pinv_sym.argtypes = [vp, i, vp]
        lib.orc_parse_sync_line.argtypes = [C.c_char_p, C.c_char_p, i, C.POINTER(C.c_uint64), vp, i]
        lib.orc_filter_locus.argtypes = [vp, i, vp, C.POINTER(OrcFilter), vp, vp]
        lib.orc_to_frequencies.argtypes = [vp, i, i, vp]
        lib.orc_sort_by_allele_freq.argtypes = [vp, i, i, vp, i]
        lib.orc_ols_fit.argtypes = [vp, vp, i, i, vp, vp, vp, vp]
        lib.orc_ols_iterate_locus.argtypes = [vp, i, vp, i, vp, C.POINTER(OrcFilter), C.POINTER(OrcHdr), vp, vp]
        lib.orc_ols_iterate_csv.argtypes = [C.c_char_p, C.c_uint64, vp, i, vp, i, vp, C.POINTER(OrcFilter), C.c_char_p, i]
        lib.orc_correlation_locus.argtypes = lib.orc_ols_iterate_locus.argtypes
        lib.orc_correlation_csv.argtypes = lib.orc_ols_iterate_csv.argtypes
        lib.orc_chisq_locus.argtypes = [vp, i, vp, C.POINTER(OrcFilter), vp, C.POINTER(d), C.POINTER(d)]
        lib.orc_chisq_csv.argtypes = [C.c_char_p, C.c_uint64, vp, i, vp, C.POINTER(OrcFilter), C.c_char_p, i]
        lib.orc_ols_with_covariate.argtypes = [vp, i64, i, i64, vp, i, d, i, vp, vp, vp, vp, vp, vp, vp, i]
        lib.orc_n_eigenvecs_rule.argtypes = [vp, i, d]
        lib.orc_mle_fit.argtypes = [vp, vp, i, i, vp, vp, vp, vp]
        lib.orc_mle_with_covariate.argtypes = [vp, i64, i, i64, vp, i, d, i, vp, vp, vp, vp, i]
        lib.orc_kinship.argtypes = [vp, i64, i, i64, vp, i]
        lib.orc_pearsons_correlation.argtypes = [vp, i64, vp, i64, i, C.POINTER(d), C.POINTER(d)]
        lib.orc_multiply_views_xx.argtypes = [vp, i, vp, i, vp, i, vp, vp, i, vp, i, vp]
        lib.orc_multiply_views_xtx.argtypes = [vp, i, vp, i, vp, i, vp, i, vp, vp, i, vp]
        lib.orc_multiply_views_xxt.argtypes = [vp, i, vp, i, vp, i, vp, i, vp, i, vp, vp]
        lib.orc_gp_ols.argtypes = [vp, i64, i, i64, vp, i, vp, i, vp, i]
        lib.orc_expand_and_contract.argtypes = [vp, vp, i64, i, d, d, vp]
        lib.orc_error_index.argtypes = [vp, i64, i, i64, vp, i, vp, vp, i, vp]
        lib.orc_penalised_lambda_path.argtypes = [vp, i64, i, i64, vp, i, vp, i, vp, i, i, d, d, vp, vp, vp, i]
        lib.orc_penalised_path_general.argtypes = [vp, i64, i, i64, vp, i, vp, i, vp, i, i, d, i, d, vp, vp, vp, vp, i]
        lib.orc_gp_proxy.argtypes = [vp, i64, i, i64, vp, i, vp, i, vp, i]
        lib.orc_define_sliding_windows.restype = i64
        lib.orc_define_sliding_windows.argtypes = [vp, vp, i64, C.c_uint64, C.c_uint64, C.c_uint64, vp, vp]
        lib.orc_fst.argtypes = [vp, i64, i, i64, vp, i64, vp, vp, vp, i64, vp, vp]
        lib.orc_theta_pi.argtypes = [vp, i64, i, i64, vp, i64, vp, vp, vp, i64, vp, vp]

    # ---- small conveniences -------------------------------------------------------------
    @staticmethod
    def filt(remove_ns=True, min_cov=1, maf=0.001, miss=0.0):
        return OrcFilter(int(remove_ns), int(min_cov), float(maf), float(miss))

    def fmt(self, x: float) -> str:
        b = C.create_string_buffer(512)
        self.lib.orc_fmt_display(float(x), b, 512)
        return b.value.decode()

    def round_own(self, x: float, nd: int) -> str:
        b = C.create_string_buffer(512)
        self.lib.orc_parse_f64_roundup_and_own(float(x), nd, b, 512)
        return b.value.decode()

    def parse_sync_line(self, line: str, max_pools=4096):
        chrom = C.create_string_buffer(256)
        pos = C.c_uint64()
        counts = np.zeros((max_pools, 6), dtype=np.uint64)
        n = self.lib.orc_parse_sync_line(line.encode(), chrom, 256, C.byref(pos), counts.ctypes.data, max_pools)
        if n <= 0:
            return n, None, None, None
        return n, chrom.value.decode(), pos.value, counts[:n].copy()

    def pileup_to_sync(self, line, pool_sizes, remove_ns=True, max_base_error_rate=0.01, min_coverage_depth=1,
                       min_coverage_breadth=1.0, min_allele_frequency=0.001, keep_lowercase_reference=False):
        """One pileup line -> (code, sync line): code > 0 ok, 0 = None (dropped), < 0 = the reference panics."""
        ps = np.ascontiguousarray(pool_sizes, dtype=np.float64)
        buf = C.create_string_buffer(64 + 80 * max(len(ps), line.count("\t")))
        fn = self.lib.orc_pileup_to_sync2
        fn.restype = C.c_int
        fn.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_double, C.c_double, C.c_void_p, C.c_int,
                       C.c_char_p, C.c_int]
        rc = fn(line.encode("latin-1"), int(remove_ns), int(keep_lowercase_reference), float(max_base_error_rate), int(min_coverage_depth),
                float(min_coverage_breadth), float(min_allele_frequency), ps.ctypes.data, len(ps), buf, len(buf))
        return rc, (buf.value.decode("latin-1") if rc > 0 else "")

    def filter_locus(self, counts, pool_sizes, f):
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        n = counts.shape[0]
        ps = np.ascontiguousarray(pool_sizes, dtype=np.float64)
        ids = np.zeros(6, dtype=np.int32)
        out = np.zeros((n, 6), dtype=np.uint64)
        a = self.lib.orc_filter_locus(counts.ctypes.data, n, ps.ctypes.data, C.byref(f), ids.ctypes.data, out.ctypes.data)
        if a == 0:
            return None
        return ids[:a].copy(), out.reshape(-1)[: n * a].reshape(n, a).copy()

    def to_frequencies(self, counts):
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        n, a = counts.shape
        fr = np.empty((n, a))
        self.lib.orc_to_frequencies(counts.ctypes.data, n, a, fr.ctypes.data)
        return fr

    def sort_by_allele_freq(self, fr, ids, decreasing=True):
        fr = np.ascontiguousarray(fr, dtype=np.float64).copy()
        ids = np.ascontiguousarray(ids, dtype=np.int32).copy()
        n, a = fr.shape
        self.lib.orc_sort_by_allele_freq(fr.ctypes.data, n, a, ids.ctypes.data, int(decreasing))
        return fr, ids

    def ols_fit(self, X, y):
        X = np.ascontiguousarray(X, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        n, P = X.shape
        b, v, t, p = (np.empty(P) for _ in range(4))
        rc = self.lib.orc_ols_fit(X.ctypes.data, y.ctypes.data, n, P, b.ctypes.data, v.ctypes.data, t.ctypes.data, p.ctypes.data)
        return rc, b, v, t, p

    def _locus_op(self, fn, counts, Y, pool_sizes, f):
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        n = counts.shape[0]
        Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(n, -1)
        k = Y.shape[1]
        ps = np.ascontiguousarray(pool_sizes, dtype=np.float64)
        h = OrcHdr()
        s = np.full((5, k), np.nan); p = np.full((5, k), np.nan)
        na = fn(counts.ctypes.data, n, Y.ctypes.data, k, ps.ctypes.data, C.byref(f), C.byref(h), s.ctypes.data, p.ctypes.data)
        if na <= 0:
            return na, None, None, None, None
        s2 = s.reshape(-1)[: na * k].reshape(na, k).copy(); p2 = p.reshape(-1)[: na * k].reshape(na, k).copy()
        return na, list(h.allele_ids)[:na], list(h.mean_freq)[:na], s2, p2

    def ols_iterate_locus(self, counts, Y, pool_sizes, f):
        return self._locus_op(self.lib.orc_ols_iterate_locus, counts, Y, pool_sizes, f)

    def correlation_locus(self, counts, Y, pool_sizes, f):
        return self._locus_op(self.lib.orc_correlation_locus, counts, Y, pool_sizes, f)

    def chisq_locus(self, counts, pool_sizes, f):
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        n = counts.shape[0]
        ps = np.ascontiguousarray(pool_sizes, dtype=np.float64)
        ids = np.zeros(6, dtype=np.int32); chi2 = C.c_double(); pv = C.c_double()
        a = self.lib.orc_chisq_locus(counts.ctypes.data, n, ps.ctypes.data, C.byref(f), ids.ctypes.data, C.byref(chi2), C.byref(pv))
        if a == 0:
            return 0, None, None, None
        return a, ids[:a].copy(), chi2.value, pv.value

    def _csv(self, fn, chrom, pos, counts, Y, pool_sizes, f):
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        n = counts.shape[0]
        ps = np.ascontiguousarray(pool_sizes, dtype=np.float64)
        buf = C.create_string_buffer(1 << 16)
        if Y is None:
            nb = fn(chrom.encode(), int(pos), counts.ctypes.data, n, ps.ctypes.data, C.byref(f), buf, 1 << 16)
        else:
            Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(n, -1)
            nb = fn(chrom.encode(), int(pos), counts.ctypes.data, n, Y.ctypes.data, Y.shape[1], ps.ctypes.data, C.byref(f), buf, 1 << 16)
        return buf.value.decode() if nb > 0 else None

    def ols_iterate_csv(self, chrom, pos, counts, Y, pool_sizes, f):
        return self._csv(self.lib.orc_ols_iterate_csv, chrom, pos, counts, Y, pool_sizes, f)

    def correlation_csv(self, chrom, pos, counts, Y, pool_sizes, f):
        return self._csv(self.lib.orc_correlation_csv, chrom, pos, counts, Y, pool_sizes, f)

    def chisq_csv(self, chrom, pos, counts, pool_sizes, f):
        return self._csv(self.lib.orc_chisq_csv, chrom, pos, counts, None, pool_sizes, f)

    def kinship(self, G, n=None, threads=0):
        G = np.ascontiguousarray(G, dtype=np.float64)
        p, ld = G.shape
        n = ld if n is None else n
        K = np.empty((n, n))
        self.lib.orc_kinship(G.ctypes.data, p, n, ld, K.ctypes.data, threads)
        return K

    def ols_with_covariate(self, G, Y, var_explained=0.75, force_m=-1, covariate=None, n=None, threads=0):
        G = np.ascontiguousarray(G, dtype=np.float64)
        p, ld = G.shape
        n = ld if n is None else n
        Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(n, -1)
        k = Y.shape[1]
        K = np.empty((n, n)); ev = np.empty(n); cov = np.zeros((n, n))
        beta, var, pv = (np.empty((p, k)) for _ in range(3))
        cptr = None
        if covariate is not None:
            covariate = np.ascontiguousarray(covariate, dtype=np.float64).reshape(n, -1)
            force_m = covariate.shape[1]
            cptr = covariate.ctypes.data
        m = self.lib.orc_ols_with_covariate(G.ctypes.data, p, n, ld, Y.ctypes.data, k, float(var_explained), int(force_m),
                                            cptr, K.ctypes.data, ev.ctypes.data, cov.ctypes.data, beta.ctypes.data,
                                            var.ctypes.data, pv.ctypes.data, threads)
        return dict(m=m, K=K, evals=ev, cov=cov.reshape(-1)[: n * m].reshape(n, m).copy(), beta=beta, var=var, pval=pv)

    def mle_fit(self, X, y):
        X = np.ascontiguousarray(X, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        n, P = X.shape
        b, v, t, p = (np.empty(P) for _ in range(4))
        rc = self.lib.orc_mle_fit(X.ctypes.data, y.ctypes.data, n, P, b.ctypes.data, v.ctypes.data, t.ctypes.data, p.ctypes.data)
        return rc, b, v, t, p

    def mle_with_covariate(self, G, Y, var_explained=0.75, force_m=-1, covariate=None, n=None, threads=0):
        G = np.ascontiguousarray(G, dtype=np.float64)
        p, ld = G.shape
        n = ld if n is None else n
        Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(n, -1)
        k = Y.shape[1]
        beta, var, pv = (np.empty((p, k)) for _ in range(3))
        cptr = None
        if covariate is not None:
            covariate = np.ascontiguousarray(covariate, dtype=np.float64).reshape(n, -1)
            force_m = covariate.shape[1]
            cptr = covariate.ctypes.data
        m = self.lib.orc_mle_with_covariate(G.ctypes.data, p, n, ld, Y.ctypes.data, k, float(var_explained), int(force_m), cptr,
                                            beta.ctypes.data, var.ctypes.data, pv.ctypes.data, threads)
        return dict(m=m, beta=beta, var=var, pval=pv)

    def n_eigenvecs(self, ev, thr):
        ev = np.ascontiguousarray(ev, dtype=np.float64)
        return self.lib.orc_n_eigenvecs_rule(ev.ctypes.data, len(ev), float(thr))

    def pearson(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        r = C.c_double(); p = C.c_double()
        self.lib.orc_pearsons_correlation(x.ctypes.data, 1, y.ctypes.data, 1, len(x), C.byref(r), C.byref(p))
        return r.value, p.value

    def sym_eig(self, A):
        A = np.ascontiguousarray(A, dtype=np.float64); n = A.shape[0]
        ev = np.empty(n); V = np.empty((n, n))
        self.lib.orc_sym_eig(A.ctypes.data, n, ev.ctypes.data, V.ctypes.data)
        return ev, V

    def gp_ols(self, Xt, Y, row_idx, n=None, threads=0):
        Xt = np.ascontiguousarray(Xt, dtype=np.float64)
        P, ld = Xt.shape
        n = ld if n is None else n
        Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(n, -1)
        ri = np.ascontiguousarray(row_idx, dtype=np.int64)
        beta = np.empty((P, Y.shape[1]))
        rc = self.lib.orc_gp_ols(Xt.ctypes.data, P, n, ld, Y.ctypes.data, Y.shape[1], ri.ctypes.data, len(ri), beta.ctypes.data, threads)
        return rc, beta

    def penalised_lambda_path(self, Xt, Y, row_idx, fold_of, n_folds, alpha=0.0, lambda_step=0.1, n=None, threads=0):
        Xt = np.ascontiguousarray(Xt, dtype=np.float64)
        P, ld = Xt.shape
        n = ld if n is None else n
        Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(n, -1)
        k = Y.shape[1]
        ri = np.ascontiguousarray(row_idx, dtype=np.int64)
        fo = np.ascontiguousarray(fold_of, dtype=np.int32).reshape(-1, len(ri))
        L = int(round(1.0 / lambda_step)) + 1
        beta = np.empty((P, k)); lam = np.empty(k); perf = np.empty((fo.shape[0], n_folds, L, k))
        self.lib.orc_penalised_lambda_path(Xt.ctypes.data, P, n, ld, Y.ctypes.data, k, ri.ctypes.data, len(ri),
                                           fo.ctypes.data, fo.shape[0], n_folds, float(alpha), float(lambda_step),
                                           beta.ctypes.data, lam.ctypes.data, perf.ctypes.data, threads)
        return beta, lam, perf

    def penalised_path_general(self, Xt, Y, row_idx, fold_of, n_folds, alpha, iterative, lambda_step=0.1, n=None, threads=0):
        """every mode of the path (alpha < 0: the alpha x lambda grid; iterative: proxy norms).
        Returns beta, alphas, lambdas, perf[reps, folds, A, L, k]."""
        Xt = np.ascontiguousarray(Xt, dtype=np.float64)
        P, ld = Xt.shape
        n = ld if n is None else n
        Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(n, -1)
        k = Y.shape[1]
        ri = np.ascontiguousarray(row_idx, dtype=np.int64)
        fo = np.ascontiguousarray(fold_of, dtype=np.int32).reshape(-1, len(ri))
        L = int(round(1.0 / lambda_step)) + 1
        A = 1 if alpha >= 0 else L
        beta = np.empty((P, k)); al = np.empty(k); lam = np.empty(k); perf = np.empty((fo.shape[0], n_folds, A, L, k))
        self.lib.orc_penalised_path_general(Xt.ctypes.data, P, n, ld, Y.ctypes.data, k, ri.ctypes.data, len(ri),
                                            fo.ctypes.data, fo.shape[0], n_folds, float(alpha), int(bool(iterative)),
                                            float(lambda_step), beta.ctypes.data, al.ctypes.data, lam.ctypes.data,
                                            perf.ctypes.data, threads)
        return beta, al, lam, perf

    def gp_proxy(self, Xt, Y, row_idx, n=None, threads=0):
        """ols_iterative_with_kinship_pca_covariate (gp/ols.rs:104-199): P x k"""
        Xt = np.ascontiguousarray(Xt, dtype=np.float64)
        P, ld = Xt.shape
        n = ld if n is None else n
        Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(n, -1)
        ri = np.ascontiguousarray(row_idx, dtype=np.int64)
        b = np.empty((P, Y.shape[1]))
        self.lib.orc_gp_proxy(Xt.ctypes.data, P, n, ld, Y.ctypes.data, Y.shape[1], ri.ctypes.data, len(ri), b.ctypes.data, threads)
        return b

    # ---- popgen ----------------------------------------------------------------------------
    def sliding_windows(self, chrom, pos, window_size_bp, window_slide_size_bp, min_loci_per_window):
        """define_sliding_windows (helpers.rs:294-403); chrom: any hashables"""
        ids = {}
        ch = np.array([ids.setdefault(c, len(ids)) for c in chrom], dtype=np.int32)
        po = np.ascontiguousarray(pos, dtype=np.uint64)
        l = len(ch)
        head = np.empty(max(l, 1), dtype=np.int64); tail = np.empty(max(l, 1), dtype=np.int64)
        nw = self.lib.orc_define_sliding_windows(ch.ctypes.data, po.ctypes.data, l, window_size_bp, window_slide_size_bp,
                                                 min_loci_per_window, head.ctypes.data, tail.ctypes.data)
        return head[:nw].copy(), tail[:nw].copy()

    @staticmethod
    def count_loci(chrom, pos):
        """count_loci (sync.rs:73-97) on the labels WITH the leading intercept entry"""
        idx, lc, lp = [], [], []
        for i in range(1, len(chrom)):
            if chrom[i - 1] != chrom[i] or pos[i - 1] != pos[i]:
                idx.append(i); lc.append(chrom[i]); lp.append(pos[i])
        idx.append(len(chrom)); lc.append(chrom[-1]); lp.append(pos[-1])
        return idx, lc, lp

    def fst(self, Xt, loci_idx, cov, win_head, win_tail, n=None):
        Xt = np.ascontiguousarray(Xt, dtype=np.float64)
        P, ld = Xt.shape
        n = ld if n is None else n
        li = np.ascontiguousarray(loci_idx, dtype=np.int64); L = len(li) - 1
        cov = np.ascontiguousarray(cov, dtype=np.float64).reshape(L, n)
        wh = np.ascontiguousarray(win_head, dtype=np.int64); wt = np.ascontiguousarray(win_tail, dtype=np.int64)
        mean = np.empty((n, n)); win = np.empty((len(wh), n * n))
        rc = self.lib.orc_fst(Xt.ctypes.data, P, n, ld, li.ctypes.data, L, cov.ctypes.data, wh.ctypes.data, wt.ctypes.data,
                              len(wh), mean.ctypes.data, win.ctypes.data)
        return rc, mean, win

    def theta_pi(self, Xt, loci_idx, cov, win_head, win_tail, n=None):
        Xt = np.ascontiguousarray(Xt, dtype=np.float64)
        P, ld = Xt.shape
        n = ld if n is None else n
        li = np.ascontiguousarray(loci_idx, dtype=np.int64); L = len(li) - 1
        cov = np.ascontiguousarray(cov, dtype=np.float64).reshape(L, n)
        wh = np.ascontiguousarray(win_head, dtype=np.int64); wt = np.ascontiguousarray(win_tail, dtype=np.int64)
        win = np.empty((len(wh), n)); mean = np.empty(n)
        self.lib.orc_theta_pi(Xt.ctypes.data, P, n, ld, li.ctypes.data, L, cov.ctypes.data, wh.ctypes.data, wt.ctypes.data,
                              len(wh), win.ctypes.data, mean.ctypes.data)
        return win, mean

    def expand_and_contract(self, b, bp, alpha, lam):
        b = np.ascontiguousarray(b, dtype=np.float64); bp = np.ascontiguousarray(bp, dtype=np.float64)
        P, k = b.shape
        out = np.empty_like(b)
        self.lib.orc_expand_and_contract(b.ctypes.data, bp.ctypes.data, P, k, float(alpha), float(lam), out.ctypes.data)
        return out


class Exact:
    """oracle/libexact.so: the binary128 arbiter (poolgen_exact.c).  Same inputs as the oracle, results rounded once."""

    def __init__(self, lib, oracle_lib_handle=None):
        self.lib = lib
        self._orc = oracle_lib_handle
        d, i, i64, vp = C.c_double, C.c_int, C.c_int64, C.c_void_p
        lib.exq_t_two_sided_p.restype = d; lib.exq_t_two_sided_p.argtypes = [d, i]
        lib.exq_ols_covariate.argtypes = [vp, i64, i, i64, vp, i, vp, i, vp, vp, vp, vp, i]
        lib.exq_kinship.argtypes = [vp, i64, i, i64, vp, i]
        lib.exq_sym_eig.argtypes = [vp, i, vp, vp]
        lib.exq_kinship_covariates.argtypes = [vp, i64, i, i64, d, i, vp, vp, vp, i]
        lib.exq_gp_ols.argtypes = [vp, i64, i, i64, vp, i, vp, i, vp, i]
        lib.exq_gp_proxy.argtypes = [vp, i64, i, i64, vp, i, vp, i, vp, i]

    def t_two_sided_p(self, t_abs, df):
        return self.lib.exq_t_two_sided_p(float(t_abs), int(df))

    def ols_covariate(self, G, Y, C_=None, n=None, threads=0):
        """exact cells of ols_with_covariate for given covariates (n x m or None) -> dict(beta, var, t, pval), each p x k"""
        G = np.ascontiguousarray(G, dtype=np.float64)
        p, ld = G.shape
        n = ld if n is None else n
        Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(n, -1)
        k = Y.shape[1]
        m, cptr = 0, None
        if C_ is not None:
            C_ = np.ascontiguousarray(C_, dtype=np.float64).reshape(n, -1)
            m, cptr = C_.shape[1], C_.ctypes.data
        beta, var, t, pv = (np.empty((p, k)) for _ in range(4))
        rc = self.lib.exq_ols_covariate(G.ctypes.data, p, n, ld, Y.ctypes.data, k, cptr, m, beta.ctypes.data, var.ctypes.data,
                                        t.ctypes.data, pv.ctypes.data, threads)
        assert rc == 0, rc
        return dict(beta=beta, var=var, t=t, pval=pv)

    def kinship(self, G, n=None, threads=0):
        G = np.ascontiguousarray(G, dtype=np.float64)
        p, ld = G.shape
        n = ld if n is None else n
        K = np.empty((n, n))
        self.lib.exq_kinship(G.ctypes.data, p, n, ld, K.ctypes.data, threads)
        return K

    def sym_eig(self, A):
        A = np.ascontiguousarray(A, dtype=np.float64); n = A.shape[0]
        ev = np.empty(n); V = np.empty((n, n))
        self.lib.exq_sym_eig(A.ctypes.data, n, ev.ctypes.data, V.ctypes.data)
        return ev, V

    def kinship_covariates(self, G, var_explained=0.75, force_m=-1, n=None, threads=0):
        """K, eigenvalues and the m leading eigenvectors, binary128 end to end -> (m, K, evals, C n x m)"""
        G = np.ascontiguousarray(G, dtype=np.float64)
        p, ld = G.shape
        n = ld if n is None else n
        K = np.empty((n, n)); ev = np.empty(n); Cb = np.zeros((n, n))
        m = self.lib.exq_kinship_covariates(G.ctypes.data, p, n, ld, float(var_explained), int(force_m), K.ctypes.data,
                                            ev.ctypes.data, Cb.ctypes.data, threads)
        return m, K, ev, Cb.reshape(-1)[: n * m].reshape(n, m).copy()

    def ols_with_covariate(self, G, Y, var_explained=0.75, force_m=-1, n=None, threads=0):
        """the whole of ols_with_covariate's numeric core in binary128 (K -> eig -> rule -> fits)"""
        m, K, ev, Cm = self.kinship_covariates(G, var_explained, force_m, n, threads)
        out = self.ols_covariate(G, Y, Cm if m > 0 else None, n, threads)
        out.update(m=m, K=K, evals=ev, cov=Cm)
        return out

    def gp_ols(self, Xt, Y, row_idx, n=None, threads=0):
        Xt = np.ascontiguousarray(Xt, dtype=np.float64)
        P, ld = Xt.shape
        n = ld if n is None else n
        Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(n, -1)
        ri = np.ascontiguousarray(row_idx, dtype=np.int64)
        beta = np.empty((P, Y.shape[1]))
        rc = self.lib.exq_gp_ols(Xt.ctypes.data, P, n, ld, Y.ctypes.data, Y.shape[1], ri.ctypes.data, len(ri), beta.ctypes.data, threads)
        return rc, beta

    def gp_proxy(self, Xt, Y, row_idx, n=None, threads=0):
        """gp::ols_iterative_with_kinship_pca_covariate (gp/ols.rs:104-199) in binary128 -> P x k"""
        Xt = np.ascontiguousarray(Xt, dtype=np.float64)
        P, ld = Xt.shape
        n = ld if n is None else n
        Y = np.ascontiguousarray(Y, dtype=np.float64).reshape(n, -1)
        ri = np.ascontiguousarray(row_idx, dtype=np.int64)
        b = np.empty((P, Y.shape[1]))
        rc = self.lib.exq_gp_proxy(Xt.ctypes.data, P, n, ld, Y.ctypes.data, Y.shape[1], ri.ctypes.data, len(ri), b.ctypes.data, threads)
        assert rc == 0, rc
        return b

    def install_into_oracle(self, oracle, on=True):
        """the oracle's penalised path then takes its fold fits from exq_gp_ols and its proxy coefficients from exq_gp_proxy
        (everything downstream stays the oracle's)"""
        fn = C.cast(self.lib.exq_gp_ols, C.c_void_p) if on else C.c_void_p(None)
        oracle.lib.orc_set_gp_ols_hook.argtypes = [C.c_void_p]
        oracle.lib.orc_set_gp_ols_hook.restype = None
        oracle.lib.orc_set_gp_ols_hook(fn)
        fn2 = C.cast(self.lib.exq_gp_proxy, C.c_void_p) if on else C.c_void_p(None)
        oracle.lib.orc_set_gp_proxy_hook.argtypes = [C.c_void_p]
        oracle.lib.orc_set_gp_proxy_hook.restype = None
        oracle.lib.orc_set_gp_proxy_hook(fn2)


_ORACLE = None
_EXACT = None


def load_exact() -> Exact:
    global _EXACT
    if _EXACT is None:
        _EXACT = Exact(C.CDLL(str(build("libexact.so"))))
    return _EXACT


def load() -> Oracle:
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle(C.CDLL(str(build())))
    return _ORACLE
