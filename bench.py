#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json:

    loci/sec ols_iter_with_kinship, 200 pools x 10M loci, 1/2/4/8 MI355X

One "step" = one full pass of the hot path over the resident genotype matrix: partial kinship
(fp64 MFMA) -> [all-reduce over ranks] -> n x n eigen step + basis (host, replicated) -> per-locus
OLS sweep.  Inputs are synthetic and resident in HBM before the timed region.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), loci sharded contiguously,
total work fixed (strong scaling, as the metric is quoted on 10M loci in total).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

# torch / numpy are imported by the RANK processes only (worker()): the launching parent of `--gpus N` must never initialise a GPU

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD MI355X datasheet, fp64 matrix (not in the local guide)


def cpu_baseline(G_sample_host, Y, var_explained: float, force_m: int):
    """The oracle (a port of the reference algorithm), timed on this box's host cores."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    o = oracle_lib.load()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # a container's CPU share (cgroup v2 quota) is the honest core count, not the host's
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("POOLGEN_BENCH_CPU_THREADS", "16")))  # one GPU's CPU share on this pool
    probe = min(50000, G_sample_host.shape[0])
    t0 = time.perf_counter()
    o.ols_with_covariate(G_sample_host[:probe], Y, var_explained, force_m, threads=cores)
    t_probe = time.perf_counter() - t0
    target = 12.0
    s = int(min(G_sample_host.shape[0], max(probe, probe * target / max(t_probe, 1e-6))))
    # the sample is bounded by host memory (--cpu-sample); repeat the pass until ~10 s of CPU work are timed
    reps, dt = 0, 0.0
    while dt < 10.0 and reps < 20:
        t0 = time.perf_counter()
        o.ols_with_covariate(G_sample_host[:s], Y, var_explained, force_m, threads=cores)
        dt += time.perf_counter() - t0
        reps += 1
    return {"value": reps * s / dt, "unit": "loci/s", "cores": cores, "kind": "port",
            "sample": f"first {s} of the same synthetic loci (n={G_sample_host.shape[1]}), {reps} passes, kinship + eig + "
                      f"per-locus LU fits, OpenMP over loci, {dt:.1f} s"}


def secondary_legs(eng, dev, args, torch, np):
    """BASELINE configs[1] (ols_iter / pearson_corr / chisq_test from counts, 100 pools x 1M loci) and configs[3] (ridge lambda path
    with 10 x 10-fold CV, 500 pools x 5M loci), each a few launches on resident synthetic inputs with HIP-event kernel times
    (pg_profile_get: one event pair per operator call, on the library's stream).  Untimed with respect to the headline `value`."""
    from poolgen_amd import Filter, synth
    t_all = time.perf_counter()
    sec = {"what": "untimed legs after the headline region; kernel_ms = HIP events around the operator's kernels on the library's "
                   "stream (mean over `launches`); frac = algorithmic bytes (or flops) per launch / kernel time / peak"}
    # configs[1]: 24 n bytes of counts per locus (integer front-end, bytes/locus = 4*6*n)
    n1, L1 = 100, int(args.secondary_loci)
    G1 = synth.genotype_matrix(min(L1, 1 << 18), n1, dev)
    Y1 = synth.phenotypes(G1, n1, k=1)
    del G1
    ps = np.full(n1, 20.0)

    def run_ops(counts, flt):
        ops = {}
        for name, fn, kid in (("ols_iter", lambda: eng.ols_iterate(counts, ps, flt, Y1, raw=True), "ols_iter"),
                              ("pearson_corr", lambda: eng.correlation(counts, ps, flt, Y1, raw=True), "pearson"),
                              ("chisq_test", lambda: eng.chisq(counts, ps, flt, raw=True), "chisq")):
            fn(); fn()
            eng.profile_reset()
            reps = 10
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(dev)
            wall = (time.perf_counter() - t0) / reps
            ms, cnt = eng.profile_get(kid)
            kms = ms / max(cnt, 1)
            by = 24.0 * n1 * L1
            loci, listed = eng.last_listed()
            ops[name] = {"kernel_ms": kms, "launches": int(cnt), "bytes_per_launch": by, "achieved_gbs": by / (kms * 1e-3) / 1e9,
                         "frac": by / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "wall_ms_per_call": wall * 1e3, "loci_per_s": L1 / (kms * 1e-3),
                         "deferred_fraction": listed / max(loci, 1)}
        return ops

    counts = synth.sync_counts(L1, n1, dev)
    sec["count_operators"] = {"config": f"BASELINE configs[1]: synthetic sync counts {n1} pools x {L1} loci, 1 trait, CLI default filter; "
                                        f"clean counts (A and T only)",
                              "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "bytes_per_locus": 24.0 * n1, **run_ops(counts, Filter())}
    del counts
    # the same operators on counts that look like real pool-seq data: every read misread with probability `error_rate` onto one of the
    # other five sync columns, so nearly every locus carries reads of alleles the MAF filter drops, and the reference recomputes the
    # frequencies on the FILTERED counts (gwas/ols.rs:210-230 -> base/sync.rs:166-192).  kernel_ms = device time of the streaming
    # pass AND of the second pass over the loci it could not close in place (deferred_fraction), HIP events around both.
    real = {"what": "count operators on error-bearing counts (synth.sync_counts(error_rate=...)): each read lands on one of the other five "
                    "sync columns with that probability; frac = 24 n bytes per locus / (streaming pass + second pass device time) / 8 TB/s; "
                    "deferred_fraction = loci handed to the second pass",
            "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "bytes_per_locus": 24.0 * n1}
    for tag, err, maf, note in () if args.no_realistic else (
            ("error_0.005_maf_0.01", 0.005, 0.01, "the headline case of this leg: --min-allele-frequency 0.01 drops every error allele (pooled frequency 0.001 "
                                                  "each), every locus stays biallelic with stray reads in ~20 % of its pools"),
            ("error_0.001_default_filter", 0.001, 0.001, "CLI default filter (maf 0.001): error alleles at 0.0002 each are dropped"),
            ("error_0.005_default_filter", 0.005, 0.001, "CLI default filter on 0.5 % errors: an error allele's pooled frequency (0.001) sits ON the "
                                                         "threshold, half of them SURVIVE -- 86 % of the loci become 3- to 5-allelic joint fits for the "
                                                         "reference too, all of them second-pass work")):
        counts = synth.sync_counts(L1, n1, dev, error_rate=err)
        real[tag] = {"error_rate": err, "min_allele_frequency": maf, "note": note, **run_ops(counts, Filter(min_allele_frequency=maf))}
        del counts
    if not args.no_realistic:
        for op in ("ols_iter", "pearson_corr", "chisq_test"):   # the leg's own figures = its headline case
            real[op] = real["error_0.005_maf_0.01"][op]
        sec["count_operators_realistic"] = real
    # configs[3]: ridge path (alpha = 0), 11 lambda, 10 repetitions x 10 folds, folds fixed by fold[i] = (i + rep) mod 10
    n3, p3, reps3, folds3 = 500, int(args.ridge_loci), 10, 10
    try:
        G3 = synth.genotype_matrix(p3, n3, dev)
        Y3 = synth.phenotypes(G3[:100000], n3, k=1)
        rows = np.arange(n3)
        fold_of = np.stack([(rows + r) % folds3 for r in range(reps3)]).astype(np.int32)
        eng.gp_ridge(G3, Y3, rows, fold_of, folds3, n=n3)
        eng.profile_reset()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        beta, lam, perf = eng.gp_ridge(G3, Y3, rows, fold_of, folds3, n=n3)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        x_ms, x_n = eng.profile_get("gp_xxt")
        b_ms, b_n = eng.profile_get("gp_beta")
        pr_ms, pr_n = eng.profile_get("gp_predict")
        gb = 8.0 * n3 * p3
        xfl = 2.0 * n3 * n3 * (p3 + 1)
        rd = {"config": f"BASELINE configs[3]: ridge lambda path, {n3} pools x {p3} loci, {reps3} repetitions x {folds3} folds, 11 lambda",
              "wall_s": wall, "lambda": [float(x) for x in np.asarray(lam).ravel()],
              "xxt": {"kernel_ms": x_ms / max(x_n, 1), "launches": int(x_n), "flops_per_launch": xfl, "bound": "mfma",
                      "achieved_tflops_algorithmic": xfl / (x_ms / max(x_n, 1) * 1e-3) / 1e12 if x_n else None,
                      "note": "2 n^2 (p+1) / time as SURVEY 8d counts it; one triangle is computed, so this is not a utilisation"},
              "coefficient_pass": {"kernel_ms": b_ms / max(b_n, 1), "launches": int(b_n), "bytes_per_launch": gb, "bound": "hbm",
                                   "achieved_gbs": gb / (b_ms / max(b_n, 1) * 1e-3) / 1e9 if b_n else None,
                                   "frac": gb / (b_ms / max(b_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS if b_n else None},
              "prediction_pass": {"kernel_ms": pr_ms / max(pr_n, 1), "launches": int(pr_n), "bytes_per_launch": gb, "bound": "hbm",
                                  "achieved_gbs": gb / (pr_ms / max(pr_n, 1) * 1e-3) / 1e9 if pr_n else None,
                                  "frac": gb / (pr_ms / max(pr_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS if pr_n else None},
              "passes_over_G": int(b_n + pr_n + x_n)}
        sec["ridge"] = rd
        del G3
    except Exception as e:   # a leg that cannot run (e.g. not enough free HBM next to the headline's G) is reported, not fatal
        sec["ridge"] = {"error": f"{type(e).__name__}: {e}"}
    sec["seconds"] = time.perf_counter() - t_all
    return sec


def shard_probe_legs(eng, G, Y, n, p_total, var_explained, steps, torch, np):
    """What ONE rank of an N-rank run does per step, measured on this one GPU with the REAL library communicator (a one-rank RCCL
    communicator: ncclCommInitRank + ncclAllReduce inside libpoolgen_hip, the path tests/test_gpu_bench.py drives): the rank's slab
    p_local = p_total / N of the resident matrix, K divided by p_total (gwas/ols.rs:295), the all-reduce of the n x n sums
    (gwas/ols.rs:291-295 is the one reduction of the analysis; the slabs are the chunks of base/sync.rs:913-939).  A PROBE, not a
    measurement of N GPUs: the all-reduce of one rank moves no bytes over xGMI, so a real N-rank step adds the ring latency of a
    320 KB message to these figures."""
    from poolgen_amd.distributed import ols_with_covariate_sharded
    legs = {"what": "one rank's step at p_local = p_total / N with a ONE-rank RCCL communicator inside libpoolgen_hip (same calls as the "
                    "headline step); kernel times = HIP events on the library's stream; fixed_remainder_ms = ms_per_step minus the kernels "
                    "and the all-reduce (host n x n decision, launches, synchronisations); a probe of the rank-shaped step, NOT an N-GPU "
                    "measurement"}
    own_comm = False
    try:
        if not getattr(eng, "_comm_ready", False):
            with _StdoutToStderr():
                eng.comm_init(eng.comm_unique_id(), 1, 0)
            own_comm = True
        legs["comm_size"] = int(eng.comm_size)
        try:
            legs["rccl_version"] = int(eng.comm_version())
        except Exception:
            legs["rccl_version"] = None
        dev = G.device
        for N in (2, 4, 8):
            p_local = p_total // N
            if p_local < 1 or p_local > G.shape[0]:
                continue
            Gl = G[:p_local]
            out = torch.empty((3, p_local, 1), dtype=torch.float64, device=dev)
            for _ in range(2):
                ols_with_covariate_sharded(eng, Gl, p_total, Y, var_explained, -1, n, out)
            eng.profile_reset()
            torch.cuda.synchronize(dev)
            stamps = [time.perf_counter()]
            for _ in range(steps):
                m = ols_with_covariate_sharded(eng, Gl, p_total, Y, var_explained, -1, n, out)[0]
                stamps.append(time.perf_counter())
            torch.cuda.synchronize(dev)
            total = time.perf_counter() - stamps[0]
            per = np.diff(np.asarray(stamps)) * 1e3
            k_ms, k_n = eng.profile_get("kinship")
            r_ms, r_n = eng.profile_get("kinship_reduce")
            f_ms, f_n = eng.profile_get("sweep_finish")
            s_ms, s_n = eng.profile_get("sweep")
            a_ms, a_n = eng.profile_get("allreduce")
            ms = total / steps * 1e3
            kern = (k_ms + r_ms + f_ms + s_ms) / steps
            legs[f"n{N}"] = {"ranks_modelled": N, "p_local": p_local, "p_total": p_total, "n_eigenvecs": int(m), "steps": steps,
                             "ms_per_step": ms, "ms_per_step_median": float(np.median(per)),
                             "kinship_ms": k_ms / max(k_n, 1), "kinship_reduce_ms": r_ms / max(r_n, 1),
                             "sweep_finish_ms": f_ms / max(f_n, 1), "sweep_ms": s_ms / max(s_n, 1) if s_n else 0.0,
                             "allreduce_ms": a_ms / max(a_n, 1), "allreduce_launches": int(a_n),
                             "fixed_remainder_ms": ms - kern - a_ms / steps,
                             "loci_per_s_if_N_ranks_ran_this": p_total / (ms * 1e-3)}
            del out
    except Exception as e:
        legs["error"] = f"{type(e).__name__}: {e}"
    finally:
        if own_comm:
            try:
                eng.comm_destroy()
            except Exception:
                pass
    return legs


def end_to_end_leg(eng, G, Y, n, loci, var_explained, torch, np):
    """SURVEY 8(d) timing protocol, second half: wall clock of the host-buffer entry point pg_ols_kinship (what main.rs:285-291 hands
    over: the matrix in HOST memory) from PINNED host buffers -- H2D of G in slabs overlapped with the partial kinship, n x n step,
    sweep slab by slab overlapped with the D2H of the results.  Never the headline `value`."""
    import ctypes as C
    leg = {"what": "pg_ols_kinship from pinned host memory: wall clock incl. H2D of G, kinship, n x n step, sweep, D2H of beta/var/p; "
                   "gpu_compute_share = (kinship + sweep kernel time by HIP events) / wall; PCIe-inclusive, never the headline value"}
    try:
        p = int(min(loci, G.shape[0]))
        ld = int(G.shape[1])
        dev = G.device
        t0 = time.perf_counter()
        pinned = torch.empty((p, ld), dtype=torch.float64, pin_memory=True)
        outs = [torch.empty((p, 1), dtype=torch.float64, pin_memory=True) for _ in range(3)]
        leg["pin_seconds"] = time.perf_counter() - t0
        pinned.copy_(G[:p]); torch.cuda.synchronize(dev)
        # the link alone: one H2D of the same buffer
        scratch = torch.empty((p, ld), dtype=torch.float64, device=dev)
        scratch.copy_(pinned, non_blocking=True); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        scratch.copy_(pinned, non_blocking=True); torch.cuda.synchronize(dev)
        link = time.perf_counter() - t0
        del scratch
        Yh = np.ascontiguousarray(Y, dtype=np.float64).reshape(n, -1)
        m = C.c_int()
        lib, ctx = eng._lib, eng._ctx
        walls = []
        for rep in range(3):
            eng.profile_reset()
            t0 = time.perf_counter()
            rc = lib.pg_ols_kinship(ctx, pinned.data_ptr(), p, n, ld, Yh.ctypes.data, 1, float(var_explained), -1, C.byref(m), None,
                                    outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr())
            walls.append(time.perf_counter() - t0)
            if rc != 0:
                raise RuntimeError(lib.pg_last_error(ctx).decode())
        k_ms, k_n = eng.profile_get("kinship")
        s_ms, s_n = eng.profile_get("sweep")
        wall = float(np.median(walls))
        gbytes = 8.0 * ld * p
        leg.update({"pools": n, "loci": p, "host_bytes": gbytes, "wall_s": wall, "wall_s_all": walls, "loci_per_s": p / wall,
                    "h2d_gbs_equiv": gbytes / wall / 1e9, "h2d_gbs_link_alone": gbytes / link / 1e9,
                    "kinship_kernel_ms_total": k_ms, "kinship_launches": int(k_n), "sweep_kernel_ms_total": s_ms,
                    "sweep_launches": int(s_n), "gpu_compute_share": (k_ms + s_ms) * 1e-3 / walls[-1], "n_eigenvecs": int(m.value)})
        del pinned, outs
    except Exception as e:
        leg["error"] = f"{type(e).__name__}: {e}"
    return leg


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pools", type=int, default=200)
    ap.add_argument("--loci", type=int, default=10_000_000, help="TOTAL loci over all GPUs")
    ap.add_argument("--var-explained", type=float, default=0.75)
    ap.add_argument("--force-m", type=int, default=-1, help=">=0 forces the number of kinship PCs")
    ap.add_argument("--ld", type=int, default=0, help="leading dimension of G in doubles (0 = default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep-legs", action="store_true", help="skip the two untimed legs that measure k_ols_sweep_mfma")
    ap.add_argument("--sweep-steps", type=int, default=5, help="steps per sweep leg (after the timed region)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the untimed BASELINE configs[1] / configs[3] legs (count operators 100 x 1M, ridge path 500 x 5M)")
    ap.add_argument("--secondary-loci", type=int, default=1_000_000, help="loci of the count-operator leg (configs[1]: 1M)")
    ap.add_argument("--ridge-loci", type=int, default=5_000_000, help="loci of the ridge leg (configs[3]: 5M)")
    ap.add_argument("--no-realistic", action="store_true", help="skip the count-operator legs on error-bearing counts")
    ap.add_argument("--no-lazy", action="store_true", help="skip the labelled lazy-kinship leg")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000)
    ap.add_argument("--no-shard-probe", action="store_true",
                    help="skip the untimed rank-shaped legs (p_local = p/2, p/4, p/8 with a one-rank RCCL communicator)")
    ap.add_argument("--probe-steps", type=int, default=10, help="steps per shard-probe leg")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the untimed PCIe-inclusive leg (pg_ols_kinship from pinned host memory)")
    ap.add_argument("--e2e-loci", type=int, default=4_000_000, help="loci of the end-to-end leg (6.4 GB of pinned host memory at 200 pools)")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="seconds the self-launching parent of --gpus N > 1 waits for its rank processes before it kills them")
    return ap.parse_args(argv)


class _StdoutToStderr:
    """RCCL prints a version banner to the C-level stdout when a communicator is initialised; stdout is reserved for the ONE JSON
    line.  While active, file descriptor 1 points at stderr, and libc's buffer is flushed before it is restored."""
    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self._libc = ctypes.CDLL(None)
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        try:
            self._libc.fflush(None)
        finally:
            os.dup2(self._saved, 1)
            os.close(self._saved)
        return False


def _free_port() -> int:
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _kill_group(proc):
    """End one rank process we started (and whatever it started): its own process group, by id -- never by pattern."""
    if proc.poll() is not None:
        return
    for sig in (signal.SIGTERM, signal.SIGKILL):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            proc.wait(timeout=10)
            return
        except subprocess.TimeoutExpired:
            continue


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): this process becomes the
    launcher.  It starts N rank processes of this same script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per GPU,
    each in its own process group), waits for them under --launch-timeout, relays rank 0's ONE JSON line and exits non-zero
    if any rank failed or hung (every rank is then killed by pid).  It never imports torch, never creates an Engine and never
    re-execs itself: nothing here has initialised a GPU.  The reference's counterpart is the thread-per-chunk fan-out of
    base/sync.rs:913-939 with the one reduction of gwas/ols.rs:291-295 inside the ranks."""
    n = args.gpus
    port = _free_port()
    env0 = dict(os.environ)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool's driver (RCCL needs it across processes)
    env0.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                POOLGEN_BENCH_LAUNCHED="1")
    procs, logs = [], []
    tmp = tempfile.mkdtemp(prefix="poolgen_bench_")
    try:
        for r in range(n):
            env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
            out = open(os.path.join(tmp, f"rank{r}.out"), "w+")
            err = open(os.path.join(tmp, f"rank{r}.err"), "w+")
            logs.append((out, err))
            procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + list(argv), env=env, stdout=out,
                                          stderr=err, cwd=str(ROOT), start_new_session=True))
        deadline = time.monotonic() + args.launch_timeout
        failed = None
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                alive = [r for r, c in enumerate(codes) if c is None]
                failed = f"ranks {alive} still running after --launch-timeout {args.launch_timeout:.0f} s (hung?)"
                break
            time.sleep(0.2)
        if failed:
            # a rank that died leaves the others inside a collective: give them a moment to notice, then end them
            t_end = time.monotonic() + 5.0
            while time.monotonic() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.2)
            for p in procs:
                _kill_group(p)
        def tail(f, nbytes=3000):
            f.flush(); f.seek(0); s = f.read(); return s[-nbytes:]
        if failed:
            print(f"bench.py --gpus {n}: {failed}; all ranks ended", file=sys.stderr)
            for r, (out, err) in enumerate(logs):
                t = tail(err).strip()
                if t:
                    print(f"---- rank {r} stderr (tail) ----\n{t}", file=sys.stderr)
            return 1
        for r, (out, err) in enumerate(logs):   # warnings of the ranks stay visible
            t = tail(err, 1500).strip()
            if t:
                print(f"---- rank {r} stderr (tail) ----\n{t}", file=sys.stderr)
        lines = [l for l in tail(logs[0][0], 1 << 20).splitlines() if l.startswith("{")]
        if len(lines) != 1:
            print(f"bench.py --gpus {n}: rank 0 printed {len(lines)} JSON lines, expected 1", file=sys.stderr)
            return 1
        print(lines[0])
        return 0
    finally:
        for p in procs:
            _kill_group(p)
        for out, err in logs:
            out.close(); err.close()
        try:
            for f in os.listdir(tmp):
                os.unlink(os.path.join(tmp, f))
            os.rmdir(tmp)
        except OSError:
            pass


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    worker(args)


def worker(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("POOLGEN_BENCH_TEST_HANG") in (str(rank), "all"):   # tests/test_bench_launcher.py: a rank that never comes back
        time.sleep(3600)
    if world != args.gpus:
        # a launcher that disagrees with --gpus is a mistake in the command, not something to paper over: the line would
        # carry an n_gpus the driver did not ask for
        if rank == 0:
            print(f"error: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    ndev = torch.cuda.device_count()          # does not initialise the GPU
    rehearsal = os.environ.get("POOLGEN_BENCH_BACKEND", "nccl") != "nccl"
    if world > ndev and not rehearsal:
        if rank == 0:
            print(f"error: --gpus {world} but only {ndev} GPU(s) are visible (POOLGEN_BENCH_BACKEND=gloo rehearses the control "
                  f"flow with several ranks on one GPU; it is never a measurement)", file=sys.stderr)
        sys.exit(2)
    dev_index = local_rank % max(ndev, 1)   # one rank per GPU; wraps only in the 1-GPU rehearsal
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # POOLGEN_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, library communicator, all-reduce, barriers) with
    # ONE rank too -- what a 1-GPU box can run of it with the real nccl (= RCCL) backend (tests/test_gpu_bench.py)
    force_dist = os.environ.get("POOLGEN_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        # nccl == RCCL on ROCm.  POOLGEN_BENCH_BACKEND=gloo exists only to rehearse the multi-rank
        # control flow on a single-GPU box (several ranks sharing cuda:0), never for measurements.
        backend = os.environ.get("POOLGEN_BENCH_BACKEND", "nccl")
        with _StdoutToStderr():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)

    from poolgen_amd import Engine, synth
    from poolgen_amd.distributed import ols_with_covariate_sharded, setup_comm, shard_range

    n, p_total, k = args.pools, args.loci, 1
    lo, hi = shard_range(p_total, rank, world)
    p_local = hi - lo
    eng = Engine(dev_index)
    # the kinship all-reduce runs inside libpoolgen_hip on its own RCCL communicator (pg_comm_*); torch.distributed only
    # carries the 128-byte unique id and the timing barriers.  Falls back to torch.distributed's all-reduce only if the
    # library cannot set its communicator up (reported in the JSON line as "allreduce").
    allreduce_impl = "none (1 rank)"
    if use_dist:
        # setup_comm answers the same on every rank (its stages end in agreements over torch.distributed), so either every
        # rank all-reduces inside the library or every rank uses dist.all_reduce -- never a mixture
        with _StdoutToStderr():
            comm_ok = setup_comm(eng, force=force_dist)
        if comm_ok:
            allreduce_impl = "RCCL inside libpoolgen_hip (pg_allreduce_sum_dev)"
        else:
            allreduce_impl = f"FALLBACK torch.distributed all_reduce ({os.environ.get('POOLGEN_BENCH_BACKEND', 'nccl')}): " \
                             f"the library communicator was not set up on every rank"
            if rank == 0:
                print("warning: " + allreduce_impl, file=sys.stderr)
    comm_size = int(eng.comm_size)          # pg_comm_size: the ranks the LIBRARY's communicator spans (1 without one)
    try:
        rccl_version = int(eng.comm_version()) if use_dist else None   # ncclGetVersion of the RCCL the library loaded
    except Exception:
        rccl_version = None
    G = synth.genotype_matrix(p_local, n, dev, start=lo, ld=(args.ld or None))
    # phenotype: 10 causal loci spread over the WHOLE matrix; any rank can regenerate any locus
    causal = [(p_total * (2 * i + 1)) // 20 for i in range(10)]
    Gc = torch.cat([synth.genotype_matrix(1, n, dev, start=c) for c in causal], dim=0)
    rng = np.random.default_rng(synth.SEED)
    gval = Gc[:, :n].T.cpu().numpy() @ rng.normal(size=10)
    Y = (gval + rng.normal(size=n) * np.sqrt(gval.var())).reshape(n, 1)   # h2 = 0.5
    out = torch.empty((3, p_local, k), dtype=torch.float64, device=dev)

    def step():
        return ols_with_covariate_sharded(eng, G, p_total, Y, args.var_explained, args.force_m, n, out)

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    m = -1
    for _ in range(args.warmup):
        m = step()[0]
    eng.profile(True)
    eng.profile_reset()
    fence()
    t0 = time.perf_counter()
    stamps = [t0]
    for _ in range(args.steps):
        m = step()[0]
        stamps.append(time.perf_counter())   # a step ends in the host's n x n decision (a stream synchronisation), so the stamps need none
    fence()
    dt = time.perf_counter() - t0
    step_ms = [(b - a) * 1e3 for a, b in zip(stamps[:-1], stamps[1:])]
    step_ms[-1] += (t0 + dt - stamps[-1]) * 1e3   # the tail of the last step's kernels, drained by the fence
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kin_ms, kin_n = eng.profile_get("kinship")
    red_ms, red_n = eng.profile_get("kinship_reduce")
    sw_ms, sw_n = eng.profile_get("sweep")
    fin_ms, fin_n = eng.profile_get("sweep_finish")

    # ---- after the timed region: the general path, so that the per-locus sweep (north_star's HBM-roofline kernel) has a
    # driver-run number too.  At the default -x 0.75 the eigen rule gives m = 0 and the fits are closed from sums fused into
    # the kinship pass: the sweep kernel (k_ols_sweep_mfma) is never launched by the headline.  Two extra legs of `extra` steps each:
    #   two_pass : same analysis with the fusion off -> kinship (no fused sums) + k_ols_sweep_mfma with [1 | g]   (m = 0)
    #   m8       : --force-m 8 -> host eigenvectors + k_ols_sweep_mfma with [1 | C(8) | g]                      (m = 8)
    legs = {}
    extra = 0 if args.no_sweep_legs else args.sweep_steps
    if extra > 0 and args.force_m < 0:
        for tag, fm, env in (("two_pass", -1, "1"), ("m8", 8, None)):
            old = os.environ.get("POOLGEN_TWO_PASS")
            if env:
                os.environ["POOLGEN_TWO_PASS"] = env
            try:
                def leg_step():
                    return ols_with_covariate_sharded(eng, G, p_total, Y, args.var_explained, fm, n, out)
                leg_step(); leg_step()
                eng.profile_reset()
                fence()
                t1 = time.perf_counter()
                for _ in range(extra):
                    lm = leg_step()[0]
                fence()
                ldt = time.perf_counter() - t1
            finally:
                if env:
                    if old is None:
                        del os.environ["POOLGEN_TWO_PASS"]
                    else:
                        os.environ["POOLGEN_TWO_PASS"] = old
            if use_dist:
                tt = torch.tensor([ldt], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ldt = float(tt.item())
            lk_ms, lk_n = eng.profile_get("kinship")
            ls_ms, ls_n = eng.profile_get("sweep")
            legs[tag] = dict(m=int(lm), ms_per_step=ldt / extra * 1e3, kin_avg=lk_ms / max(lk_n, 1),
                             sw_avg=ls_ms / max(ls_n, 1), sw_n=int(ls_n))
    # ---- the lazy-kinship route (labelled, NEVER the headline): K_out = NULL lets the library decide m = 0 from a bound that needs no K ----
    lazy = None
    if world == 1 and extra > 0 and args.force_m < 0 and not args.no_lazy:
        try:
            eng.ols_with_covariate(G, Y, args.var_explained, n=n, out=out, want_K=False)
            eng.profile_reset()
            fence()
            t1 = time.perf_counter()
            for _ in range(extra):
                lm = eng.ols_with_covariate(G, Y, args.var_explained, n=n, out=out, want_K=False)[0]
            fence()
            ldt = (time.perf_counter() - t1) / extra
            lk_ms, lk_n = eng.profile_get("kinship")
            ls_ms, ls_n = eng.profile_get("sweep")
            sb = (8.0 * n + 24.0 * k) * p_local
            lazy = {"what": "NOT the headline: pg_ols_kinship_dev with K_out = NULL.  The n_eigenvecs rule (gwas/ols.rs:297-311) gives m = 0 "
                            "as soon as lambda_1 / trace(K) >= x, and lambda_1 >= 1'K1 / n = sum_l (sum_i g_li)^2 / n; the intercept-only sweep "
                            "forms that bound on the side, K is never built, one HBM-bound pass.  Outputs bit-identical to the two-pass route "
                            "(tests/test_gpu_kinship_path.py::test_lazy_kinship_route); the headline value and roofline keep forming K on "
                            "the matrix cores as north_star specifies",
                    "n_eigenvecs": int(lm), "ms_per_step": ldt * 1e3, "loci_per_s": p_total / ldt, "kinship_launches": int(lk_n),
                    "sweep_avg_ms": ls_ms / max(ls_n, 1), "sweep_launches": int(ls_n), "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "achieved": sb / (ls_ms / max(ls_n, 1) * 1e-3) / 1e9 if ls_n else None,
                    "frac": sb / (ls_ms / max(ls_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS if ls_n else None}
        except Exception as e:
            lazy = {"error": f"{type(e).__name__}: {e}"}
    # ---- rank-shaped probe and the PCIe-inclusive leg: after the timed region, 1 GPU only, bounded ----
    shard_probe = None
    if world == 1 and not args.no_shard_probe and args.force_m < 0:
        shard_probe = shard_probe_legs(eng, G, Y, n, p_total, args.var_explained, args.probe_steps, torch, np)
    end_to_end = None
    if world == 1 and not args.no_end_to_end and args.force_m < 0:
        end_to_end = end_to_end_leg(eng, G, Y, n, args.e2e_loci, args.var_explained, torch, np)
    # ---- BASELINE configs[1] and configs[3], driver-visible: after the timed region, bounded, 1 GPU only ----
    secondary = None
    if world == 1 and not args.no_secondary and args.force_m < 0:
        secondary = secondary_legs(eng, dev, args, torch, np)
    eng.profile(False)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = p_total / (dt / args.steps)
        kin_avg = kin_ms / max(kin_n, 1)
        sw_avg = sw_ms / max(sw_n, 1)
        sweep_bytes = (8.0 * n + 24.0 * k) * p_local       # 8n read + 24k written per locus
        kin_flops = 2.0 * n * n * p_local                   # reference computes the full product
        kin_tflops = kin_flops / (kin_avg * 1e-3) / 1e12 if kin_avg > 0 else 0.0
        traffic_db = {}
        traffic_state = "none"
        tfile = ROOT / "profiles" / "pmc_traffic.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                if tj.get("workload") == f"{n}x{p_local}":
                    traffic_db = tj
                    import hashlib
                    h = hashlib.sha256()
                    csrc = ROOT / "poolgen_amd" / "csrc"
                    for f in sorted(list(csrc.glob("*.hip")) + list(csrc.glob("*.h"))):
                        h.update(f.read_bytes())
                    traffic_state = "measured on these kernel sources" if tj.get("kernel_sources_sha256") == h.hexdigest() else \
                        "STALE: the kernel sources have changed since the counter passes (tools/r04_profile.sh re-measures)"
            except Exception:
                traffic_db = {}
        traffic_note = ("HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes of this command "
                        "(counters cannot be read from inside the process); file profiles/pmc_traffic.json, see its 'source'; " + traffic_state)
        tiles = (n + 15) // 16
        # MFMA work the kernel EXECUTES (upper triangle, padding and the redundant parts of the diagonal tiles included);
        # USEFUL = the n (n + 1) / 2 distinct products of the triangle
        if tiles == 13:
            # the 13-tile kernel: 66 (n <= 200) or 78 full off-diagonal 16x16x4 tiles + 13 diagonal tiles as 3 instructions of four
            # 4x4x4 blocks + (n <= 200) 12 last-column tiles as 2 such instructions; 512 / 128 flop per locus per 16x16 tile / 4-block instruction
            kin_exec_flops = ((66 if n <= 200 else 78) * 512.0 + 13 * 3 * 128.0 + (12 * 2 * 128.0 if n <= 200 else 0.0)) * p_local
        else:
            kin_exec_flops = (tiles * (tiles + 1) // 2 if tiles <= 13 else tiles * tiles) * 512.0 * p_local
        kin_useful_flops = float(n) * (n + 1) * p_local
        per_s = lambda fl, ms: fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        roof = {"kernel": "k_kinship_syrk", "bound": "mfma",
                "achieved": per_s(kin_exec_flops, kin_avg), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": per_s(kin_exec_flops, kin_avg) / FP64_MFMA_PEAK_TFLOPS,
                "what": "EXECUTED fp64 MFMA flop/s (matrix-pipe utilisation); the peak is AMD's datasheet 78.6 TFLOP/s, confirmed at "
                        "77.4 (4 waves/SIMD) by tools/microbench.hip on the box (profiles/r02_microbench.log)",
                "useful_tflops": per_s(kin_useful_flops, kin_avg),
                "useful_frac": per_s(kin_useful_flops, kin_avg) / FP64_MFMA_PEAK_TFLOPS,
                "algorithmic_tflops": kin_tflops,
                "algorithmic_note": "2 n^2 p / time (SURVEY 8d: the reference forms the full product); exceeds the peak because only "
                                    "one triangle is computed -- not a utilisation",
                "traffic": traffic_db.get("kinship_hbm_bytes_per_launch"), "traffic_note": traffic_note,
                "traffic_stale": traffic_state.startswith("STALE"),
                "avg_ms": kin_avg, "launches": kin_n}
        rec = {
            "metric": "loci/sec ols_iter_with_kinship, 200 pools x 10M loci",
            "value": value, "unit": "loci/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_step_median": float(np.median(step_ms)),
            "ms_per_step_min": float(np.min(step_ms)), "ms_per_step_max": float(np.max(step_ms)),
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"ols_iter_with_kinship {n} pools x {p_total} loci (BASELINE configs[2])",
                       "pools": n, "loci_total": p_total, "loci_per_gpu": p_local, "traits": k,
                       "xxt_eigen_variance_explained": args.var_explained, "n_eigenvecs": m,
                       "parallelism": f"locus-sharded x{world}, 1 all-reduce of {n}x{n} fp64", "allreduce": allreduce_impl,
                       "comm_size": comm_size, "rccl_version": rccl_version,
                       "launcher": ("bench.py (self-launched ranks)" if os.environ.get("POOLGEN_BENCH_LAUNCHED") == "1" else
                                    "external (torch.distributed.run)" if "WORLD_SIZE" in os.environ else "none (single process)"),
                       "rehearsal": bool(rehearsal and world > 1)},
            "roofline": roof,
            "kernels": {
                "k_kinship_syrk": {"avg_ms": kin_avg, "executed_mfma_tflops": per_s(kin_exec_flops, kin_avg),
                                   "useful_tflops": per_s(kin_useful_flops, kin_avg), "algorithmic_tflops": kin_tflops},
                "k_kinship_reduce": {"avg_ms": red_ms / max(red_n, 1)},
                "k_ols_sweep_mfma": ({"avg_ms": sw_avg, "gbs_algorithmic": sweep_bytes / (sw_avg * 1e-3) / 1e9,
                                 "frac_of_hbm_peak": sweep_bytes / (sw_avg * 1e-3) / 1e9 / HBM_PEAK_GBS} if sw_n else
                                {"avg_ms": 0.0, "note": "not launched by the headline: m = 0 fits closed from the sums fused into the "
                                                        "kinship pass; see roofline_sweep for the general path"}),
                "k_sweep_finish": {"avg_ms": fin_ms / max(fin_n, 1), "launches": fin_n},
                "host_eig_and_glue_ms": ms_per_step - kin_avg - sw_avg - red_ms / max(red_n, 1) - fin_ms / max(fin_n, 1),
            },
        }
        if legs:
            rs = {"kernel": "k_ols_sweep_mfma", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "bytes_per_locus": 8.0 * n + 24.0 * k, "loci_per_launch": p_local, "steps_per_leg": extra,
                  "what": "algorithmic bytes (SURVEY 8d: 8n read + 24k written per locus) / HIP-event launch time, measured after "
                          "the timed headline region in the same process", "traffic_note": traffic_note}
            for tag, L in legs.items():
                gbs = sweep_bytes / (L["sw_avg"] * 1e-3) / 1e9 if L["sw_avg"] > 0 else 0.0
                rs[tag] = {"n_eigenvecs": L["m"], "avg_ms": L["sw_avg"], "launches": L["sw_n"], "achieved": gbs,
                           "frac": gbs / HBM_PEAK_GBS, "ms_per_step": L["ms_per_step"],
                           "loci_per_s": p_total / (L["ms_per_step"] * 1e-3), "kinship_avg_ms": L["kin_avg"],
                           "traffic": traffic_db.get(f"sweep_{tag}_hbm_bytes_per_launch")}
            rs["achieved"] = rs["two_pass"]["achieved"]; rs["frac"] = rs["two_pass"]["frac"]
            rs["traffic"] = rs["two_pass"]["traffic"]
            rec["roofline_sweep"] = rs
        if shard_probe:
            rec["shard_probe"] = shard_probe
        if end_to_end:
            rec["end_to_end"] = end_to_end
        if secondary:
            co = secondary.get("count_operators", {})
            for op, key in (("ols_iter", "ols_iter_stream_hbm_bytes_per_launch"), ("pearson_corr", "pearson_stream_hbm_bytes_per_launch"),
                            ("chisq_test", "chisq_stream_hbm_bytes_per_launch")):
                if op in co and int(args.secondary_loci) == 1_000_000:
                    co[op]["traffic"] = traffic_db.get(key)
            rec["secondary"] = secondary
        if lazy:
            rec.setdefault("secondary", {"what": "untimed legs after the headline region"})["lazy_kinship"] = lazy
        if world == 1 and not args.no_cpu_baseline:
            s = min(args.cpu_sample, p_local)
            rec["cpu_baseline"] = cpu_baseline(G[:s, :n].cpu().numpy(), Y, args.var_explained, args.force_m)
        print(json.dumps(rec))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
