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
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD MI355X datasheet, fp64 matrix (not in the local guide)


def cpu_baseline(G_sample_host: np.ndarray, Y: np.ndarray, var_explained: float, force_m: int):
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


def main():
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
    ap.add_argument("--cpu-sample", type=int, default=4_000_000)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)   # one rank per GPU; wraps only in the 1-GPU rehearsal below
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
        try:
            allreduce_impl = "RCCL inside libpoolgen_hip (pg_allreduce_sum_dev)" if setup_comm(eng, force=force_dist) else \
                f"torch.distributed ({os.environ.get('POOLGEN_BENCH_BACKEND', 'nccl')})"
        except Exception as e:  # keep the run alive: the scaling numbers are worth more than the purity of the path
            print(f"warning: library communicator unavailable ({e}); using torch.distributed all_reduce", file=sys.stderr)
            allreduce_impl = "torch.distributed (library communicator failed: %s)" % type(e).__name__
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
    for _ in range(args.steps):
        m = step()[0]
    fence()
    dt = time.perf_counter() - t0
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
        tfile = ROOT / "profiles" / "pmc_traffic.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                if tj.get("workload") == f"{n}x{p_local}":
                    traffic_db = tj
            except Exception:
                traffic_db = {}
        traffic_note = ("HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes of this command "
                        "(counters cannot be read from inside the process); file profiles/pmc_traffic.json, see its 'source'")
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
                "avg_ms": kin_avg, "launches": kin_n}
        rec = {
            "metric": "loci/sec ols_iter_with_kinship, 200 pools x 10M loci",
            "value": value, "unit": "loci/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"ols_iter_with_kinship {n} pools x {p_total} loci (BASELINE configs[2])",
                       "pools": n, "loci_total": p_total, "loci_per_gpu": p_local, "traits": k,
                       "xxt_eigen_variance_explained": args.var_explained, "n_eigenvecs": m,
                       "parallelism": f"locus-sharded x{world}, 1 all-reduce of {n}x{n} fp64", "allreduce": allreduce_impl},
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
        if world == 1 and not args.no_cpu_baseline:
            s = min(args.cpu_sample, p_local)
            rec["cpu_baseline"] = cpu_baseline(G[:s, :n].cpu().numpy(), Y, args.var_explained, args.force_m)
        print(json.dumps(rec))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
