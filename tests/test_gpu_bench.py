"""bench.py contract checks on the GPU box: the JSON line has the fields the driver reads, the
N = 1 path works under torch.distributed.run, and the multi-rank control flow (shard, all-reduce,
max-over-ranks timing, rank-0 print) runs -- rehearsed with two ranks sharing cuda:0 over gloo,
because a 1-GPU box cannot host two RCCL ranks."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
NEED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def last_json(out: str):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--loci", "1000000",
                        "--cpu-sample", "20000", "--e2e-loci", "400000", "--probe-steps", "3", "--secondary-loci", "100000",
                        "--ridge-loci", "100000"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert len([l for l in r.stdout.splitlines() if l.strip()]) == 1, "stdout carries the ONE JSON line and nothing else (RCCL's banner goes to stderr)"
    assert NEED <= set(d) and "cpu_baseline" in d
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_max"]
    # the rank-shaped probe: a ONE-rank RCCL communicator inside the library, slabs of p/2, p/4, p/8 with K over p_total
    sp = d["shard_probe"]
    assert "error" not in sp, sp
    assert sp["comm_size"] == 1 and sp["rccl_version"] >= 20000
    for N in (2, 4, 8):
        leg = sp[f"n{N}"]
        assert leg["p_local"] == 1000000 // N and leg["p_total"] == 1000000 and leg["allreduce_launches"] == 3
        assert leg["kinship_ms"] > 0 and leg["allreduce_ms"] > 0 and leg["ms_per_step"] > leg["kinship_ms"]
    # the PCIe-inclusive leg: pg_ols_kinship from pinned host memory
    ee = d["end_to_end"]
    assert "error" not in ee, ee
    assert ee["loci"] == 400000 and ee["wall_s"] > 0 and 0 < ee["gpu_compute_share"] < 1 and ee["h2d_gbs_equiv"] > 1
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "loci/s" and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert d["value"] > 50 * cb["value"]


def test_two_rank_control_flow_rehearsal():
    env = dict(os.environ, POOLGEN_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(ROOT / "bench.py"), "--gpus", "2",
                        "--steps", "2", "--warmup", "1", "--loci", "1000001", "--no-cpu-baseline"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["loci_total"] == 1000001 and d["config"]["loci_per_gpu"] == 500000
    assert d["scaling"] == "strong" and d["value"] > 0
    # two ranks on ONE device: RCCL refuses (on every rank: setup_comm answers the same everywhere), the line says which
    # all-reduce ran instead
    assert d["config"]["allreduce"].startswith("FALLBACK torch.distributed"), d["config"]
    assert d["config"]["launcher"].startswith("external") and d["config"]["comm_size"] == 1


def _no_launcher_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_self_launched_two_ranks_rehearsal():
    """`python bench.py --gpus 2` exactly as the driver starts it -- no launcher, no WORLD_SIZE: bench.py starts its own two rank
    processes (the parent never touches the GPU), and prints ONE line with n_gpus = 2.  On this 1-GPU box the two ranks share
    cuda:0 over gloo (POOLGEN_BENCH_BACKEND=gloo, the rehearsal switch; the line says so)."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--loci", "1000001",
                        "--no-cpu-baseline", "--sweep-steps", "1"], capture_output=True, text=True, cwd=ROOT,
                       env=_no_launcher_env(POOLGEN_BENCH_BACKEND="gloo"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["loci_per_gpu"] == 500000 and d["value"] > 0
    assert d["config"]["launcher"].startswith("bench.py") and d["config"]["rehearsal"] is True
    assert d["config"]["allreduce"].startswith("FALLBACK torch.distributed") and d["config"]["comm_size"] == 1
    assert "secondary" not in d            # configs[1] / configs[3] are single-GPU legs


def test_self_launch_refuses_more_ranks_than_gpus():
    # without the rehearsal switch two ranks on one GPU are refused by every rank, and the parent reports it: non-zero, no line
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--loci", "100000",
                        "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, env=_no_launcher_env(), timeout=600)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_secondary_legs_in_the_line():
    """BASELINE configs[1] and configs[3] are part of the driver-run line (reduced sizes here; the default run uses 1M / 5M)."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--loci", "500000", "--no-cpu-baseline",
                        "--sweep-steps", "1", "--secondary-loci", "200000", "--ridge-loci", "200000"], capture_output=True, text=True,
                       cwd=ROOT, env=_no_launcher_env(), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    sec = d["secondary"]
    co = sec["count_operators"]
    for op in ("ols_iter", "pearson_corr", "chisq_test"):
        assert co[op]["kernel_ms"] > 0 and co[op]["launches"] == 10 and 0 < co[op]["frac"] < 1
        assert co[op]["bytes_per_launch"] == 24.0 * 100 * 200000
    cr = sec["count_operators_realistic"]
    for tag in ("error_0.005_maf_0.01", "error_0.001_default_filter", "error_0.005_default_filter"):
        for op in ("ols_iter", "pearson_corr", "chisq_test"):
            leg = cr[tag][op]
            assert leg["kernel_ms"] > 0 and leg["launches"] == 10 and 0 < leg["frac"] < 1 and 0 <= leg["deferred_fraction"] <= 1
    assert cr["error_0.005_maf_0.01"]["ols_iter"]["deferred_fraction"] < 0.01           # the order-free kernel knows the survivors before it sums
    assert 0 < cr["error_0.005_maf_0.01"]["pearson_corr"]["deferred_fraction"] < 0.15   # the streaming pass: its speculated pair holds for most loci
    assert cr["error_0.005_default_filter"]["ols_iter"]["deferred_fraction"] > 0.5     # error alleles that survive: multi-allelic fits
    assert cr["error_0.005_default_filter"]["chisq_test"]["deferred_fraction"] == 0.0   # chisq_test closes any number of survivors in place
    assert cr["ols_iter"] == cr["error_0.005_maf_0.01"]["ols_iter"]
    rd = sec["ridge"]
    assert "error" not in rd, rd
    # 10 repetitions x 10 folds + the all-rows fit = 101 coefficient columns, 16 per pass over G (round 4); one prediction pass per repetition
    assert rd["coefficient_pass"]["launches"] == 7 and rd["prediction_pass"]["launches"] == 10 and rd["wall_s"] > 0
    assert rd["passes_over_G"] == 18
    assert 0 < rd["coefficient_pass"]["frac"] < 1 and 0 < rd["prediction_pass"]["frac"] < 1


def test_one_rank_through_the_real_rccl_path():
    """The multi-rank code path of bench.py with the REAL backends, as far as one GPU allows: torch.distributed nccl (= RCCL)
    process group of one rank, the library's own communicator set up through it (pg_comm_init_rank with the id carried by
    torch.distributed), the kinship all-reduce inside libpoolgen_hip, barriers and max-over-ranks timing."""
    env = dict(os.environ, POOLGEN_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--loci", "2000000",
                        "--no-cpu-baseline", "--sweep-steps", "1"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["config"]["allreduce"].startswith("RCCL inside libpoolgen_hip"), d["config"]
    assert d["config"]["comm_size"] == 1 and d["config"]["rccl_version"] >= 20000
    assert d["value"] > 0 and "roofline_sweep" in d and d["roofline_sweep"]["two_pass"]["launches"] == 1
