"""bench.py as its own launcher (`python bench.py --gpus N` with no WORLD_SIZE in the environment), on CPU: the parent starts
N rank processes, and a rank that fails or hangs ends the whole run with a non-zero exit and no process left behind -- it never
hangs and never prints a line with the wrong n_gpus.  (The ranks themselves need GPUs; here they fail or are told to hang, which
is exactly what the launcher must survive.  The working N = 2 path is rehearsed on the GPU box in tests/test_gpu_bench.py.)
Also: the go / abort agreement of the CLI's rank threads (host/rank_gate.h), through hostcheck."""
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HC = ROOT / "poolgen_amd" / "csrc" / "hostcheck"


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_parent_reports_a_failed_rank():
    # no GPU here: every rank exits with an error of its own; the parent must relay that, not a JSON line
    t0 = time.time()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--loci", "1000",
                        "--no-cpu-baseline", "--launch-timeout", "120"], capture_output=True, text=True, cwd=ROOT,
                       env=_clean_env(), timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "all ranks ended" in r.stderr and "exited with code" in r.stderr
    assert time.time() - t0 < 200


def test_parent_kills_a_hung_rank():
    t0 = time.time()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--loci", "1000",
                        "--no-cpu-baseline", "--launch-timeout", "8"], capture_output=True, text=True, cwd=ROOT,
                       env=_clean_env(POOLGEN_BENCH_TEST_HANG="all"), timeout=300)
    assert r.returncode != 0
    assert "all ranks ended" in r.stderr and "still running after --launch-timeout" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 120
    # nothing of ours is left running: the hung rank slept in `time.sleep(3600)` under a marker only this test sets
    ps = subprocess.run(["ps", "-eo", "pid,args"], capture_output=True, text=True).stdout
    mine = [l for l in ps.splitlines() if "bench.py" in l and "--launch-timeout 8" in l]
    assert not mine, mine


def test_launcher_is_not_used_under_an_external_launcher():
    # WORLD_SIZE set (torch.distributed.run did the launching): bench.py must NOT start ranks of its own, and a WORLD_SIZE
    # that contradicts --gpus is an error of the command
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--loci", "1000"], capture_output=True,
                       text=True, cwd=ROOT, env=_clean_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr


def test_rank_gate_go_or_abort_together():
    subprocess.check_call(["make", "-C", str(HC.parent), "hostcheck", "-s"])
    def gate(*a):
        return subprocess.run([str(HC), "gate", *map(str, a)], capture_output=True, text=True, timeout=60).stdout.strip()
    assert gate(4, -1, 3) == "entered 4 own 0 aborted 0 finished 4"
    # a rank that fails before the collective: NO rank enters it, the failing rank keeps its own error, the others stop
    assert gate(4, 2, 3) == "entered 0 own 1 aborted 3 finished 0"
    assert gate(8, 0, 1) == "entered 0 own 1 aborted 7 finished 0"
    assert gate(2, 1, 50) == "entered 0 own 1 aborted 1 finished 0"
