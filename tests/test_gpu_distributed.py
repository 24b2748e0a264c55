"""The locus-sharded path with the REAL engine: two ranks (gloo, both on cuda:0 -- RCCL refuses two ranks on one
device, and a one-GPU box is what the tests get) run poolgen_amd.distributed on their slabs; the all-reduced kinship,
m and the concatenated fits must equal the single-process result of the same engine."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _data(force_m):
    from poolgen_amd import synth
    n, p = 200, 40_000 + 37
    G = synth.genotype_matrix(p, n, "cuda", seed=5)
    Y = synth.phenotypes(G[:4096], n, k=2)
    return G, Y, n, p


def _worker(rank, world, port, force_m, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from poolgen_amd import Engine
    from poolgen_amd.distributed import ols_with_covariate_sharded, shard_range
    G, Y, n, p = _data(force_m)
    lo, hi = shard_range(p, rank, world)
    eng = Engine(0)
    m, K, beta, var, pval = ols_with_covariate_sharded(eng, G[lo:hi].contiguous(), p, Y, force_m=force_m, n=n, want_K=True)
    torch.cuda.synchronize()
    q.put((rank, lo, hi, m, K, beta.cpu().numpy(), var.cpu().numpy(), pval.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("force_m", [-1, 3])
def test_two_ranks_equal_one(engine, force_m):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (7 if force_m > 0 else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, force_m, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    from poolgen_amd.distributed import ols_with_covariate_sharded
    G, Y, n, p = _data(force_m)
    m, K, beta, var, pval = ols_with_covariate_sharded(engine, G, p, Y, force_m=force_m, n=n, want_K=True)
    assert res[0][3] == res[1][3] == m and (force_m < 0 or m == force_m)
    assert res[0][2] == res[1][1] and res[0][1] == 0 and res[1][2] == p
    for r in res:
        assert np.allclose(r[4], K, rtol=1e-13, atol=0)
    # m > 0: the covariates are eigenvectors of a K that differs in its last bits (summation order of the two partial sums);
    # that moves the fits by ~1e-13 absolute (tests/test_gpu_exact.py: either K is within 1e-10 of the binary128 chain)
    rtol, atol = (1e-12, 1e-12) if m == 0 else (1e-10, 1e-11)
    for i, ref in ((5, beta), (6, var), (7, pval)):
        cat = np.concatenate([res[0][i], res[1][i]], axis=0)
        assert np.allclose(cat, ref.cpu().numpy(), rtol=rtol, atol=atol, equal_nan=True)


def test_rccl_inside_the_library_one_rank(engine):
    """RCCL really loads and all-reduces through libpoolgen_hip (pg_comm_*): a 1-rank communicator is all a 1-GPU box
    allows (RCCL refuses two ranks on one device), and on it the sum over ranks is the identity -- so the sharded entry
    point must reproduce pg_ols_kinship_dev bit for bit, and a buffer must come back unchanged."""
    from poolgen_amd import Engine
    from poolgen_amd import synth
    eng = Engine(0)
    assert eng.comm_size == 1 and eng.comm_rank == 0
    v = eng.comm_version()
    assert v >= 20000, v                        # ncclGetVersion of the loaded RCCL (2.x.y -> 2xxyy)
    eng.comm_init(eng.comm_unique_id(), 1, 0)
    assert eng.comm_size == 1 and eng.comm_rank == 0
    x = torch.arange(40000, dtype=torch.float64, device="cuda") * 0.25 - 3.0
    y = x.clone()
    eng.allreduce_sum(y)
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    n, p = 200, 30_011
    G = synth.genotype_matrix(p, n, "cuda", seed=9)
    Y = synth.phenotypes(G[:4096], n, k=2)
    for force_m in (-1, 2):
        m, K, b, v_, pv = eng.ols_with_covariate_sharded(G, p, Y, 0.75, force_m, n=n, want_K=True)
        m0, K0, b0, v0, p0 = engine.ols_with_covariate(G, Y, 0.75, force_m, n=n)
        assert m == m0 and np.array_equal(K, K0)
        assert torch.equal(b, b0) and torch.equal(v_, v0) and torch.equal(pv, p0)
    eng.profile(True); eng.profile_reset()
    eng.ols_with_covariate_sharded(G, p, Y, 0.75, n=n)
    ms, launches = eng.profile_get("allreduce")
    assert launches == 1 and ms >= 0.0
    with pytest.raises(Exception):
        eng.comm_init(eng.comm_unique_id(), 1, 0)   # one communicator per context
    assert eng._comm_ready
    eng.comm_destroy()
    assert eng.comm_size == 1 and not eng._comm_ready
    # without a communicator the all-reduce is the identity: a slab that is not the whole matrix must be refused, not summed wrongly
    with pytest.raises(Exception, match="communicator"):
        eng.ols_with_covariate_sharded(G[:1000].contiguous(), p, Y, 0.75, n=n)
    eng.close()
