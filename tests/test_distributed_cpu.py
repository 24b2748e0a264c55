"""world_size-2 gloo rehearsal of the locus-sharded path.  A stand-in engine
built on the CPU oracle exercises exactly the host logic that runs on the GPUs: contiguous shards,
ONE all-reduce of the partial kinship, replicated eigen step, per-rank sweep, rank-order concat."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))


class OracleEngine:
    """Same surface as poolgen_amd.Engine for the three staged calls, computed by the oracle."""

    def __init__(self):
        import oracle_lib
        self.o = oracle_lib.load()

    def kinship_partial(self, G, n=None):
        Gn = G.numpy()
        return torch.from_numpy(self.o.kinship(Gn, threads=1) * Gn.shape[0])

    def kinship_set(self, S, p_total, Y, var_explained=0.75, force_m=-1, want_K=True):
        K = S.numpy() / p_total
        ev, V = self.o.sym_eig(K)
        m = force_m if force_m >= 0 else self.o.n_eigenvecs(ev, var_explained)
        self.C, self.Y = V[:, :m].copy(), np.asarray(Y, dtype=float).reshape(K.shape[0], -1)
        return m, K, ev

    def ols_sweep(self, G, k, n=None, out=None):
        r = self.o.ols_with_covariate(G.numpy(), self.Y, covariate=self.C, threads=1)
        return tuple(torch.from_numpy(r[x]) for x in ("beta", "var", "pval"))


def _data():
    rng = np.random.default_rng(42)
    n, p = 12, 301
    G = np.clip(rng.random((p, 1)) * 0.8 + 0.1 + 0.08 * rng.normal(size=(p, n)), 0.01, 0.99)
    Y = rng.normal(size=(n, 2)) + G[7][:, None] * np.array([2.0, -1.0])
    return G, Y


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from poolgen_amd.distributed import ols_with_covariate_sharded, shard_range
    G, Y = _data()
    lo, hi = shard_range(G.shape[0], rank, world)
    eng = OracleEngine()
    m, K, beta, var, pval = ols_with_covariate_sharded(eng, torch.from_numpy(G[lo:hi].copy()), G.shape[0], Y,
                                                       var_explained=0.995, want_K=True)
    q.put((rank, lo, hi, m, K, beta.numpy(), var.numpy(), pval.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_equals_single_process():
    from poolgen_amd.distributed import shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import oracle_lib
    o = oracle_lib.load()
    G, Y = _data()
    ref = o.ols_with_covariate(G, Y, var_explained=0.995, threads=1)
    assert res[0][3] == res[1][3] == ref["m"] and ref["m"] >= 1
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 150, 150, 301)
    for r in res:
        assert np.allclose(r[4], ref["K"], rtol=1e-13, atol=0)
    for i, key in ((5, "beta"), (6, "var"), (7, "pval")):
        cat = np.concatenate([res[0][i], res[1][i]], axis=0)
        # [1 | v1 ...] is nearly collinear (cond(Z'Z) ~ 6e5 here), so last-bit differences of the
        # all-reduced K (summation order) are amplified to ~1e-9 by the literal normal equations
        assert np.allclose(cat, ref[key], rtol=1e-6, atol=1e-12), key
