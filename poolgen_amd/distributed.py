"""Locus-sharded ols_iter_with_kinship over one process per GPU (torch.distributed, RCCL).

Every column of G is an independent fit once (Z, ytilde) are known, so loci are split into
contiguous per-rank slabs.  The only exchange is ONE all-reduce(sum) of the
n x n partial kinship sums (320 KB at n = 200; latency-bound, xGMI bandwidth is irrelevant);
every rank then runs the same deterministic n x n eigen step and sweeps its own slab.  Outputs
stay sharded: rank r holds rows [lo_r, hi_r) -- the host concatenates in rank order.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_range(p_total: int, rank: int, world: int):
    """Contiguous slab [lo, hi) of rank `rank` (slab sizes differ by at most one locus)."""
    lo = (p_total * rank) // world
    hi = (p_total * (rank + 1)) // world
    return lo, hi


def _all_agree(ok: bool, group=None) -> bool:
    """True on every rank iff `ok` is True on every rank (MIN all-reduce of a flag over torch.distributed)."""
    on_gpu = str(dist.get_backend(group)).lower().startswith("nccl")
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=(torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu"))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


def setup_comm(engine, group=None, force: bool = False) -> bool:
    """Give `engine` the library's own RCCL communicator over the ranks of the torch.distributed job: rank 0 draws the
    unique id, torch.distributed only carries those 128 bytes to the other ranks (its store / object broadcast), then
    every rank runs ncclCommInitRank inside libpoolgen_hip (pg_comm_init_rank).

    Returns the SAME answer on every rank: True = every rank holds a communicator of `world` ranks; False = no rank does and
    the caller stays on torch.distributed's all-reduce -- a single rank, an engine without RCCL entry points (the CPU
    stand-in of the gloo tests), POOLGEN_COMM != rccl, or any rank failing at any stage (RCCL not loadable, id not drawn,
    ncclCommInitRank refused -- e.g. several ranks on one GPU).  Every stage ends in an agreement over torch.distributed
    before the next one starts, so no rank is left alone inside a collective (a rank that cannot load RCCL never lets the
    others enter ncclCommInitRank)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    world = dist.get_world_size(group)
    if (world <= 1 and not force) or not hasattr(engine, "comm_init") or os.environ.get("POOLGEN_COMM", "rccl") != "rccl":
        return False
    # a communicator may already be there -- but "already there" must be the answer of EVERY rank (a rank that lost its
    # communicator, e.g. an engine re-created after a partial comm_destroy, would otherwise enter the collectives below alone)
    ready = bool(getattr(engine, "_comm_ready", False) and engine.comm_size == world)
    if _all_agree(ready, group):
        return True
    if getattr(engine, "_comm_ready", False):   # some rank is not ready: everybody starts from nothing
        try:
            engine.comm_destroy()
        except Exception:
            pass
    rank = dist.get_rank(group)
    # stage 1: can every rank load RCCL at all?
    try:
        engine.comm_version()
        ok = True
    except Exception:
        ok = False
    if not _all_agree(ok, group):
        return False
    # stage 2: the id.  The broadcast always runs; a failure travels as None.
    box = [None]
    if rank == 0:
        try:
            box[0] = engine.comm_unique_id()
        except Exception:
            box[0] = None
    dist.broadcast_object_list(box, src=0, group=group)
    if box[0] is None:
        return False
    # stage 3: the collective init, then agree on its outcome; a partial success is undone everywhere
    try:
        engine.comm_init(box[0], world, rank)
        ok = engine.comm_size == world
    except Exception:
        ok = False
    if not _all_agree(ok, group):
        try:
            engine.comm_destroy()
        except Exception:
            pass
        return False
    return True


def ols_with_covariate_sharded(engine, G_local: torch.Tensor, p_total: int, Y, var_explained=0.75,
                               force_m: int = -1, n: int | None = None, out=None, group=None, want_K: bool = False):
    """One step of the sharded path.  Returns (m, K, beta_local, var_local, pval_local); K is None unless
    want_K (the caller rarely needs the kinship matrix itself)."""
    if hasattr(engine, "set_phenotypes"):
        # POOLGEN_TWO_PASS=1 keeps the plain two-pass path (kinship, then a full sweep) for measurement
        engine.set_phenotypes(None if os.environ.get("POOLGEN_TWO_PASS") == "1" or force_m > 0 else Y)  # lets the kinship pass pre-compute the intercept-only fits
    S = engine.kinship_partial(G_local, n)
    if getattr(engine, "_comm_ready", False):
        engine.allreduce_sum(S)                      # RCCL inside the library, on the engine's stream
    elif dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(S, op=dist.ReduceOp.SUM, group=group)   # engines without their own communicator (gloo rehearsals)
    m, K, _ = engine.kinship_set(S, p_total, Y, var_explained, force_m, want_K=want_K)
    k = 1 if getattr(Y, "ndim", 1) == 1 else Y.shape[1]
    beta, var, pval = engine.ols_sweep(G_local, k, n, out)
    return m, K, beta, var, pval
