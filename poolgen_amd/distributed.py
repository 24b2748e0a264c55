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


def setup_comm(engine, group=None, force: bool = False) -> bool:
    """Give `engine` the library's own RCCL communicator over the ranks of the torch.distributed job: rank 0 draws the
    unique id, torch.distributed only carries those 128 bytes to the other ranks (its store / object broadcast), then
    every rank runs ncclCommInitRank inside libpoolgen_hip (pg_comm_init_rank).  Returns False -- and the caller stays
    on torch.distributed's all-reduce -- when there is a single rank, when the engine has no RCCL entry points (the CPU
    stand-in of the gloo tests) or when several ranks share one GPU (RCCL refuses that; the 1-GPU rehearsal)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    world = dist.get_world_size(group)
    if (world <= 1 and not force) or not hasattr(engine, "comm_init") or os.environ.get("POOLGEN_COMM", "rccl") != "rccl":
        return False
    if getattr(engine, "_comm_ready", False) and engine.comm_size == world:
        return True
    rank = dist.get_rank(group)
    box = [engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    engine.comm_init(box[0], world, rank)
    engine._comm_ready = True
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
