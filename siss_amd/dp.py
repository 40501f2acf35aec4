"""Data-parallel semantics of the SISS step, device-agnostic host logic.

Every rank computes the flat pair [g_x ; g_a] on its own shard with the loss normalised by the
GLOBAL batch; ONE sum all-reduce of that buffer per optimizer step makes every rank hold
g_x = sum_r g_x^r, g_a = sum_r g_a^r; the norm-fix / recombine / clip / AdamW that follow are
replicated.  (The reference's DDP only reduces the first of its two backward passes -- SURVEY.md §5
-- so there is no reference multi-GPU semantics to match; the definition here is global-batch
equivalence with the single-process step.)  Used by SISSStepper on RCCL and by the gloo CPU tests.
"""
import math
import os

import torch
import torch.distributed as dist

# TEST-ONLY switch: with one rank there is nothing to exchange and every collective is skipped.  SISS_DP_FORCE_COLLECTIVES=1 (or
# dp.FORCE_COLLECTIVES = True) issues them anyway, so that a world-size-1 RCCL communicator on the ONE GPU of a build box exercises
# the real code path -- library load, communicator init, stream ordering against the engine's kernels, the coalescing window, the
# overlap hook -- with results that must equal the no-group step (a sum over one rank is the identity).  tests/test_hip_rccl.py.
FORCE_COLLECTIVES = os.environ.get("SISS_DP_FORCE_COLLECTIVES") == "1"


def active(group=None):
    """True when a step has to run its collectives: a process group of more than one rank (or the test-only force switch)."""
    return bool(dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES))


def allreduce_flat_grads(flat_pair, group=None):
    """In-place sum over ranks of the [2, P] gradient buffer -- the only collective of a step."""
    if active(group):
        dist.all_reduce(flat_pair, op=dist.ReduceOp.SUM, group=group)
    return flat_pair


_SCRATCH = {}


def allreduce_pieces(tensors, group=None, async_op=False):
    """Sum over ranks of several (contiguous) pieces of the flat pair as ONE collective call: the pieces are issued inside
    a coalescing window, which RCCL runs as one grouped launch (ncclGroupStart / End) -- the overlapped exchange is two
    such calls per step ([tail_x, tail_a] from inside the backward, [head_x, head_a] after it), not four all-reduces.
    Returns a handle with .wait() when async_op."""
    if not active(group):
        return None
    try:
        from torch.distributed.distributed_c10d import _coalescing_manager
    except ImportError:                                  # (a torch without the coalescing window: one all-reduce per piece)
        _coalescing_manager = None
    if _coalescing_manager is None:
        works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op) for t in tensors]
        return _Works(works) if async_op else None
    # (no `device`: the fast path -- the window records the all-reduces and hands them to the backend's allreduce_coalesced,
    # which RCCL and gloo both implement)
    with _coalescing_manager(group=group, async_ops=async_op) as cm:
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return cm if async_op else None


class _Works:
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


def can_shard(P, world, align=4):
    """The sharded update hands every rank the parameter range [r P / N, (r + 1) P / N): the ranges must be equal and start
    on 16-byte boundaries (the flat kernels take float4 accesses)."""
    return (world > 1 or FORCE_COLLECTIVES) and P % world == 0 and (P // world) % align == 0


def reduce_scatter_param_shards(flat_pair, group=None):
    """First half of the SHARDED update: rank j receives, and sums in rank order, every rank's slice
    [j P/N, (j+1) P/N) of g_x and of g_a (two all-to-alls, one per gradient set, so that a rank owns the SAME parameter
    range of both sets).  Returns (g_x shard, g_a shard, lo, hi); P must divide by the world size."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nsets, P = flat_pair.shape
    assert nsets == 2 and P % world == 0 and flat_pair.is_contiguous()
    shard = P // world
    key = ("rs", flat_pair.device, flat_pair.dtype, P, world)
    scratch = _SCRATCH.get(key)
    if scratch is None:
        scratch = _SCRATCH[key] = torch.empty(2 * P + 2 * shard, dtype=flat_pair.dtype, device=flat_pair.device)
    out = []
    for s in range(2):
        recv, mine = scratch[s * P:(s + 1) * P], scratch[2 * P + s * shard:2 * P + (s + 1) * shard]
        dist.all_to_all_single(recv, flat_pair[s], group=group)
        torch.sum(recv.view(world, shard), dim=0, out=mine)         # fixed rank order: the same bits whoever asks
        out.append(mine)
    return out[0], out[1], rank * shard, (rank + 1) * shard


def all_gather_params(flat_params, lo, hi, group=None):
    """Second half: every rank contributes its updated shard [lo, hi); afterwards all replicas hold the same parameters
    (each element was written by exactly one rank)."""
    key = ("ag", flat_params.device, flat_params.dtype, hi - lo)
    mine = _SCRATCH.get(key)
    if mine is None:
        mine = _SCRATCH[key] = torch.empty(hi - lo, dtype=flat_params.dtype, device=flat_params.device)
    mine.copy_(flat_params[lo:hi])
    dist.all_gather_into_tensor(flat_params, mine, group=group)
    return flat_params


# "sharded" is not a gradient exchange with the replicated update behind it: SISSStepper handles it (reduce-scatter ->
# shard-local norm-fix / clip / AdamW -> all-gather of the parameters); listed here so that the name validates.
EXCHANGES = {"allreduce": allreduce_flat_grads, "sharded": None}


def recombine_reference(gx, ga, scaling_norm, max_norm=1.0):
    """Closed form of delete_celeb.py:725-767 on flat tensors (host-side check for the DP tests)."""
    nx, na = float(gx.norm()), float(ga.norm())
    s = scaling_norm / na
    g = gx - s * ga
    pre = float(g.norm())
    g = g * min(1.0, max_norm / (pre + 1e-6))
    return g, dict(norm_loss_x=nx, norm_loss_a=na, scaling_factor=s, pre_clip_norm=pre)


def shard_range(n_items, rank, world):
    per = math.ceil(n_items / world)
    return rank * per, min(n_items, (rank + 1) * per)
