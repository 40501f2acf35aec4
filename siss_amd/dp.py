"""Data-parallel semantics of the SISS step, device-agnostic host logic.

Every rank computes the flat pair [g_x ; g_a] on its own shard with the loss normalised by the
GLOBAL batch; ONE sum all-reduce of that buffer per optimizer step makes every rank hold
g_x = sum_r g_x^r, g_a = sum_r g_a^r; the norm-fix / recombine / clip / AdamW that follow are
replicated.  (The reference's DDP only reduces the first of its two backward passes -- SURVEY.md §5
-- so there is no reference multi-GPU semantics to match; the definition here is global-batch
equivalence with the single-process step.)  Used by SISSStepper on RCCL and by the gloo CPU tests.
"""
import math

import torch
import torch.distributed as dist


def allreduce_flat_grads(flat_pair, group=None):
    """In-place sum over ranks of the [2, P] gradient buffer -- the only collective of a step."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_pair, op=dist.ReduceOp.SUM, group=group)
    return flat_pair


_SCRATCH = {}


def direct_exchange_flat_grads(flat_pair, group=None):
    """The same sum over ranks as `allreduce_flat_grads`, as a DIRECT reduce-scatter + all-gather:
    every rank sends slice j of its flat pair straight to rank j (one all-to-all: all 7 xGMI links of a GPU carry
    S/8 each, concurrently, instead of a ring pushing 2 (N-1)/N S over one link), sums the N slices it received in
    rank order, and the reduced slices are all-gathered back.  Each element is summed by exactly ONE rank in a fixed
    order, so the replicas are bit-identical by construction.  Falls back to the all-reduce when the buffer does not
    split evenly."""
    if not (dist.is_available() and dist.is_initialized()):
        return flat_pair
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world <= 1:
        return flat_pair
    flat = flat_pair.view(-1)
    n = flat.numel()
    if n % world != 0 or not flat_pair.is_contiguous():
        return allreduce_flat_grads(flat_pair, group)
    shard = n // world
    key = (flat.device, flat.dtype, n, world)
    scratch = _SCRATCH.get(key)
    if scratch is None:
        scratch = _SCRATCH[key] = torch.empty(n + shard, dtype=flat.dtype, device=flat.device)
    recv, mine = scratch[:n], scratch[n:]
    dist.all_to_all_single(recv, flat, group=group)                 # recv[i*shard:(i+1)*shard] = rank i's slice `rank`
    torch.sum(recv.view(world, shard), dim=0, out=mine)             # fixed rank order
    dist.all_gather_into_tensor(flat, mine, group=group)
    return flat_pair


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
EXCHANGES = {"allreduce": allreduce_flat_grads, "direct": direct_exchange_flat_grads, "sharded": None}


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
