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
