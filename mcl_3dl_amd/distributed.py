"""Particle sharding across the GPUs of one node.

Particles are independent until pf::measure's `sum` (include/mcl_3dl/pf.h:256-260), so each rank owns a contiguous block
of particles, the map structures and the scan are replicated, and one update needs exactly ONE collective: an
all-reduce(SUM) of 2 + 2*world doubles that carries
    [0] sum of w_new            [1] sum of w_new * ln(w_new)
    [2 + 2r], [3 + 2r]          rank r's  max match_ratio  and  -min match_ratio  (every other rank contributes 0)
so max / min fall out of the same SUM (slot r is written by rank r only).  16-80 bytes: latency-bound over xGMI.
The functions here are backend-agnostic (`nccl` == RCCL on the GPU box, `gloo` in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world, rank):
    """Contiguous block [lo, hi) of rank `rank` when n particles are split over `world` ranks (sizes differ by <= 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_partials(partial4, rank, world, out=None):
    """partial4 = [sum w, sum w ln w, max ratio, -min ratio] of this shard -> the 2 + 2*world vector to all-reduce."""
    if out is None:
        out = torch.zeros(2 + 2 * world, dtype=torch.float64, device=partial4.device)
    else:
        out.zero_()
    out[0:2] = partial4[0:2]
    out[2 + 2 * rank:4 + 2 * rank] = partial4[2:4]
    return out


def unpack_totals(packed, out=None):
    """all-reduced vector -> total4 = [sum w, sum w ln w, max ratio, -min ratio] over every shard."""
    if out is None:
        out = torch.zeros(4, dtype=torch.float64, device=packed.device)
    out[0:2] = packed[0:2]
    out[2] = packed[2::2].max()
    out[3] = packed[3::2].max()
    return out


def allreduce_partials(partial4, group=None, scratch=None, out=None):
    """The single collective of one measurement update.  Returns total4 (same on every rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if out is None:
            return partial4.clone()
        out.copy_(partial4)
        return out
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    packed = pack_partials(partial4, rank, world, out=scratch)
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    return unpack_totals(packed, out=out)
