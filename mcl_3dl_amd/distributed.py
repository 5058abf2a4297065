"""Particle sharding across the GPUs of one node.

Particles are independent until pf::measure's `sum` (include/mcl_3dl/pf.h:256-260), so each rank owns a contiguous block
of particles, the map structures and the scan are replicated, and one update needs exactly ONE collective: an
all-reduce(SUM) of 2 + 2*world doubles that carries
    [0] sum of w_new            [1] sum of w_new * ln(w_new)
    [2 + 2r], [3 + 2r]          rank r's  max match_ratio  and  -min match_ratio  (every other rank contributes 0)
so max / min fall out of the same SUM (slot r is written by rank r only).  16-80 bytes: latency-bound over xGMI.
The functions here are backend-agnostic (`nccl` == RCCL on the GPU box, `gloo` in the CPU tests).

Resampling is the one step with a real exchange: pf::resample (include/mcl_3dl/pf.h:187-225) walks
the GLOBAL prefix sums of the weights, so an output slot on one GPU may copy a particle that lives on another.
`sharded_resample` all-gathers the weights (4 B/particle) and the 13-float states (52 B/particle; 13.6 MB at 262 144
particles, one collective each), every rank builds the same plan, and each rank fills its own output slice.

Stream contract.  The engine enqueues its kernels on its own context stream (`Engine.set_stream`; by default a private
non-blocking stream), torch enqueues its fills and collectives on torch's current stream.  Every helper below that mixes
the two calls `_fence(engine, tensor)` around the engine call: a no-op when the engine has been bound to torch's current
stream (`engine.set_stream(torch.cuda.current_stream().cuda_stream)` — what bench.py does, and the fast path), otherwise a
full synchronisation of both streams, so that a collective can never read a record the engine has not written yet and a
zero-fill can never land after the engine's write.
"""
import torch
import torch.distributed as dist


def _fence(engine, tensor):
    """Order torch's current stream and the engine's stream against each other (see the module docstring)."""
    get = getattr(engine, "get_stream", None)
    if get is None or not tensor.is_cuda:
        return  # CPU tensors / test doubles: nothing is asynchronous
    cur = torch.cuda.current_stream(tensor.device)
    if get() == cur.cuda_stream and cur.cuda_stream != 0:
        return
    cur.synchronize()
    engine.synchronize()


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


def all_gather_shards(local, n_total, group=None):
    """Concatenate the contiguous shards (shard_bounds) of every rank along dim 0. Shards may differ by one row."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_total, world, r) for r in range(world)]
    rows = max(hi - lo for lo, hi in sizes)
    if local.shape[0] < rows:
        pad = torch.zeros((rows - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((world * rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    if all(hi - lo == rows for lo, hi in sizes):
        return out
    return torch.cat([out[r * rows:r * rows + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


class EngineResampleOps:
    """The three resampling calls of the C ABI on device tensors (mcl_3dl_amd/capi.py:Engine)."""

    def __init__(self, engine):
        self.engine = engine

    def begin(self, weight_all):
        _fence(self.engine, weight_all)  # the all-gather that produced weight_all ran on torch's stream
        return self.engine.resample_begin_device(weight_all, weight_all.shape[0])

    def plan(self, initial_p):
        return self.engine.resample_plan(0, initial_p)

    def apply_slice(self, state_all, noise13, lo, count):
        out = torch.empty((count, 13), dtype=torch.float32, device=state_all.device)
        _fence(self.engine, state_all)
        self.engine.resample_apply_device(state_all, noise13, out, lo, count)
        _fence(self.engine, out)
        return out


def sharded_resample(ops, state_local, weight_local, n_total, draw_initial_p, draw_noise, group=None):
    """pf::resample over particles sharded by shard_bounds. Every rank must pass callbacks that return the SAME draws
    (the reference's single random engine: seed it identically on every rank, or draw on rank 0 and broadcast):
    draw_initial_p(pstep) -> float in [0, pstep) (pf.h:203), draw_noise(n_dup) -> (n_dup, 13) noise states in slot order
    (pf.h:216). Returns (this rank's new states, its new uniform weights 1/n_total (pf.h:207), (source, dup) plan)."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    lo, hi = shard_bounds(n_total, world, rank)
    weight_all = all_gather_shards(weight_local, n_total, group)
    state_all = all_gather_shards(state_local, n_total, group)
    pstep = ops.begin(weight_all)
    source, dup, n_dup = ops.plan(draw_initial_p(pstep))
    noise = draw_noise(n_dup)
    new_state = ops.apply_slice(state_all, noise, lo, hi - lo)
    new_weight = torch.full((hi - lo,), 1.0 / n_total, dtype=torch.float32, device=weight_local.device)
    return new_state, new_weight, (source, dup)


def sharded_expectation(engine, d_pose, d_weight, d_bias, n_local, n_total, group=None):
    """pf::expectationBiased + max + maxBiased (include/mcl_3dl/pf.h:294-303, 361-390) over particle shards: one
    16-double record per rank, all-gathered (128 B per rank), combined on the host in rank order.
    Returns (mean7, total weight, global index of the max-weight particle, of the max biased-weight particle)."""
    rec = torch.zeros(16, dtype=torch.float64, device=d_pose.device)
    _fence(engine, rec)
    engine.moments_partial_device(d_pose, d_weight, d_bias, n_local, rec)
    _fence(engine, rec)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        allrec = torch.empty(16 * world, dtype=torch.float64, device=rec.device)
        dist.all_gather_into_tensor(allrec, rec, group=group)
    else:
        world, allrec = 1, rec
    offsets = [shard_bounds(n_total, world, r)[0] for r in range(world)]
    return engine.moments_finish(allrec.cpu().numpy(), offsets)


def sharded_covariance(engine, d_pose, d_weight, n_local, mean7, group=None):
    """pf::covariance (pf.h:304-360, pass ratio 1, all particles) over particle shards: 22 sums per rank, one
    all-reduce(SUM) of 176 bytes, host division."""
    rec = torch.zeros(22, dtype=torch.float64, device=d_pose.device)
    _fence(engine, rec)
    engine.covariance_partial_device(d_pose, d_weight, n_local, mean7, rec)
    _fence(engine, rec)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=group)
    return engine.covariance_finish(rec.cpu().numpy())
