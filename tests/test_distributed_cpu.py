"""CPU test of the N > 1 path: two gloo ranks each own a particle shard, compute their partials, run the single
all-reduce of mcl_3dl_amd.distributed and normalise their shard; the stitched result must equal pf::measure over the
whole particle set (checked against the oracle's pf_measure, which is the reference's code when oracle/_ref is built).
The per-shard arithmetic below is the same arithmetic the pf_partial / pf_apply kernels perform."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mcl_3dl_amd.distributed import allreduce_partials, pack_partials, shard_bounds, unpack_totals
from oracle import pyoracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partials(w, lik, ratio):
    wn = (w * lik).astype(np.float32)
    pos = wn > 0
    t = np.sum(wn[pos].astype(np.float64) * np.log(wn[pos].astype(np.float64)))
    return wn, np.array([wn.sum(dtype=np.float64), t, max(0.0, ratio.max(initial=0.0)),
                         max(-1.0, (-ratio).max(initial=-1.0))], np.float64)


def _worker(rank, world, port, w, lik, ratio, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(len(w), world, rank)
    wn, part = _partials(w[lo:hi], lik[lo:hi], ratio[lo:hi])
    total = allreduce_partials(torch.from_numpy(part)).numpy()
    S = np.float32(total[0])
    w_out = (wn / S).astype(np.float32) if S > 0 else w[lo:hi]
    entropy = np.float32(np.log(total[0]) - total[1] / total[0]) if S > 0 else np.float32("nan")
    np.savez(out_path % rank, w=w_out, entropy=entropy, rmax=total[2], rmin=-total[3], lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (1, 7, 64, 4096, 262144, 65537):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(n, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip():
    world = 4
    parts = [torch.tensor([1.0 + r, -0.5 * r, 0.1 * r, -1.0 + 0.2 * r], dtype=torch.float64) for r in range(world)]
    packed = sum(pack_partials(p, r, world) for r, p in enumerate(parts))
    total = unpack_totals(packed)
    assert total[0] == sum(p[0] for p in parts) and total[1] == sum(p[1] for p in parts)
    assert total[2] == max(p[2] for p in parts) and total[3] == max(p[3] for p in parts)


@pytest.mark.parametrize("world", [2])
def test_two_rank_update_matches_single_process(tmp_path, world):
    rng = np.random.default_rng(42)
    n = 1001  # not divisible by the world size
    w = rng.uniform(0.5, 1.5, n).astype(np.float32)
    w /= w.sum()
    lik = rng.uniform(0.0, 900.0, n).astype(np.float32)
    lik[rng.integers(0, n, 50)] = 0.0  # dead particles
    ratio = rng.uniform(0.1, 0.95, n).astype(np.float32)
    out_path = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(world, _free_port(), w, lik, ratio, out_path), nprocs=world, join=True)
    parts = [np.load(out_path % r) for r in range(world)]
    w_got = np.concatenate([p["w"] for p in parts])
    kind = "ref" if pyoracle.available("ref") else "port"
    w_want, ent_want, restored = pyoracle.Oracle(kind).pf_measure(w, lik)
    assert not restored
    np.testing.assert_allclose(w_got, w_want, rtol=1e-5)
    for p in parts:  # every rank derived the same global statistics
        np.testing.assert_allclose(p["entropy"], ent_want, rtol=1e-5)
        assert np.float32(p["rmax"]) == ratio.max() and np.float32(p["rmin"]) == ratio.min()


def test_all_dead_restores_on_every_rank(tmp_path):
    n = 64
    w = np.full(n, 1.0 / n, np.float32)
    lik = np.zeros(n, np.float32)
    ratio = np.zeros(n, np.float32)
    out_path = str(tmp_path / "dead%d.npz")
    mp.spawn(_worker, args=(2, _free_port(), w, lik, ratio, out_path), nprocs=2, join=True)
    for r in range(2):
        p = np.load(out_path % r)
        np.testing.assert_array_equal(p["w"], w[p["lo"]:p["hi"]])
        assert np.isnan(p["entropy"])


# ---- sharded resampling (multi-GPU clause) ---------------------------------------------------------
class _OracleResampleOps:
    """CPU stand-in with the interface of mcl_3dl_amd.distributed.EngineResampleOps, backed by the plain-C oracle."""

    def __init__(self):
        self.o = pyoracle.Oracle("port")

    def begin(self, weight_all):
        self.w = weight_all.numpy().copy()
        return self.o.resample_pstep(self.w, len(self.w))

    def plan(self, initial_p):
        self.src, self.dup = self.o.resample_plan(self.w, len(self.w), 0, initial_p)
        return self.src, self.dup, int(self.dup.sum())

    def apply_slice(self, state_all, noise13, lo, count):
        slot = np.cumsum(self.dup) - 1  # noise is indexed by the GLOBAL duplicate slot
        nz = np.zeros((len(self.src), 13), np.float32)
        nz[self.dup > 0] = noise13[slot[self.dup > 0]]
        # the oracle consumes noise in order of the duplicates it meets: hand it the slice's own
        sl = slice(lo, lo + count)
        return torch.from_numpy(self.o.resample_apply(state_all.numpy(), self.src[sl], self.dup[sl],
                                                      nz[sl][self.dup[sl] > 0]))


def _resample_worker(rank, world, port, s, w, initial_frac, noise_all, out_path):
    from mcl_3dl_amd.distributed import sharded_resample
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(len(w), world, rank)
    new_s, new_w, (src, dup) = sharded_resample(
        _OracleResampleOps(), torch.from_numpy(s[lo:hi].copy()), torch.from_numpy(w[lo:hi].copy()), len(w),
        lambda pstep: np.float32(pstep) * np.float32(initial_frac), lambda n_dup: noise_all[:n_dup])
    np.savez(out_path % rank, s=new_s.numpy(), w=new_w.numpy(), src=src, dup=dup)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,dead", [(1001, 0), (512, 100)])
def test_two_rank_resample_matches_single_process(tmp_path, n, dead):
    import resample_cases as rc
    s, w = rc.make_case(n, dead)
    rng = np.random.default_rng(7)
    noise_all = rng.normal(0, 0.05, (n, 13)).astype(np.float32)
    frac = 0.37
    o = pyoracle.Oracle("port")
    pstep = o.resample_pstep(w, n)
    src, dup = o.resample_plan(w, n, 0, np.float32(pstep) * np.float32(frac))
    want = o.resample_apply(s, src, dup, noise_all[:int(dup.sum())])
    out_path = str(tmp_path / "rs%d.npz")
    mp.spawn(_resample_worker, args=(2, _free_port(), s, w, frac, noise_all, out_path), nprocs=2, join=True)
    parts = [np.load(out_path % r) for r in range(2)]
    np.testing.assert_array_equal(parts[0]["src"], src)
    np.testing.assert_array_equal(parts[1]["dup"], dup)
    np.testing.assert_array_equal(np.concatenate([p["s"] for p in parts]), want)
    assert all(np.all(p["w"] == np.float32(1.0 / n)) for p in parts)


def test_group_shard_bookkeeping_matches_the_python_rule():
    """mcl3dl_hip_group_shard (the C side's contiguous shards, host_group.h) == distributed.shard_bounds for every
    (particles, devices) pair, including fewer particles than devices; the shards tile [0, n) exactly."""
    from mcl_3dl_amd import capi
    from mcl_3dl_amd.distributed import shard_bounds
    for n in (0, 1, 2, 7, 64, 100, 4096, 262144, 262145):
        for world in (1, 2, 3, 4, 8, 64):
            end = 0
            for r in range(world):
                lo, cnt = capi.group_shard(n, world, r)
                assert (lo, lo + cnt) == shard_bounds(n, world, r)
                assert lo == end
                end = lo + cnt
            assert end == n
    with pytest.raises(capi.EngineError):
        capi.group_shard(10, 4, 4)
