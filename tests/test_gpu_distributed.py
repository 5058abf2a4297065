"""GPU test of the N > 1 path with REAL processes: two ranks, each with its own engine context (both on GPU 0 — the test
boxes expose one GPU, so the ranks time-slice it), own a particle shard, run measure_device + pf_partial_device, all-reduce
the packed partials through torch.distributed (gloo here: two ranks cannot share one device under RCCL; bench.py --gpus N
runs the same calls over backend "nccl"), apply, and resample their output slice with sharded_resample. The stitched result
must equal the single-context update and resampling plan."""
import os
import socket

import numpy as np
import pytest

from mcl_3dl_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sc_args, w0, noise_all, out_path):
    import torch  # before the engine library: it initialises the HIP runtime
    import torch.distributed as dist
    from mcl_3dl_amd import capi
    from mcl_3dl_amd.distributed import (EngineResampleOps, allreduce_partials, shard_bounds, sharded_covariance,
                                         sharded_expectation, sharded_resample)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    sc = make_scene(**sc_args)
    eng = capi.Engine(0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)
    eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 5.0))
    eng.set_beam_params(num_points=len(sc.scan_beam))
    eng.upload_scan(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
    n = len(sc.poses)
    lo, hi = shard_bounds(n, world, rank)
    m = hi - lo
    d_pose = torch.from_numpy(sc.poses[lo:hi].copy()).to(dev)
    d_w = torch.from_numpy(w0[lo:hi].copy()).to(dev)
    d_lik, d_ratio, d_beam = (torch.empty(m, dtype=torch.float32, device=dev) for _ in range(3))
    d_pack = torch.zeros(2 + 2 * world, dtype=torch.float64, device=dev)
    d_stats = torch.zeros(4, dtype=torch.float32, device=dev)
    eng.measure_device(d_pose, m, d_lik, d_ratio, d_beam)
    eng.pf_partial_device(d_w, d_lik, d_beam, None, d_ratio, m, d_pack, rank, world)
    torch.cuda.synchronize()
    host_pack = d_pack.cpu()                       # gloo reduces host tensors
    dist.all_reduce(host_pack, op=dist.ReduceOp.SUM)
    d_pack.copy_(host_pack)
    eng.pf_apply_device(d_w, m, d_pack, d_stats, world)
    torch.cuda.synchronize()
    # the reductions that follow the update in the node (expectationBiased / max / covariance), over the shards
    mean7, total, imax, ibias = sharded_expectation(eng, d_pose, d_w, None, m, n)
    cov = sharded_covariance(eng, d_pose, d_w, m, mean7)
    # resampling over the shards (states = pose + 6 zero odometry-error terms)
    st = np.zeros((m, 13), np.float32)
    st[:, :7] = sc.poses[lo:hi]
    new_s, new_w, (src, dup) = sharded_resample(EngineResampleOps(eng), torch.from_numpy(st).to(dev), d_w, n,
                                                lambda pstep: np.float32(pstep) * np.float32(0.41),
                                                lambda n_dup: noise_all[:n_dup])
    torch.cuda.synchronize()
    np.savez(out_path % rank, w=d_w.cpu().numpy(), stats=d_stats.cpu().numpy(), lik=d_lik.cpu().numpy(),
             beam=d_beam.cpu().numpy(), new_s=new_s.cpu().numpy(), src=src, dup=dup, mean7=mean7, total=total,
             imax=imax, ibias=ibias, cov=cov)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def test_two_processes_shard_one_update_and_resample(engine, tmp_path):
    import torch.multiprocessing as mp
    sc_args = dict(n=91, n_p=301, n_s=700, n_b=32, seed=9)
    sc = make_scene(**sc_args)
    n = len(sc.poses)
    rng = np.random.default_rng(4)
    w0 = rng.uniform(0.5, 1.5, n).astype(np.float32)
    w0 /= w0.sum()
    noise_all = rng.normal(0, 0.02, (n, 13)).astype(np.float32)
    out_path = str(tmp_path / "rank%d.npz")
    mp.get_context("spawn")
    mp.spawn(_worker, args=(2, _free_port(), sc_args, w0, noise_all, out_path), nprocs=2, join=True)
    parts = [np.load(out_path % r) for r in range(2)]
    # single context, whole particle set
    engine.set_map(sc.map_xyz, sc.map_label, stamp=9100, dist_weight=(1.0, 1.0, 5.0))
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=len(sc.scan_beam))
    try:
        whole = engine.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        np.testing.assert_array_equal(np.concatenate([p["lik"] for p in parts]), whole["lik"])
        np.testing.assert_array_equal(np.concatenate([p["beam"] for p in parts]), whole["beam"])
        w_sharded = np.concatenate([p["w"] for p in parts])
        np.testing.assert_allclose(w_sharded, whole["weights"], rtol=2e-7)
        for p in parts:  # both ranks hold the same global statistics
            np.testing.assert_allclose(p["stats"][0], whole["entropy"], rtol=1e-6)
            assert p["stats"][1] == np.float32(whole["match_ratio_min"]) and p["stats"][2] == np.float32(whole["match_ratio_max"])
        np.testing.assert_array_equal(parts[0]["stats"], parts[1]["stats"])
        # reductions: every rank holds the same mean / covariance / arg-max, equal to the single-context ones
        mean_w, total_w, imax_w, ibias_w = engine.expectation(sc.poses, w_sharded)
        cov_w = engine.covariance(sc.poses, w_sharded, mean_w)
        for p in parts:
            np.testing.assert_allclose(p["mean7"][:3], mean_w[:3], rtol=2e-6)
            dot = abs(float(np.dot(p["mean7"][3:], mean_w[3:])))
            assert 2.0 * np.arccos(min(1.0, dot)) < 5e-4  # same rotation (tests/test_gpu_moments.py explains the metric)
            np.testing.assert_allclose(p["total"], total_w, rtol=1e-6)
            assert int(p["imax"]) == imax_w and int(p["ibias"]) == ibias_w
            np.testing.assert_allclose(p["cov"], cov_w, rtol=2e-3, atol=1e-9)
        np.testing.assert_array_equal(parts[0]["mean7"], parts[1]["mean7"])
        np.testing.assert_array_equal(parts[0]["cov"], parts[1]["cov"])
        # resampling: the plan of the sharded run equals the plan on the stitched weights, and so do the states
        st = np.zeros((n, 13), np.float32)
        st[:, :7] = sc.poses
        pstep = engine.resample_begin(w_sharded)
        src, dup, nd = engine.resample_plan(0, np.float32(pstep) * np.float32(0.41))
        want = engine.resample_apply(st, noise_all[:nd])
        np.testing.assert_array_equal(parts[0]["src"], src)
        np.testing.assert_array_equal(parts[1]["dup"], dup)
        np.testing.assert_array_equal(np.concatenate([p["new_s"] for p in parts]), want)
    finally:
        engine.set_beam_params()
