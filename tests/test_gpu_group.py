"""Device groups (include/mcl3dl_hip.h): N contexts behind one handle in ONE process — the in-process multi-GPU host the
reference's single C++ process needs.  The GPU box has one MI355X, so
  * N = 1 must equal the plain context bit for bit (direct path), and again through the sharded path with the RCCL
    all-reduce running with one rank (librccl dlopen'ed, ncclCommInitAll, ncclAllReduce on the context's stream);
  * N = 2, 3, 5 contexts on the SAME device (collective = host combine, RCCL needs one GPU per rank) must reproduce the
    unsharded update: likelihood / ratio / beam bit-identical, weights to 2e-7 (only the fp64 sum's association moves)."""
import os
import subprocess

import numpy as np
import pytest

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_scene
from oracle import pyoracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DW = (1.0, 1.0, 5.0)


@pytest.fixture(scope="module")
def scene():
    return make_scene(n=91, n_p=203, n_s=1200, n_b=48, seed=31)


@pytest.fixture(scope="module")
def single(engine, scene):
    sc = scene
    engine.set_map(sc.map_xyz, sc.map_label, stamp=7001, dist_weight=DW)
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=48)
    rng = np.random.default_rng(2)
    w0 = rng.uniform(0.5, 1.5, len(sc.poses)).astype(np.float32)
    w0 /= w0.sum()
    extra = rng.uniform(0.2, 1.0, len(sc.poses)).astype(np.float32)
    return w0, extra, engine.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins,
                                            extra=extra)


def make_group(devices, sc, collective=None):
    g = capi.Group(devices, collective=collective)
    g.set_map(sc.map_xyz, sc.map_label, stamp=7002, dist_weight=DW)
    g.set_likelihood_params()
    g.set_beam_params(num_points=48)
    return g


def test_one_device_group_is_the_plain_context(scene, single):
    sc = scene
    w0, extra, want = single
    g = make_group([0], sc)
    try:
        got = g.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, extra=extra)
        for k in ("lik", "quality", "beam", "weights"):
            np.testing.assert_array_equal(got[k], want[k])
        assert got["entropy"] == want["entropy"] and got["match_ratio_min"] == want["match_ratio_min"]
        lik, ratio, beam = g.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        np.testing.assert_array_equal(lik, want["lik"])
        np.testing.assert_array_equal(beam, want["beam"])
        assert g.collective_stats() == dict(rccl=0, host=0)
    finally:
        g.close()


def test_rccl_all_reduce_with_one_rank(scene, single):
    """The sharded path with N = 1: ncclCommInitAll + ncclAllReduce really run (in-process, on the context's stream)."""
    sc = scene
    w0, extra, want = single
    g = make_group([0], sc)
    try:
        g.set_option("direct_single", 0)
        for _ in range(3):
            got = g.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, extra=extra)
        for k in ("lik", "quality", "beam", "weights"):
            np.testing.assert_array_equal(got[k], want[k])
        assert got["entropy"] == want["entropy"]
        assert g.collective_stats() == dict(rccl=3, host=0)
    finally:
        g.close()


@pytest.mark.parametrize("n_dev", [2, 3, 5])
def test_shards_on_one_gpu_reproduce_the_unsharded_update(scene, single, n_dev):
    sc = scene
    w0, extra, want = single
    g = make_group([0] * n_dev, sc, collective="host")
    try:
        got = g.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, extra=extra)
        for k in ("lik", "quality", "beam"):
            np.testing.assert_array_equal(got[k], want[k])
        np.testing.assert_allclose(got["weights"], want["weights"], rtol=2e-7)
        np.testing.assert_allclose(got["entropy"], want["entropy"], rtol=1e-6)
        assert got["match_ratio_min"] == want["match_ratio_min"] and got["match_ratio_max"] == want["match_ratio_max"]
        assert got["restored"] is False and g.collective_stats()["host"] == 1
        # poses uploaded once, then one batch per model (what the drop-in classes do)
        g.upload_poses(sc.poses)
        lik, ratio, _ = g.measure_batch(None, sc.scan_lik)
        _, _, beam = g.measure_batch(None, None, sc.scan_beam, sc.scan_beam_label, sc.origins)
        np.testing.assert_array_equal(lik, want["lik"])
        np.testing.assert_array_equal(ratio, want["quality"])
        np.testing.assert_array_equal(beam, want["beam"])
    finally:
        g.close()


def test_fewer_particles_than_devices_and_dead_filter(scene, engine):
    """An empty shard contributes the neutral record (sum 0, max ratio 0, -min ratio -1); all weights zero -> restored."""
    sc = scene
    g = make_group([0, 0, 0], sc, collective="host")
    try:
        poses = sc.poses[:2]
        w0 = np.array([0.25, 0.75], np.float32)
        got = g.measure_update(poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        engine.set_map(sc.map_xyz, sc.map_label, stamp=7003, dist_weight=DW)
        engine.set_likelihood_params()
        engine.set_beam_params(num_points=48)
        want = engine.measure_update(poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        np.testing.assert_array_equal(got["lik"], want["lik"])
        np.testing.assert_allclose(got["weights"], want["weights"], rtol=2e-7)
        assert got["match_ratio_min"] == want["match_ratio_min"] and got["match_ratio_max"] == want["match_ratio_max"]
        far = sc.poses[:7].copy()
        far[:, :3] += 500.0
        dead = g.measure_update(far, np.full(7, 1 / 7, np.float32), sc.scan_lik)
        assert dead["restored"] is True
        np.testing.assert_array_equal(dead["weights"], np.full(7, 1 / 7, np.float32))
    finally:
        g.close()


@pytest.mark.parametrize("bad_rank", [0, 1, 2])
def test_a_rank_that_fails_ahead_of_the_collective_does_not_strand_the_others(scene, single, bad_rank, monkeypatch):
    """The update's collective is entered by every rank or by none (host_group.h:VoteBarrier): a rank that fails after its
    kernels are enqueued makes the call return ITS error — promptly, the other ranks stand down instead of waiting inside an
    all-reduce that cannot complete — weights untouched, and the next update on the same group is correct."""
    sc = scene
    w0, extra, want = single
    g = make_group([0, 0, 0], sc, collective="host")
    try:
        with pytest.raises(capi.EngineError, match="test hook"):
            g.set_option("inject_failure_rank", bad_rank)   # off unless the environment enables the hooks
        monkeypatch.setenv("MCL3DL_HIP_TEST_HOOKS", "1")
        g.set_option("inject_failure_rank", bad_rank)
        w_in = w0.copy()
        with pytest.raises(capi.EngineError, match=r"rank %d\): injected failure" % bad_rank):
            g.measure_update(sc.poses, w_in, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, extra=extra)
        np.testing.assert_array_equal(w_in, w0)
        got = g.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, extra=extra)
        for k in ("lik", "quality", "beam"):
            np.testing.assert_array_equal(got[k], want[k])
        np.testing.assert_allclose(got["weights"], want["weights"], rtol=2e-7)
        assert g.collective_stats()["host"] == 1   # the abandoned update combined nothing
    finally:
        g.close()


def test_injected_failure_on_the_rccl_path_rebuilds_the_communicator(scene, single, monkeypatch):
    """One rank, RCCL all-reduce (the only RCCL shape one GPU allows): the failed update destroys the communicator, the next
    one brings it up again and is correct."""
    sc = scene
    w0, extra, want = single
    g = make_group([0], sc)
    try:
        g.set_option("direct_single", 0)
        g.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, extra=extra)
        monkeypatch.setenv("MCL3DL_HIP_TEST_HOOKS", "1")
        g.set_option("inject_failure_rank", 0)
        with pytest.raises(capi.EngineError, match="injected failure"):
            g.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, extra=extra)
        got = g.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, extra=extra)
        np.testing.assert_array_equal(got["weights"], want["weights"])
        assert g.collective_stats() == dict(rccl=2, host=0)
    finally:
        g.close()


def test_group_over_every_device_of_the_box_plain_c():
    """tests/cpp/group_all_devices.c (C99, the C ABI only): one group over mcl3dl_hip_device_count() GPUs with the RCCL
    all-reduce — an N-rank in-process RCCL test wherever this suite meets more than one GPU, one rank here."""
    exe = os.path.join(ROOT, "tests", "cpp", "group_all_devices.bin")
    assert os.path.exists(exe), "tests/cpp/group_all_devices.bin is not built: run __graft_entry__.build()"
    for n_p in ("3001", "5"):   # also fewer particles than a large box has GPUs
        proc = subprocess.run([exe, n_p], capture_output=True, text=True, timeout=600)
        assert proc.returncode == 0, proc.stdout + proc.stderr
        assert "device(s)" in proc.stdout and "mismatching likelihood/ratio/beam 0" in proc.stdout
        print(proc.stdout.strip())


def test_rccl_refuses_repeated_devices_with_a_clear_error(scene):
    g = make_group([0, 0], scene)  # collective left at RCCL
    try:
        with pytest.raises(capi.EngineError, match="more than once"):
            g.measure_update(scene.poses, scene.weights, scene.scan_lik)
    finally:
        g.close()


def test_node_call_site_on_a_device_group(tmp_path, oracle_kind):
    """tests/cpp/adapter_demo.bin (the reference's plugin surface, src/mcl_3dl.cpp untouched) with the particles sharded
    over three contexts: MCL3DL_HIP_DEVICES=0,0,0."""
    from test_gpu_adapter import DEMO, need_binary, write_scene
    need_binary(DEMO)
    sc = make_scene(n=91, n_p=96, n_s=777, n_b=40, seed=21, label_wall=2)
    scene_f, result = str(tmp_path / "scene.bin"), str(tmp_path / "result.bin")
    write_scene(scene_f, sc, DW, 40, True, 1, 0.6)
    env = dict(os.environ, MCL3DL_HIP_DEVICES="0,0,0", MCL3DL_HIP_COLLECTIVE="host")
    proc = subprocess.run([DEMO, scene_f, result], capture_output=True, text=True, timeout=300, env=env)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    raw = np.fromfile(result, dtype=np.float32)
    n = len(sc.poses)
    w, lik, beam, quality = raw[:n], raw[n:2 * n], raw[2 * n:3 * n], raw[3 * n:4 * n]
    o = pyoracle.Oracle(oracle_kind)
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=DW)
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    o.set_beam_params(pyoracle.BeamParams(num_points=40, filter_label_max=1))
    want = o.measure_update(sc.poses, sc.weights, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, sc.odom_err,
                            0.6)
    np.testing.assert_allclose(lik, want["lik"], rtol=1e-5)
    np.testing.assert_array_equal(beam, want["beam"])
    np.testing.assert_array_equal(quality, want["quality"])
    np.testing.assert_allclose(w, want["weights"], rtol=1e-5)
