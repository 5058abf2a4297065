"""mcl3dl_hip_measure_batch_begin / _wait / _end (include/mcl3dl_hip.h): the batch of the two LiDAR models delivered in particle
order while the GPU works on the later particles — what lets the reference's per-particle loop (include/mcl_3dl/pf.h:255-260
with the lambda of src/mcl_3dl.cpp:399-426) start on the first slice. The results are those of mcl3dl_hip_measure_batch bit
for bit (a particle's likelihood, match ratio and beam score do not depend on which particles share its launch), whatever
the slice, whether the output arrays are pageable or page-locked, and whatever else is called on the context meanwhile."""
import numpy as np
import pytest

from mcl_3dl_amd.capi import EngineError, Group
from mcl_3dl_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    return make_scene(n=61, n_p=4096, n_s=4096, n_b=256, seed=777)


@pytest.fixture(scope="module")
def configured(engine, scene):
    sc = scene
    engine.set_map(sc.map_xyz, sc.map_label, stamp=9101, dist_weight=(1.0, 1.0, 5.0))
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=256)
    return engine


def whole(engine, sc, n_p, n_s, n_b):
    return engine.measure_batch(sc.poses[:n_p], sc.scan_lik[:n_s], sc.scan_beam[:n_b] if n_b else None,
                                sc.scan_beam_label[:n_b] if n_b else None, sc.origins)


def begin(engine, sc, n_p, n_s, n_b, slice_particles, out=None):
    return engine.measure_batch_begin(sc.poses[:n_p], sc.scan_lik[:n_s], sc.scan_beam[:n_b] if n_b else None,
                                      sc.scan_beam_label[:n_b] if n_b else None, sc.origins, slice_particles=slice_particles,
                                      out=out)


@pytest.mark.parametrize("n_p,n_s,n_b,slice_particles", [(4096, 4096, 256, 0), (4096, 4096, 0, 1024), (3000, 2048, 64, 512),
                                                         (1025, 1000, 32, 256), (2048, 96, 3, 100), (4096, 4096, 256, 4096),
                                                         (700, 300, 16, 0), (1, 500, 8, 0), (4096, 0, 128, 1000)])
def test_slices_arrive_in_particle_order_with_the_results_of_the_whole_batch(configured, scene, n_p, n_s, n_b, slice_particles):
    engine, sc = configured, scene
    ref = whole(engine, sc, n_p, n_s, n_b)
    run0 = engine.get_option("batch_slices_run")
    lik, ratio, beam = begin(engine, sc, n_p, n_s, n_b, slice_particles)
    ready, waits = 0, 0
    for i in range(n_p):
        if i >= ready:
            now = engine.measure_batch_wait(i)
            assert now > i and now >= ready and now <= n_p
            ready, waits = now, waits + 1
            # everything up to `ready` is final already
            np.testing.assert_array_equal(lik[:ready], ref[0][:ready])
            np.testing.assert_array_equal(ratio[:ready], ref[1][:ready])
            np.testing.assert_array_equal(beam[:ready], ref[2][:ready])
    engine.measure_batch_end()
    for a, b in zip((lik, ratio, beam), ref):
        np.testing.assert_array_equal(a, b)
    ran = engine.get_option("batch_slices_run") - run0
    eff = slice_particles if slice_particles else (max(512, -(-((n_p + 3) // 4) // 64) * 64) if n_p >= 1024 else n_p)
    eff = -(-eff // 16) * 16
    expected = -(-n_p // eff) if eff < n_p else 0
    assert ran == expected, (ran, expected)
    assert waits <= max(expected, 1)


def test_page_locked_outputs_are_written_in_place(configured, scene):
    engine, sc = configured, scene
    n_p = 4096
    ref = whole(engine, sc, n_p, 4096, 256)
    out = tuple(engine.host_array(n_p) for _ in range(3))
    try:
        for a in out:
            a[:] = -1.0
        begin(engine, sc, n_p, 4096, 256, 1024, out=out)
        assert engine.measure_batch_wait(0) >= 1024
        np.testing.assert_array_equal(out[0][:1024], ref[0][:1024])
        assert engine.measure_batch_wait(n_p - 1) == n_p
        engine.measure_batch_end()
        for a, b in zip(out, ref):
            np.testing.assert_array_equal(a, b)
    finally:
        for a in out:
            engine.host_free(a)


def test_end_without_wait_and_calls_in_between(configured, scene):
    engine, sc = configured, scene
    ref = whole(engine, sc, 4096, 4096, 256)
    # _end alone delivers everything
    got = begin(engine, sc, 4096, 4096, 256, 512)
    engine.measure_batch_end()
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
    # a synchronising call in the middle finishes the batch first; _wait then reports it complete
    got = begin(engine, sc, 4096, 4096, 256, 512)
    other = whole(engine, sc, 100, 300, 0)
    assert engine.measure_batch_wait(5) == 4096
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
    engine.measure_batch_end()
    np.testing.assert_array_equal(other[0], whole(engine, sc, 100, 300, 0)[0])
    # a second _begin ends the first
    first = begin(engine, sc, 4096, 4096, 256, 512)
    second = begin(engine, sc, 2048, 4096, 0, 512)
    for a, b in zip(first, ref):
        np.testing.assert_array_equal(a, b)
    engine.measure_batch_end()
    np.testing.assert_array_equal(second[0], ref[0][:2048])
    # _end with no batch open is a no-op
    engine.measure_batch_end()


def test_a_wrong_index_is_refused_without_abandoning_the_batch(configured, scene):
    engine, sc = configured, scene
    ref = whole(engine, sc, 2048, 2048, 0)
    got = begin(engine, sc, 2048, 2048, 0, 512)
    with pytest.raises(EngineError, match="not part of the batch"):
        engine.measure_batch_wait(2048)
    # another call that fails its argument check does not abandon the batch either
    with pytest.raises(EngineError, match="bad arguments"):
        engine.match_split(sc.true_pose, sc.scan_lik[:10], unmatch_dist=0.0)
    assert engine.measure_batch_wait(2047) == 2048
    engine.measure_batch_end()
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])


def test_uploaded_poses_and_the_update_after_a_batch(configured, scene):
    """pose = NULL uses the uploaded set (what the drop-in classes do); a measure_update after the batch sees a clean context."""
    engine, sc = configured, scene
    n_p = 2048
    ref = whole(engine, sc, n_p, 4096, 256)
    engine.upload_poses(sc.poses[:n_p])
    got = engine.measure_batch_begin(None, sc.scan_lik[:4096], sc.scan_beam[:256], sc.scan_beam_label[:256], sc.origins,
                                     slice_particles=512)
    engine.measure_batch_end()
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
    w0 = np.full(n_p, 1.0 / n_p, np.float32)
    a = engine.measure_update(sc.poses[:n_p], w0, sc.scan_lik[:4096], sc.scan_beam[:256], sc.scan_beam_label[:256], sc.origins)
    np.testing.assert_array_equal(a["lik"], ref[0])
    np.testing.assert_array_equal(a["beam"], ref[2])


@pytest.mark.parametrize("devices,collective", [((0,), None), ((0, 0), "host"), ((0, 0, 0), "host")])
def test_group_form(scene, devices, collective):
    """One device: the context's slices. A sharded group (round 5): every rank runs its shard as a progressive batch of its
    own, the shards' slices arrive in the caller's arrays while the other ranks still compute, _wait is answered by the owner
    of the particle. Results are those of measure_batch bit for bit."""
    import time
    sc = scene
    g = Group(devices, collective=collective)
    N = len(devices)
    try:
        g.set_map(sc.map_xyz, sc.map_label, stamp=9102, dist_weight=(1.0, 1.0, 5.0))
        g.set_likelihood_params()
        g.set_beam_params(num_points=256)
        n_p = 3000 * N if N > 1 else 3000
        poses = np.ascontiguousarray(np.tile(sc.poses, (n_p // len(sc.poses) + 1, 1))[:n_p])
        poses[:, :3] += np.random.default_rng(1).normal(0, 0.02, (n_p, 3)).astype(np.float32)
        ref = g.measure_batch(poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        # (one untimed batch in slices first: a slice of 1000 particles may take another kernel than the shard of 3000 did —
        # caller-order rows instead of the replay, same bits — and the first launch of a kernel loads its code object)
        g.measure_batch_begin(poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, slice_particles=1000)
        g.measure_batch_end()
        # (the timestamps are wall clock on a box that is not ours alone, and the whole batch is a few hundred microseconds long
        # since round 6: up to three rounds, every one checked for its results, the first with clean timestamps counts)
        for attempt in range(3):
            t0 = time.perf_counter()
            got = g.measure_batch_begin(poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, slice_particles=1000)
            t_begin = time.perf_counter()
            n = g.measure_batch_wait(0)
            t_first = time.perf_counter()
            assert 1000 <= n < n_p                      # the first slice of the first shard, not the batch
            np.testing.assert_array_equal(got[0][:n], ref[0][:n])
            if N > 1:
                # a particle of the LAST shard: everything in front of it has arrived by then
                m = g.measure_batch_wait(n_p - 2999)
                assert m > n_p - 2999
                np.testing.assert_array_equal(got[0][:m], ref[0][:m])
                np.testing.assert_array_equal(got[2][:m], ref[2][:m])
            assert g.measure_batch_wait(n_p - 1) == n_p
            t_last = time.perf_counter()
            with pytest.raises(EngineError, match="not part of the batch"):
                g.measure_batch_wait(n_p)
            g.measure_batch_end()
            for a, b in zip(got, ref):
                np.testing.assert_array_equal(a, b)
            if (t_first - t0) < 0.8 * (t_last - t0):
                break
        # the first _wait returned before the last results arrived (timestamps; the batch is ~N x 3 slices long)
        assert (t_first - t0) < 0.8 * (t_last - t0), (t_begin - t0, t_first - t0, t_last - t0)
        # a new _begin ends an open batch; fewer particles than devices
        g.measure_batch_begin(poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, slice_particles=1000)
        few = g.measure_batch_begin(poses[:2], sc.scan_lik[:500])
        assert g.measure_batch_wait(1) == 2
        g.measure_batch_end()
        np.testing.assert_array_equal(few[0], g.measure_batch(poses[:2], sc.scan_lik[:500])[0])
    finally:
        g.close()
