"""GPU tests of the C ABI's error behaviour and odd inputs (no exceptions across the boundary, negative
codes + mcl3dl_hip_last_error; the reference's own edge cases for empty / invalid data)."""
import ctypes as C

import numpy as np
import pytest

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_scene
from oracle import pyoracle

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fresh():
    e = capi.Engine(0)
    yield e
    e.close()


def test_measure_without_map_is_an_error(fresh):
    sc = make_scene(n=21, n_p=4, n_s=16)
    with pytest.raises(capi.EngineError, match="no map"):
        fresh.measure_batch(sc.poses, sc.scan_lik)
    # the context stays usable after an error
    fresh.set_map(sc.map_xyz, sc.map_label)
    lik, ratio, beam = fresh.measure_batch(sc.poses, sc.scan_lik)
    assert np.all(np.isfinite(lik))


def test_bad_arguments_return_codes(fresh):
    lib = fresh.lib
    assert lib.mcl3dl_hip_set_map(fresh.h, None, None, 0, 1, None) != 0
    assert b"empty map" in lib.mcl3dl_hip_last_error(fresh.h)
    assert lib.mcl3dl_hip_set_likelihood_params(fresh.h, -1.0, 0.05, 5.0) != 0
    assert lib.mcl3dl_hip_set_beam_params(fresh.h, 0.1, 0.1, 0.1, 0.0, 0.004, 0.3, 0.2, 3, 0.5, 0xFFFFFFFF, 1) != 0
    assert lib.mcl3dl_hip_set_option(fresh.h, b"no_such_option", 1.0) != 0
    assert lib.mcl3dl_hip_set_option(fresh.h, b"lik_index", 7.0) != 0
    assert lib.mcl3dl_hip_get_kernel_time(fresh.h, 99, None, None) != 0
    # null context: negative, never a crash
    assert lib.mcl3dl_hip_synchronize(None) != 0
    assert lib.mcl3dl_hip_measure_device(None, None, 0, None, None, None) != 0
    lib.mcl3dl_hip_destroy(None)
    out = C.c_void_p()
    assert lib.mcl3dl_hip_create(C.byref(out), 12345) != 0 and not out  # no such device


def test_beam_origin_out_of_range(fresh):
    sc = make_scene(n=21, n_p=4, n_s=16, n_b=8)
    fresh.set_map(sc.map_xyz, sc.map_label)
    bad = sc.scan_beam_label.copy()
    bad[3] = 5
    with pytest.raises(capi.EngineError, match="origin"):
        fresh.measure_batch(sc.poses, None, sc.scan_beam, bad, sc.origins)


def test_zero_particles_is_a_no_op(fresh):
    sc = make_scene(n=21, n_p=4, n_s=16)
    fresh.set_map(sc.map_xyz, sc.map_label)
    lik, ratio, beam = fresh.measure_batch(np.zeros((0, 7), np.float32), sc.scan_lik)
    assert len(lik) == 0


def test_non_finite_scan_points_do_not_match(fresh, oracle_kind):
    """NaN / Inf scan coordinates find no neighbour (FLANN returns nothing for them) but still count in the quality
    denominator (likelihood.cpp:136)."""
    sc = make_scene(n=41, n_p=8, n_s=64)
    fresh.set_map(sc.map_xyz, sc.map_label)
    scan = sc.scan_lik.copy()
    scan[5] = np.nan
    scan[9, 1] = np.inf
    scan[11] = 1e30
    lik, ratio, _ = fresh.measure_batch(sc.poses, scan)
    o = pyoracle.Oracle(oracle_kind)
    o.set_map(sc.map_xyz, sc.map_label)
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    wl, wq = o.likelihood_measure(sc.poses, scan)
    np.testing.assert_allclose(lik, wl, rtol=1e-5)
    np.testing.assert_array_equal(ratio, wq)
    assert np.all(np.isfinite(lik))


def test_map_replaced_and_stamp_semantics(fresh, oracle_kind):
    """A new map (new stamp) replaces every device structure; results follow the new map."""
    a = make_scene(n=41, n_p=8, n_s=64, n_b=8, seed=1)
    b = make_scene(n=61, n_p=8, n_s=64, n_b=8, seed=2)
    for stamp, sc in ((1, a), (2, b), (3, a)):
        fresh.set_map(sc.map_xyz, sc.map_label, stamp=stamp)
        fresh.set_beam_params(num_points=8)
        lik, ratio, beam = fresh.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        o = pyoracle.Oracle(oracle_kind)
        o.set_map(sc.map_xyz, sc.map_label)
        o.set_likelihood_params(pyoracle.LikelihoodParams())
        o.set_beam_params(pyoracle.BeamParams(num_points=8))
        wl, wq = o.likelihood_measure(sc.poses, sc.scan_lik)
        wb, _ = o.beam_measure(sc.poses, sc.scan_beam, sc.scan_beam_label, sc.origins)
        np.testing.assert_allclose(lik, wl, rtol=1e-5)
        np.testing.assert_array_equal(ratio, wq)
        np.testing.assert_array_equal(beam, wb)


def test_two_contexts_are_independent():
    a, b = capi.Engine(0), capi.Engine(0)
    sa = make_scene(n=41, n_p=8, n_s=64, seed=1)
    sb = make_scene(n=61, n_p=8, n_s=64, seed=2)
    a.set_map(sa.map_xyz, sa.map_label)
    b.set_map(sb.map_xyz, sb.map_label)
    la1, _, _ = a.measure_batch(sa.poses, sa.scan_lik)
    lb, _, _ = b.measure_batch(sb.poses, sb.scan_lik)
    la2, _, _ = a.measure_batch(sa.poses, sa.scan_lik)
    np.testing.assert_array_equal(la1, la2)
    assert not np.array_equal(la1, lb)
    a.close()
    b.close()


def test_automatic_strict_order_never_fails_an_update_over_memory(engine):
    """strict_order = 2 replays scans of >= strict_auto_min points in the reference's float order, at the cost of an n_s x n_p
    float buffer; when that buffer is refused (here: by the byte cap) the launch sums in fp64 like a smaller scan — within the
    reference's own rounding of the strict result — instead of failing."""
    import numpy as np
    from mcl_3dl_amd.synthetic import make_scene
    # (13 000 points: beyond the LDS rows of the per-particle kernels, which need no buffer at all — round 6)
    sc = make_scene(n=91, n_p=300, n_s=13000, seed=99)
    engine.set_map(sc.map_xyz, sc.map_label, stamp=7700, dist_weight=(1.0, 1.0, 1.0))
    engine.set_likelihood_params()
    auto_min = engine.get_option("strict_auto_min")
    try:
        engine.set_option("strict_auto_min", 12500)
        strict, ratio_s, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        before = engine.get_option("strict_auto_skipped")
        engine.set_option("strict_auto_max_bytes", 1 << 20)   # 13 000 x 304 x 4 B = 15.8 MB does not fit
        loose, ratio_l, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        assert engine.get_option("strict_auto_skipped") == before + 1
    finally:
        engine.set_option("strict_auto_min", auto_min)
        engine.set_option("strict_auto_max_bytes", 0)
    np.testing.assert_array_equal(ratio_s, ratio_l)
    np.testing.assert_allclose(loose, strict, rtol=1e-5)
    assert not np.array_equal(loose, strict)   # (the float recurrence and the fp64 tree do differ somewhere over 300 particles)


def test_a_long_update_does_not_hold_a_core(engine):
    """the polled completion word used to be spun on for the whole kernel (one core pegged for the 26 ms
    of a C5 update). The wait now spins for poll_spin_us (default 2 ms: every update up to a few thousand particles), then naps
    between looks (1/32 of the time already waited). Measured here: CPU time / wall time of the CALLING THREAD (RUSAGE_THREAD)
    over updates of 10-20 ms — the process-wide figure also counts whatever helper threads earlier tests of the session left
    behind (RCCL proxies, the HIP runtime's handlers): measured alone the two agree (19 % of a core at 11 ms; a pure spin and
    hipStreamSynchronize both hold 100 %: scripts/r05_dbg_cpushare.py, profiles/r05i_cpushare.txt)."""
    import resource
    import time

    import torch
    sc = make_scene(n=91, n_p=64, n_s=8192, seed=77)
    n_p = 400000
    rng = np.random.default_rng(1)
    poses = np.repeat(sc.poses, (n_p + 63) // 64, axis=0)[:n_p].copy()
    poses[:, :3] += rng.normal(0, 0.05, (n_p, 3)).astype(np.float32)
    engine.set_map(sc.map_xyz, sc.map_label, stamp=99100, dist_weight=(1.0, 1.0, 1.0))
    engine.set_likelihood_params()
    engine.upload_scan(sc.scan_lik)
    dev = torch.device("cuda", 0)
    d_pose = torch.from_numpy(poses).to(dev)
    d_lik, d_q = torch.zeros(n_p, device=dev), torch.zeros(n_p, device=dev)
    torch.cuda.synchronize()

    def run(reps):
        r0, t0 = resource.getrusage(resource.RUSAGE_THREAD), time.perf_counter()
        for _ in range(reps):
            engine.measure_device(d_pose, n_p, d_lik, d_q, None)
            engine.synchronize()
        r1, t1 = resource.getrusage(resource.RUSAGE_THREAD), time.perf_counter()
        return (r1.ru_utime + r1.ru_stime - r0.ru_utime - r0.ru_stime), t1 - t0

    run(2)
    # (wall-clock comparisons on a box that is not ours alone: up to three rounds, the first clean one counts — a round-6 suite
    # run saw the latency comparison miss once in six runs)
    for attempt in range(3):
        cpu, wall = run(8)
        per_update_ms = wall / 8 * 1e3
        assert per_update_ms > 6.0, "the update is too short (%.2f ms) to say anything about the napping phase" % per_update_ms
        # a pure spin for comparison (poll_spin_us far beyond the update): the same updates, (nearly) a whole core
        try:
            engine.set_option("poll_spin_us", 1e6)
            cpu_spin, wall_spin = run(4)
        finally:
            engine.set_option("poll_spin_us", 2000)
        # the nap costs the caller little latency: the napping wait is within 5 % + 0.1 ms of the spinning one
        if cpu / wall < 0.4 and cpu_spin / wall_spin > 0.8 and wall / 8 < 1.05 * wall_spin / 4 + 1e-4:
            break
    assert cpu / wall < 0.4, "caller used %.0f %% of a core over %.1f ms updates" % (100 * cpu / wall, per_update_ms)
    assert cpu_spin / wall_spin > 0.8
    assert wall / 8 < 1.05 * wall_spin / 4 + 1e-4, (wall / 8, wall_spin / 4)
