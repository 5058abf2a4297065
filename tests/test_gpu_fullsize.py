"""GPU tests at BASELINE.json's full sizes (C2: 4096 particles x 16 384 points, 998 784-pt map; C3 adds 512 rays per
particle).  The oracle cannot run the whole configuration in seconds, so correctness is pinned by
  * an oracle check on a 64-particle slice of the SAME launch, and
  * size-independent properties: additivity over a split of the scan, invariance to the order of scan points and of
    particles, duplicated particles giving duplicated results, beam counts bounded by the table, weights summing to 1."""
import numpy as np
import pytest

from mcl_3dl_amd.synthetic import make_config
from oracle import pyoracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3():
    return make_config("C3")


@pytest.fixture(scope="module")
def full(engine, c3):
    sc = c3
    engine.set_map(sc.map_xyz, sc.map_label, stamp=900, dist_weight=(1.0, 1.0, 1.0))
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=len(sc.scan_beam))
    lik, ratio, beam = engine.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
    return lik, ratio, beam


def test_slice_against_oracle(full, c3, oracle_kind):
    lik, ratio, beam = full
    sc = c3
    o = pyoracle.Oracle(oracle_kind)
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=(1.0, 1.0, 1.0))
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    o.set_beam_params(pyoracle.BeamParams(num_points=len(sc.scan_beam)))
    idx = np.arange(0, len(sc.poses), len(sc.poses) // 64)[:64]
    wl, wq = o.likelihood_measure(sc.poses[idx], sc.scan_lik)
    wb, _ = o.beam_measure(sc.poses[idx], sc.scan_beam, sc.scan_beam_label, sc.origins)
    np.testing.assert_allclose(lik[idx], wl, rtol=1e-5)
    np.testing.assert_array_equal(ratio[idx], wq)
    np.testing.assert_array_equal(beam[idx], wb)


def test_additive_over_scan_split(engine, full, c3):
    """score(A u B) = score(A) + score(B); matched counts add exactly."""
    lik, ratio, _ = full
    sc = c3
    n = len(sc.scan_lik)
    cut = 7001  # ragged on purpose
    la, ra, _ = engine.measure_batch(sc.poses, sc.scan_lik[:cut])
    lb, rb, _ = engine.measure_batch(sc.poses, sc.scan_lik[cut:])
    np.testing.assert_allclose(la.astype(np.float64) + lb.astype(np.float64), lik, rtol=2e-7)
    counts = np.rint(ra.astype(np.float64) * cut) + np.rint(rb.astype(np.float64) * (n - cut))
    np.testing.assert_array_equal(np.rint(ratio.astype(np.float64) * n), counts)


def test_invariant_to_scan_and_particle_order(engine, full, c3):
    lik, ratio, beam = full
    sc = c3
    rng = np.random.default_rng(1)
    ps = rng.permutation(len(sc.scan_lik))
    pb = rng.permutation(len(sc.scan_beam))
    pp = rng.permutation(len(sc.poses))
    l2, r2, b2 = engine.measure_batch(sc.poses[pp], sc.scan_lik[ps], sc.scan_beam[pb], sc.scan_beam_label[pb],
                                      sc.origins)
    np.testing.assert_array_equal(l2, lik[pp])   # the library orders the scan itself: same terms, same sums
    np.testing.assert_array_equal(r2, ratio[pp])
    np.testing.assert_array_equal(b2, beam[pp])


def test_duplicated_particles_and_update(engine, full, c3):
    lik, ratio, beam = full
    sc = c3
    poses = np.concatenate([sc.poses[:2048], sc.poses[:2048]], 0)
    got = engine.measure_update(poses, sc.weights, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
    np.testing.assert_array_equal(got["lik"][:2048], got["lik"][2048:])
    np.testing.assert_array_equal(got["lik"][:2048], lik[:2048])
    np.testing.assert_array_equal(got["beam"][:2048], beam[:2048])
    np.testing.assert_array_equal(got["weights"][:2048], got["weights"][2048:])
    np.testing.assert_allclose(got["weights"].sum(dtype=np.float64), 1.0, rtol=1e-6)
    w = got["weights"].astype(np.float64)
    np.testing.assert_allclose(got["entropy"], -(w[w > 0] * np.log(w[w > 0])).sum(), rtol=1e-5)
    assert got["match_ratio_min"] == ratio[:2048].min() and got["match_ratio_max"] == ratio[:2048].max()
    # beam scores are table entries b^k clamped from below
    b = np.float32(np.float64(np.float32(0.2)) ** (1.0 / np.float32(len(sc.scan_beam))))
    assert np.all(beam >= np.float32(0.2)) and np.all(beam <= 1.0)
    assert np.isin(beam, np.maximum(np.cumprod(np.concatenate([[np.float32(1)], np.full(512, b, np.float32)]),
                                               dtype=np.float32), np.float32(0.2))).all()


def test_strict_order_slice_is_bit_identical(engine, c3, oracle_kind):
    """C2-size launch in "strict_order" mode: a 64-particle slice equals the reference bit for bit."""
    sc = c3
    engine.set_map(sc.map_xyz, sc.map_label, stamp=901, dist_weight=(1.0, 1.0, 1.0))
    engine.set_likelihood_params()
    try:
        engine.set_option("strict_order", 1)
        lik, ratio, _ = engine.measure_batch(sc.poses, sc.scan_lik)
    finally:
        engine.set_option("strict_order", 2)
    o = pyoracle.Oracle(oracle_kind)
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=(1.0, 1.0, 1.0))
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    idx = np.arange(0, len(sc.poses), len(sc.poses) // 64)[:64]
    wl, wq = o.likelihood_measure(sc.poses[idx], sc.scan_lik)
    np.testing.assert_array_equal(lik[idx], wl)
    np.testing.assert_array_equal(ratio[idx], wq)


@pytest.mark.parametrize("n_p,n_s,group", [(8200, 2100, 16), (4203, 1100, 8), (1100, 1300, 4)])
def test_strict_order_with_more_particle_groups_than_cus(engine, oracle_kind, n_p, n_s, group):
    """More than 256 particle groups of every size the tiled kernel uses (ragged last group, odd group count): particles of
    the first groups, of the last ones and a sample in between equal the reference bit for bit."""
    sc = make_config("C2", n_p=n_p, n_s=n_s)
    engine.set_map(sc.map_xyz, sc.map_label, stamp=902, dist_weight=(1.0, 1.0, 1.0))
    engine.set_likelihood_params()
    try:
        engine.set_option("strict_order", 1)
        engine.set_option("lik_group", group)
        lik, ratio, _ = engine.measure_batch(sc.poses, sc.scan_lik)
    finally:
        engine.set_option("strict_order", 2)
        engine.set_option("lik_group", 0)
    assert (n_p + group - 1) // group > 256
    o = pyoracle.Oracle(oracle_kind)
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=(1.0, 1.0, 1.0))
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    idx = np.unique(np.concatenate([np.arange(0, n_p, n_p // 40), np.arange(group - 3, group + 5),
                                    np.arange(n_p - 2 * group - 1, n_p)]))
    wl, wq = o.likelihood_measure(sc.poses[idx], sc.scan_lik)
    np.testing.assert_array_equal(lik[idx], wl)
    np.testing.assert_array_equal(ratio[idx], wq)


def test_c2_on_a_map_of_centroids(engine, oracle_kind):
    """C2's sizes on the map whose points are displaced inside their voxels (a voxel-filtered real map: a quarter of the
    candidate voxels overflow their record): the default path — 64-byte packed records, overflow rounds queued per wavefront —
    against the reference on a slice, against the immediate rounds and against the cell scan bit for bit."""
    sc = make_config("C2", map_jitter=0.045)
    engine.set_likelihood_params()
    try:
        engine.set_map(sc.map_xyz, sc.map_label, stamp=950, dist_weight=(1.0, 1.0, 1.0))
        lik, ratio, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        st = engine.index_stats()
        assert st["deferred_overflow"] == 1 and st["packed_words"] == 1 and st["record_parts"] == 4
        assert st["voxels_with_overflow"] > 0.2 * st["voxels_with_candidates"]
        o = pyoracle.Oracle(oracle_kind)
        o.set_map(sc.map_xyz, sc.map_label, dist_weight=(1.0, 1.0, 1.0))
        o.set_likelihood_params(pyoracle.LikelihoodParams())
        idx = np.arange(0, len(sc.poses), len(sc.poses) // 32)[:32]
        wl, wq = o.likelihood_measure(sc.poses[idx], sc.scan_lik)
        np.testing.assert_allclose(lik[idx], wl, rtol=1e-5)
        np.testing.assert_array_equal(ratio[idx], wq)
        sub = sc.poses[:512]
        engine.set_option("lik_defer", 0)
        engine.set_option("cand_record_parts", 4)
        engine.set_map(sc.map_xyz, sc.map_label, stamp=951, dist_weight=(1.0, 1.0, 1.0))
        l0, r0, _ = engine.measure_batch(sub, sc.scan_lik)
        engine.set_option("lik_index", 0)
        engine.set_map(sc.map_xyz, sc.map_label, stamp=952, dist_weight=(1.0, 1.0, 1.0))
        lc, rc, _ = engine.measure_batch(sub, sc.scan_lik)
    finally:
        engine.set_option("lik_index", 2)
        engine.set_option("lik_defer", 1)
        engine.set_option("cand_record_parts", 0)
    # (the particle groups of a 512-particle launch are the first 32 groups of the 4096-particle launch: same sums)
    np.testing.assert_array_equal(l0, lik[:512])
    np.testing.assert_array_equal(r0, ratio[:512])
    np.testing.assert_array_equal(lc, lik[:512])
    np.testing.assert_array_equal(rc, ratio[:512])
