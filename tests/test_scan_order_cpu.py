"""mcl3dl_hip_scan_order_host (pure host function, no GPU): the engine's order of a likelihood scan — what a caller uses to
HOLD its sampled cloud in that order, so that strict_order = 3 (the float recurrence inside the likelihood kernel) sums in the
caller's own order (include/mcl3dl_hip.h; the drop-in classes' MCL3DL_HIP_ENGINE_ORDER=1)."""
import numpy as np

from mcl_3dl_amd import capi


def test_order_is_a_stable_permutation_and_an_ordered_cloud_stays_put():
    rng = np.random.default_rng(7)
    for n in (1, 2, 255, 256, 1000, 5000):
        scan = rng.uniform(-9.0, 9.0, (n, 3)).astype(np.float32)
        scan[n // 3:n // 3 + 3] = scan[0]                     # equal points: equal keys, the sort must keep their order
        order = capi.scan_order_host(scan)
        assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32))
        dup = np.nonzero((scan == scan[0]).all(1))[0]
        pos = np.argsort(order)[dup]                           # positions of the equal points in the engine's order
        assert np.all(np.diff(pos) > 0)
        held = np.ascontiguousarray(scan[order])
        np.testing.assert_array_equal(capi.scan_order_host(held), np.arange(n, dtype=np.uint32))


def test_order_follows_0_25_m_cells_of_the_bounding_box():
    # two clusters far apart: the engine's order never interleaves them (a work-group's 256 points are neighbours)
    rng = np.random.default_rng(3)
    a = rng.uniform(0.0, 0.2, (300, 3)).astype(np.float32)
    b = (rng.uniform(0.0, 0.2, (300, 3)) + 20.0).astype(np.float32)
    scan = np.concatenate([a, b])[rng.permutation(600)]
    order = capi.scan_order_host(scan)
    far = scan[order][:, 0] > 10.0
    assert np.count_nonzero(np.diff(far.astype(np.int8)) != 0) == 1


def test_non_finite_points_and_empty_scans():
    scan = np.array([[0, 0, 0], [np.nan, 1, 1], [1, 1, 1], [np.inf, 0, 0]], np.float32)
    order = capi.scan_order_host(scan)
    assert np.array_equal(np.sort(order), np.arange(4, dtype=np.uint32))
    assert len(capi.scan_order_host(np.zeros((0, 3), np.float32))) == 0
