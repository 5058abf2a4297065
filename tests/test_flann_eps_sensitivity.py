"""What pcl::KdTreeFLANN's epsilon is worth ("parity unpinned at the PCL/FLANN boundary"; VERDICT round 2,
missing #5). The node searches with eps = map_grid_min / 16 (src/mcl_3dl.cpp:1328); both oracles and the GPU engine define
radiusSearch as the EXACT nearest neighbour. Here the reference's own measure() loop (oracle/_ref) runs twice on the same
scene — once on the exact stand-in, once on a restatement of FLANN's KDTreeSingleIndex with (1 + eps) pruning
(oracle/shims/pcl/kdtree/flann_single_index.h, written from the published algorithm: FLANN itself is not available) — and the
difference is MEASURED against the bound DESIGN.md section 5 derives on paper:
   eps = 0                    the restated tree is an exact search: scores and match ratios equal bit for bit
   eps = 0.1 / 16 = 0.00625   per-point squared distance at most (1 + eps) x the true one; likelihood moves by < 3.1e-3 of
                              match_weight x match_dist_min per point on paper; MEASURED: 0 (dist_weight 1,1,1) and 8.6e-6
                              relative at worst (dist_weight 1,1,5) on a 1000-point scan, no change of any match ratio
CPU only (the oracle is test infrastructure)."""
import numpy as np
import pytest

from mcl_3dl_amd.synthetic import make_scene
from oracle import pyoracle

EPS_NODE = 0.1 / 16.0   # map_grid_min / 16 with the default map_downsample of 0.1 m (src/mcl_3dl.cpp:1328)


@pytest.fixture(scope="module")
def ref():
    if not pyoracle.available("ref"):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return pyoracle.Oracle("ref")


def measure(o, sc, eps, mode, dist_weight):
    o.set_flann_epsilon_mode(mode)
    try:
        o.set_map(sc.map_xyz, sc.map_label, dist_weight=dist_weight, epsilon=eps)
        o.set_likelihood_params(pyoracle.LikelihoodParams())
        return o.likelihood_measure(sc.poses, sc.scan_lik)
    finally:
        o.set_flann_epsilon_mode(0)


@pytest.mark.parametrize("dist_weight", [(1.0, 1.0, 1.0), (1.0, 1.0, 5.0)])
def test_restated_flann_index_with_eps_zero_is_the_exact_search(ref, dist_weight):
    sc = make_scene(n=91, n_p=48, n_s=1000, n_b=0, seed=5)
    exact_l, exact_q = measure(ref, sc, 0.0, 0, dist_weight)
    # eps = 0 never reaches the tree (the stand-in only routes eps > 0 there): use the smallest positive float
    tree_l, tree_q = measure(ref, sc, np.float32(1e-30), 1, dist_weight)
    np.testing.assert_array_equal(tree_l, exact_l)
    np.testing.assert_array_equal(tree_q, exact_q)


@pytest.mark.parametrize("dist_weight", [(1.0, 1.0, 1.0), (1.0, 1.0, 5.0)])
def test_sensitivity_of_the_likelihood_to_the_nodes_epsilon(ref, dist_weight):
    sc = make_scene(n=91, n_p=64, n_s=1000, n_b=0, seed=6)
    exact_l, exact_q = measure(ref, sc, EPS_NODE, 0, dist_weight)
    approx_l, approx_q = measure(ref, sc, EPS_NODE, 1, dist_weight)
    p = pyoracle.LikelihoodParams()
    n_s = len(sc.scan_lik)
    # bound: a returned neighbour is at most sqrt(1 + eps) times as far as the true one, so a term
    # w (r - max(d, flat)) moves by at most w r (sqrt(1 + eps) - 1), and never upwards
    per_point = p.match_weight * p.match_dist_min * (np.sqrt(1.0 + EPS_NODE) - 1.0)
    diff = exact_l.astype(np.float64) - approx_l.astype(np.float64)
    assert (diff >= -1e-4 * np.abs(exact_l)).all()          # the approximate search never finds anything closer
    assert (diff <= n_s * per_point).all()
    rel = np.abs(diff) / np.maximum(np.abs(exact_l), 1e-30)
    dq = np.abs(exact_q.astype(np.float64) - approx_q.astype(np.float64)) * n_s
    print("eps = %.5f, dist_weight %s: likelihood rel. diff max %.3g mean %.3g; matched points differ by at most %d of %d"
          % (EPS_NODE, dist_weight, rel.max(), rel.mean(), int(dq.max()), n_s))
    # measured, with margin: three orders of magnitude above the 1e-5 parity gate, which is why that gate is defined
    # against the exact search and the boundary is called unpinned — and why it does not matter to localisation
    assert rel.max() < 2e-2
    assert dq.max() <= 0.02 * n_s


def test_the_probe_responds_to_a_coarse_epsilon(ref):
    """Not vacuous: with eps = 1 (prune everything farther than half the current best) the approximate search does return
    farther neighbours and scores drop visibly — so the ~1e-5 measured at the node's eps is a property of that eps."""
    sc = make_scene(n=91, n_p=32, n_s=1000, n_b=0, seed=7)
    exact_l, _ = measure(ref, sc, 1.0, 0, (1.0, 1.0, 1.0))
    approx_l, _ = measure(ref, sc, 1.0, 1, (1.0, 1.0, 1.0))
    rel = (exact_l.astype(np.float64) - approx_l) / np.maximum(exact_l, 1e-30)
    print("eps = 1: likelihood rel. diff max %.3g mean %.3g" % (rel.max(), rel.mean()))
    assert rel.max() > 1e-4 and (rel >= -1e-6).all()
