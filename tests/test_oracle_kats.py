"""CPU tests (no GPU): pin the oracles to the reference's own golden vectors and to each other.

`kind` runs over "port" (oracle/libmcl3dl_oracle.so, the plain-C restatement — always built) and "ref"
(oracle/_ref/libmcl3dl_ref.so, the real reference sources — present wherever /root/reference was available
to build()).
"""
import numpy as np
import pytest

import kats
from mcl_3dl_amd.synthetic import make_scene
from oracle import pyoracle

KINDS = [k for k in ("port", "ref") if pyoracle.available(k)]


def test_some_oracle_is_built():
    assert "port" in KINDS, "oracle/libmcl3dl_oracle.so is not built: run __graft_entry__.build()"


def fresh(kind, map_xyz, chunk=(10.0, 1.0), dist_weight=None, label=None):
    o = pyoracle.Oracle(kind, chunk[0], chunk[1])
    o.set_map(map_xyz, label, dist_weight=dist_weight)
    return o


def trace(o, caster, begin, end):
    return o.dda_waypoints(caster["map_grid"], caster["dda_grid_size"], caster["ray_angle_half"],
                           caster["hit_tolerance"], begin, end)


@pytest.mark.parametrize("kind", KINDS)
def test_dda_waypoints(kind):
    """test/src/test_raycast_dda.cpp:185-244: exact voxel-centre sequences, tolerance 1e-6."""
    o = fresh(kind, kats.WAYPOINT_MAP)
    for case in kats.WAYPOINT_CASES:
        wp, collided, hit, n = trace(o, kats.WAYPOINT_CASTER, case["begin"], case["end"])
        assert collided == case["collision"], case["name"]
        assert n == len(case["expected"]), case["name"]
        np.testing.assert_allclose(wp, np.array(case["expected"], np.float32), atol=1e-6, err_msg=case["name"])
        assert hit == 0


@pytest.mark.parametrize("kind", KINDS)
def test_dda_intersection(kind):
    """test/src/test_raycast_dda.cpp:246-286: hit, and pass-through-voxel-without-hit."""
    o = fresh(kind, kats.INTERSECTION_MAP)
    for case in kats.INTERSECTION_CASES:
        wp, collided, hit, n = trace(o, kats.INTERSECTION_CASTER, case["begin"], case["end"])
        assert collided == case["collision"], case["name"]
        assert n == len(case["expected"]), case["name"]
        np.testing.assert_allclose(wp, np.array(case["expected"], np.float32), atol=1e-6, err_msg=case["name"])


@pytest.mark.parametrize("kind", KINDS)
def test_dda_collision_sweeps(kind):
    """test/src/test_raycast_dda.cpp:40-104."""
    o = fresh(kind, kats.collision_wall_map())
    for begin, end, target in kats.collision_rays_must_hit():
        wp, collided, hit, n = trace(o, kats.COLLISION_CASTER, begin, end)
        assert collided
        assert np.linalg.norm(wp[-1] - np.array(target, np.float32)) <= 0.2
    for begin, end in kats.collision_rays_must_miss():
        wp, collided, hit, n = o.dda_waypoints(kats.COLLISION_CASTER["map_grid"], 0.1, 0.5,
                                               kats.COLLISION_CASTER["hit_tolerance"], begin, end,
                                               stop_at_collision=False)
        assert not collided


@pytest.mark.parametrize("kind", KINDS)
def test_dda_collision_tolerance(kind):
    """test/src/test_raycast_dda.cpp:106-155."""
    o = fresh(kind, kats.tolerance_wall_map())
    for case in kats.TOLERANCE_CASES:
        wp, collided, hit, n = trace(o, case["caster"], case["begin"], case["end"])
        assert collided == case["collision"]


@pytest.mark.parametrize("kind", KINDS)
def test_chunked_kdtree_radius_search(kind):
    """test/src/test_chunked_kdtree.cpp:38-88: nearest index incl. across chunk borders (chunk 1.0, margin 0.3)."""
    o = fresh(kind, kats.KDTREE_MAP, chunk=kats.KDTREE_CHUNK)
    found, idx, sq = o.radius_search(kats.KDTREE_QUERIES, kats.KDTREE_RADIUS)
    assert found.tolist() == [1] * 6
    assert idx.tolist() == kats.KDTREE_EXPECTED


@pytest.mark.parametrize("kind", KINDS)
def test_quat_rotation_table(kind):
    """test/src/test_quat.cpp:234-273 and the associativity property :275-292."""
    o = pyoracle.Oracle(kind)
    for j, (axis, ang) in enumerate(kats.QUAT_ROTATIONS):
        q = kats.quat_axis_angle(axis, ang)
        for i, v in enumerate(kats.QUAT_VECS):
            np.testing.assert_allclose(o.quat_rotate(q, v), np.array(kats.QUAT_ANSWERS[j][i], np.float32), atol=1e-6)
    rng = np.random.default_rng(0)
    for _ in range(20):
        a = rng.normal(size=4).astype(np.float32)
        a /= np.linalg.norm(a)
        v = np.array([1.0, 2.0, 3.0], np.float32)
        # |q v q*| = |v| for unit q
        assert abs(np.linalg.norm(o.quat_rotate(a, v)) - np.linalg.norm(v)) < 1e-5


@pytest.mark.parametrize("kind", KINDS)
def test_transform_matches_float64_rotation(kind):
    """State6DOF::transform (state_6dof.h:214-225) against an independent float64 rotation-matrix evaluation."""
    from mcl_3dl_amd.synthetic import quat_to_matrix
    o = pyoracle.Oracle(kind)
    rng = np.random.default_rng(1)
    pts = rng.uniform(-10, 10, (100, 3)).astype(np.float32)
    for scale in (1.0, 0.5, 3.0):  # transform() normalises the quaternion first
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        pose = np.concatenate([[1.0, -2.0, 0.5], q * scale]).astype(np.float32)
        got = o.transform(pose, pts)
        want = pts.astype(np.float64) @ quat_to_matrix(pose[3:].astype(np.float64) / np.linalg.norm(pose[3:].astype(np.float64))).T + pose[:3]
        np.testing.assert_allclose(got, want, atol=2e-5)


@pytest.mark.parametrize("kind", KINDS)
def test_pf_entropy(kind):
    """test/src/test_pf.cpp:330-391."""
    o = pyoracle.Oracle(kind)
    w0 = np.full(10, 0.1, np.float32)
    for case in kats.PF_ENTROPY_CASES:
        w, ent, restored = o.pf_measure(w0, np.array(case["lik"], np.float32))
        assert not restored
        assert abs(ent - case["entropy"]) <= case["tol"]
        assert abs(w.sum() - 1.0) < 1e-6
    e = [o.pf_measure(w0, np.array(lk, np.float32))[1] for lk in kats.PF_ENTROPY_ORDER]
    assert e[1] > e[0]
    # no particle alive -> weights restored (pf.h:274-278)
    w, ent, restored = o.pf_measure(w0, np.zeros(10, np.float32))
    assert restored and np.array_equal(w, w0)


@pytest.mark.parametrize("kind", KINDS)
def test_nearest_neighbour_against_scipy(kind):
    """Independent third check of the exact-NN definition: scipy cKDTree on the rescaled coordinates (float64)."""
    from scipy.spatial import cKDTree
    sc = make_scene(n=41, n_p=4, n_s=500)
    for dw in ((1.0, 1.0, 1.0), (1.0, 1.0, 5.0)):
        o = fresh(kind, sc.map_xyz, chunk=(20.0, 0.4), dist_weight=dw)
        q = o.transform(sc.poses[0], sc.scan_lik)
        found, idx, sq = o.radius_search(q, 0.2)
        w = np.array(dw, np.float32)
        tree = cKDTree((sc.map_xyz * w).astype(np.float64))
        d, i = tree.query((q * w).astype(np.float64), k=1)
        clear = np.abs(d - 0.2) > 1e-5
        assert np.array_equal(found[clear] == 1, d[clear] < 0.2)
        ok = (found == 1) & clear
        np.testing.assert_allclose(np.sqrt(sq[ok]), d[ok], rtol=1e-4, atol=1e-6)
        assert ok.sum() > 100


@pytest.mark.skipif(len(KINDS) < 2, reason="oracle/_ref not built here (needs /root/reference at build time)")
@pytest.mark.parametrize("dist_weight", [(1.0, 1.0, 1.0), (1.0, 1.0, 5.0), None])
def test_port_equals_reference_bit_for_bit(dist_weight):
    """The C restatement against the reference sources themselves on a seeded scene: every output identical."""
    sc = make_scene(n=61, n_p=24, n_s=400, n_b=48, label_wall=2)
    outs = {}
    for kind in ("ref", "port"):
        o = fresh(kind, sc.map_xyz, chunk=(20.0, 0.4), dist_weight=dist_weight, label=sc.map_label)
        o.set_likelihood_params(pyoracle.LikelihoodParams())
        res = {}
        for use_dda in (True, False):
            for short_only in (True, False):
                o.set_beam_params(pyoracle.BeamParams(num_points=48, use_raycast_using_dda=use_dda,
                                                      add_penalty_short_only_mode=short_only, filter_label_max=1))
                res[("beam", use_dda, short_only)] = o.beam_measure(sc.poses, sc.scan_beam, sc.scan_beam_label, sc.origins)
        o.set_beam_params(pyoracle.BeamParams(num_points=48))
        res["lik"] = o.likelihood_measure(sc.poses, sc.scan_lik)
        rng = np.random.default_rng(5)
        begin = rng.uniform(-3.2, 3.2, (2000, 3)).astype(np.float32)
        end = (begin + rng.normal(0, 1.5, (2000, 3))).astype(np.float32)
        res["status"] = o.beam_status(begin, end)
        upd = o.measure_update(sc.poses, sc.weights, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins,
                               sc.odom_err, 0.7)
        res["upd"] = (upd["weights"], upd["lik"], upd["beam"], upd["quality"],
                      np.array([upd["entropy"], upd["match_ratio_min"], upd["match_ratio_max"]], np.float32))
        outs[kind] = res
    for key in outs["ref"]:
        for a, b in zip(outs["ref"][key], outs["port"][key]):
            np.testing.assert_array_equal(a, b, err_msg=str(key))


@pytest.mark.parametrize("kind", KINDS)
def test_beam_wall_fixture_shape(kind):
    """test/src/test_beam_likelihood.cpp:81-210 prints (does not assert) this sweep; its qualitative shape is:
    SHORT in front of the wall, HIT within hit_range of it, LONG behind."""
    raw_pc, pc_map = kats.beam_wall_fixture()
    o = fresh(kind, pc_map, chunk=(10.0, 1.0))
    o.set_beam_params(pyoracle.BeamParams(map_grid_x=0.1, map_grid_y=0.1, map_grid_z=0.1, num_points=len(raw_pc),
                                          hit_range=0.4, dda_grid_size=0.1, clip_z_min=-0.3, clip_z_max=4.1))
    xs = (0.1 * np.arange(-50, 50)).astype(np.float32)
    end = np.stack([xs, np.zeros_like(xs), np.zeros_like(xs)], 1)
    st, hit = o.beam_status(np.zeros_like(end), end)
    assert np.all(st[xs > 2.45] == 0)                     # beam ends well behind the wall -> SHORT
    assert np.all(st[(xs > 1.75) & (xs < 2.25)] == 1)     # ends within hit_range of the wall -> HIT
    assert np.all(st[xs < 1.4] == 2)                      # ends in front of it -> LONG


@pytest.mark.parametrize("kind", KINDS)
def test_expectation_and_covariance_against_float64(kind):
    """pf::expectationBiased / max / covariance (pf.h:280-390, state_6dof.h:162-184,316-355) against an independent
    float64 evaluation of the same definitions."""
    from mcl_3dl_amd.synthetic import quat_to_matrix
    sc = make_scene(n=41, n_p=400, n_s=4, seed=77, sigma_rpy=(0.05, 0.05, 0.3))
    rng = np.random.default_rng(9)
    w = rng.uniform(0, 1, 400).astype(np.float32)
    w /= w.sum()
    bias = rng.uniform(0.1, 1, 400).astype(np.float32)
    o = pyoracle.Oracle(kind)
    mean, im, ib = o.expectation(sc.poses, w, bias)
    wb = w.astype(np.float64) * bias
    np.testing.assert_allclose(mean[:3], (sc.poses[:, :3] * wb[:, None]).sum(0) / wb.sum(), rtol=1e-5)
    assert im == int(np.argmax(w)) and ib == int(np.argmax(w * bias))
    front = sum(wb[i] * quat_to_matrix(sc.poses[i, 3:])[:, 0] for i in range(400))
    got_front = quat_to_matrix(mean[3:])[:, 0]
    assert np.dot(front / np.linalg.norm(front), got_front) > 1 - 1e-6
    cov, e = o.covariance(sc.poses, w)
    d = sc.poses[:, :3].astype(np.float64) - e[:3]
    want_pos = (d[:, :, None] * d[:, None, :] * w[:, None, None].astype(np.float64)).sum(0) / w.sum(dtype=np.float64)
    np.testing.assert_allclose(cov[:3, :3], want_pos, rtol=1e-3, atol=1e-8)
    np.testing.assert_array_equal(cov, cov.T)
    assert np.all(np.diag(cov) > 0)


@pytest.mark.skipif(len(KINDS) < 2, reason="oracle/_ref not built here")
def test_port_expectation_covariance_equal_reference():
    sc = make_scene(n=41, n_p=777, n_s=4, seed=12)
    rng = np.random.default_rng(2)
    w = (rng.uniform(0, 1, 777) ** 3).astype(np.float32)
    w[::13] = 0
    w /= w.sum()
    bias = rng.uniform(1e-6, 1, 777).astype(np.float32)
    a, b = pyoracle.Oracle("ref"), pyoracle.Oracle("port")
    for bb in (None, bias):
        ma, ia, ja = a.expectation(sc.poses, w, bb)
        mb, ib, jb = b.expectation(sc.poses, w, bb)
        np.testing.assert_array_equal(ma, mb)
        assert (ia, ja) == (ib, jb)
    ca, ea = a.covariance(sc.poses, w)
    cb, eb = b.covariance(sc.poses, w)
    np.testing.assert_array_equal(ca, cb)
    np.testing.assert_array_equal(ea, eb)
