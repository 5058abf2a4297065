"""GPU tests (-m gpu): the HIP path against the reference's own golden vectors (tests/kats.py) and against the committed
outputs of the real reference (tests/golden/*.npz) — no oracle involved, so these also hold on a box that has neither
/root/reference nor oracle/_ref."""
import os

import numpy as np
import pytest

import golden_util
import kats

pytestmark = pytest.mark.gpu
IDENTITY = np.array([[0, 0, 0, 0, 0, 0, 1]], np.float32)


def set_caster(engine, caster, **kw):
    engine.set_beam_params(map_grid=caster["map_grid"], dda_grid_size=caster["dda_grid_size"],
                           ray_angle_half=caster["ray_angle_half"], hit_range=caster["hit_tolerance"], **kw)


@pytest.mark.parametrize("which", ["waypoints", "intersection"])
def test_dda_waypoint_kats(engine, which):
    """test/src/test_raycast_dda.cpp:185-286: exact voxel-centre sequences (tol 1e-6), hit / pass-through."""
    m, caster, cases = ((kats.WAYPOINT_MAP, kats.WAYPOINT_CASTER, kats.WAYPOINT_CASES) if which == "waypoints" else
                        (kats.INTERSECTION_MAP, kats.INTERSECTION_CASTER, kats.INTERSECTION_CASES))
    engine.set_map(m, None, stamp=101 if which == "waypoints" else 102, dist_weight=None)
    set_caster(engine, caster)
    for case in cases:
        wp, collided, hit, n = engine.dda_trace(case["begin"], case["end"])
        assert collided == case["collision"], case["name"]
        assert n == len(case["expected"]), case["name"]
        np.testing.assert_allclose(wp, np.array(case["expected"], np.float32), atol=1e-6, err_msg=case["name"])
        assert hit == (0 if case["collision"] else -1)


def test_dda_collision_sweeps(engine):
    """test/src/test_raycast_dda.cpp:40-104."""
    engine.set_map(kats.collision_wall_map(), None, stamp=103, dist_weight=None)
    set_caster(engine, kats.COLLISION_CASTER)
    for begin, end, target in kats.collision_rays_must_hit():
        wp, collided, hit, n = engine.dda_trace(begin, end)
        assert collided
        assert np.linalg.norm(wp[-1] - np.array(target, np.float32)) <= 0.2
    rays = kats.collision_rays_must_miss()
    st, hit = engine.beam_status(np.array([r[0] for r in rays], np.float32), np.array([r[1] for r in rays], np.float32))
    assert np.all(st == 2) and np.all(hit == -1)  # no collision anywhere along the ray -> LONG


def test_dda_collision_tolerance(engine):
    """test/src/test_raycast_dda.cpp:106-155."""
    engine.set_map(kats.tolerance_wall_map(), None, stamp=104, dist_weight=None)
    for case in kats.TOLERANCE_CASES:
        set_caster(engine, case["caster"])
        wp, collided, hit, n = engine.dda_trace(case["begin"], case["end"])
        assert collided == case["collision"]


def test_chunked_kdtree_kat_through_the_likelihood_score(engine):
    """test/src/test_chunked_kdtree.cpp:38-88 pins the nearest INDEX; the GPU path exposes the nearest DISTANCE through
    the score (r - max(d, flat)) * weight of a one-point scan under the identity pose."""
    engine.set_map(kats.KDTREE_MAP, None, stamp=105, dist_weight=None)
    r = np.float32(kats.KDTREE_RADIUS)
    engine.set_likelihood_params(match_dist_min=float(r), match_dist_flat=0.0, match_weight=1.0)
    for q, want_idx in zip(kats.KDTREE_QUERIES, kats.KDTREE_EXPECTED):
        lik, ratio, _ = engine.measure_batch(IDENTITY, q[None, :])
        d = kats.KDTREE_MAP[want_idx] - q
        d2 = np.float32(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1])) + np.float32(d[2] * d[2])
        assert ratio[0] == 1.0
        assert lik[0] == np.float32(r - np.sqrt(np.float32(d2)))
        others = [np.linalg.norm(kats.KDTREE_MAP[j] - q) for j in range(len(kats.KDTREE_MAP)) if j != want_idx]
        assert np.sqrt(d2) <= min(others) + 1e-7  # (0,-0.05,0) is equidistant from #3 and #4 in float
    engine.set_likelihood_params()


def test_chunked_kdtree_kat_indices(engine):
    """test/src/test_chunked_kdtree.cpp:38-88 directly: the nearest INDEX for each of the six queries."""
    engine.set_map(kats.KDTREE_MAP, None, stamp=107, dist_weight=None)
    idx, sq = engine.radius_search(kats.KDTREE_QUERIES, kats.KDTREE_RADIUS)
    assert idx.tolist() == kats.KDTREE_EXPECTED


def test_radius_search_against_oracle(engine, oracle_kind):
    """ChunkedKdtree::radiusSearch for several radii (likelihood 0.2, unmatch_output_dist 0.5, a large 1.3) and both
    metrics: found flag, map index and squared distance identical to the oracle."""
    from mcl_3dl_amd.synthetic import make_scene
    from oracle import pyoracle
    sc = make_scene(n=41, n_p=4, n_s=600, seed=9)
    rng = np.random.default_rng(3)
    for dw in (None, (1.0, 1.0, 5.0)):
        engine.set_map(sc.map_xyz, sc.map_label, stamp=108, dist_weight=dw)
        o = pyoracle.Oracle(oracle_kind, 20.0, 1.5)
        o.set_map(sc.map_xyz, sc.map_label, dist_weight=dw)
        q = o.transform(sc.poses[0], sc.scan_lik)
        q = np.concatenate([q, rng.uniform(-3, 3, (400, 3)).astype(np.float32), [[np.nan, 0, 0], [1e6, 0, 0]]], 0).astype(np.float32)
        for radius in (0.2, 0.5, 1.3):
            idx, sq = engine.radius_search(q, radius)
            found, widx, wsq = o.radius_search(q, radius)
            np.testing.assert_array_equal(idx >= 0, found == 1)
            np.testing.assert_array_equal(idx, widx)
            np.testing.assert_array_equal(sq, wsq)
            assert (found == 1).sum() > 100 and (found == 0).sum() > 1


def test_quat_rotation_table_through_the_transform(engine):
    """test/src/test_quat.cpp:234-273: r * v, observed as the position of the nearest map point found."""
    # a map holding the 6 axis unit points; a scan point v under pose (0, r) must land on the expected one exactly
    axis_pts = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    engine.set_map(axis_pts, None, stamp=106, dist_weight=None)
    engine.set_likelihood_params(match_dist_min=0.05, match_dist_flat=0.0, match_weight=1.0)
    for j, (axis, ang) in enumerate(kats.QUAT_ROTATIONS):
        q = kats.quat_axis_angle(axis, ang)
        pose = np.concatenate([[0, 0, 0], q]).astype(np.float32)[None, :]
        for i, v in enumerate(kats.QUAT_VECS):
            want = np.array(kats.QUAT_ANSWERS[j][i], np.float32)
            lik, ratio, _ = engine.measure_batch(pose, np.array([v], np.float32))
            assert ratio[0] == 1.0            # something within 0.05 of the rotated point ...
            assert lik[0] > 0.05 - 2e-6       # ... at distance < 2e-6: it can only be `want`
            assert np.any(np.all(axis_pts == want, 1))
    engine.set_likelihood_params()


def test_pf_entropy_kats(engine):
    """test/src/test_pf.cpp:330-391."""
    w0 = np.full(10, 0.1, np.float32)
    for case in kats.PF_ENTROPY_CASES:
        got = engine.pf_measure(w0, np.array(case["lik"], np.float32))
        assert not got["restored"]
        assert abs(got["entropy"] - case["entropy"]) <= max(case["tol"], 1e-7)
        assert abs(got["weights"].sum() - 1.0) < 1e-6
    e = [engine.pf_measure(w0, np.array(lk, np.float32))["entropy"] for lk in kats.PF_ENTROPY_ORDER]
    assert e[1] > e[0]
    dead = engine.pf_measure(w0, np.zeros(10, np.float32))
    assert dead["restored"] and np.array_equal(dead["weights"], w0) and np.isnan(dead["entropy"])


@pytest.mark.parametrize("name", golden_util.SCENE_FIXTURES)
def test_against_reference_goldens(engine, name):
    """tests/golden/<name>.npz: outputs of the real reference (oracle/_ref) on seeded scenes."""
    g, sc, dw, bkw = golden_util.load(name)
    engine.set_map(sc.map_xyz, sc.map_label, stamp=110 + golden_util.SCENE_FIXTURES.index(name), dist_weight=dw)
    engine.set_likelihood_params()
    engine.set_beam_params(**bkw)
    lik, ratio, beam = engine.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
    np.testing.assert_allclose(lik, g["lik"], rtol=1e-5)
    np.testing.assert_array_equal(ratio, g["quality"])
    np.testing.assert_array_equal(beam, g["beam"])
    begin, end = golden_util.rays_for(sc, 99)
    st, hit = engine.beam_status(begin, end)
    np.testing.assert_array_equal(st, g["status"])
    np.testing.assert_array_equal(hit, g["hit"])
    extra = golden_util.odom_factor(sc.odom_err, float(g["odom_sigma"]))
    upd = engine.measure_update(sc.poses, sc.weights, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins,
                                extra=extra)
    np.testing.assert_allclose(upd["weights"], g["upd_weights"], rtol=1e-5)
    np.testing.assert_allclose(upd["entropy"], g["upd_entropy"], rtol=1e-5)
    assert np.float32(upd["match_ratio_min"]) == g["upd_ratio"][0]
    assert np.float32(upd["match_ratio_max"]) == g["upd_ratio"][1]


def test_beam_wall_fixture_goldens(engine):
    """The upstream fixture of test/src/test_beam_likelihood.cpp:81-210: 2 modes x 6 hit ranges, per-ray status and
    measure() over the 100-pose sweep, against the reference's frozen output. (200 m x 200 m x 4 m map at 0.1 m voxels:
    a 164 M-voxel occupancy grid, the largest the reference's own tests build.)"""
    g = np.load(os.path.join(golden_util.HERE, "golden", "beam_wall_fixture.npz"))
    raw_pc, pc_map = kats.beam_wall_fixture()
    pc = raw_pc[(raw_pc[:, 2] > -0.3) & (raw_pc[:, 2] < 4.1)]
    xs = (0.1 * np.arange(-50, 50)).astype(np.float32)
    end = np.stack([xs, np.zeros_like(xs), np.zeros_like(xs)], 1)
    engine.set_map(pc_map, None, stamp=120, dist_weight=None)
    for mode in (0, 1):
        for k, hr in enumerate((0.0, 0.2, 0.4, 0.6, 0.8, 1.0)):
            engine.set_beam_params(num_points=len(raw_pc), hit_range=hr, dda_grid_size=0.1,
                                   add_penalty_short_only_mode=bool(mode))
            st, _ = engine.beam_status(np.zeros_like(end), end)
            np.testing.assert_array_equal(st, g["status_m%d_h%d" % (mode, k)])
            if k in (0, 2, 5):  # each pose has its own origin list {pos}: one small call per pose
                for i in range(0, 100, 7):
                    pose = np.array([[xs[i], 0, 0, 0, 0, 0, 1]], np.float32)
                    _, _, b = engine.measure_batch(pose, None, pc, np.zeros(len(pc), np.uint32), pose[:, :3])
                    assert b[0] == g["lik_m%d_h%d" % (mode, k)][i]
