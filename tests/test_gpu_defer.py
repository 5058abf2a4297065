"""Deferred overflow rounds of the tile-major likelihood kernel (likelihood_kernels.h: DeferQueue / defer_drain) and the packed
w words of the voxel records (map_compiler.h) against the plain forms: a minimum over the same candidates, the same term
arithmetic and the same fixed-order fp64 sums — likelihoods, match ratios and (strict_order) reference-order sums are
bit-identical, on maps where few, a quarter and most of the voxels hold more than four candidates."""
import numpy as np
import pytest

from mcl_3dl_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu


def crowded_scene(jitter, n_p, n_s, seed=5, extra_copies=0):
    sc = make_scene(n=61, n_p=n_p, n_s=n_s, n_b=0, seed=seed, map_jitter=jitter)
    if extra_copies:
        # several points per lattice site, a few centimetres apart: most voxels of the candidate index overflow, many hold
        # more than eight candidates (two and more overflow records per evaluation)
        rng = np.random.default_rng(seed + 1)
        parts = [sc.map_xyz] + [sc.map_xyz + rng.uniform(-0.03, 0.03, sc.map_xyz.shape).astype(np.float32)
                                for _ in range(extra_copies)]
        sc.map_xyz = np.ascontiguousarray(np.concatenate(parts, 0), np.float32)
        sc.map_label = np.zeros(len(sc.map_xyz), np.uint32)
    return sc


def run(engine, sc, n_p, n_s, defer, packed=1, parts=4, group=0, strict=0, stamp=9000, dist_weight=(1.0, 1.0, 5.0), ratio=0.0):
    try:
        engine.set_option("cand_voxel_ratio", ratio)
        engine.set_option("cand_record_parts", parts)
        engine.set_option("cand_packed", packed)
        engine.set_option("lik_defer", defer)
        engine.set_option("lik_group", group)
        engine.set_option("strict_order", strict)
        engine.set_option("lik_tiled_min", 256)
        engine.set_map(sc.map_xyz, sc.map_label, stamp=stamp, dist_weight=dist_weight)
        engine.set_likelihood_params()
        lik, ratio, _beam = engine.measure_batch(sc.poses[:n_p], sc.scan_lik[:n_s])
        st = engine.index_stats()
        active = int(engine.get_option("lik_defer_active")), int(engine.get_option("cand_packed_active"))
        return lik.copy(), ratio.copy(), st, active
    finally:
        engine.set_option("cand_voxel_ratio", 0.0)
        engine.set_option("cand_record_parts", 0)
        engine.set_option("cand_packed", 1)
        engine.set_option("lik_defer", 1)
        engine.set_option("lik_group", 0)
        engine.set_option("strict_order", 2)
        engine.set_option("lik_tiled_min", 1024)


@pytest.mark.parametrize("jitter,copies", [(0.0, 0), (0.045, 0), (0.02, 10)])
@pytest.mark.parametrize("n_p,n_s", [(64, 1500), (300, 4096), (37, 2049)])
def test_deferred_equals_immediate(engine, jitter, copies, n_p, n_s):
    sc = crowded_scene(jitter, n_p, n_s, extra_copies=copies)
    ratio = 0.5 if copies else 0.0   # eleven points per site in voxels of r / 2: most of them overflow, many twice and more
    a = run(engine, sc, n_p, n_s, defer=0, stamp=9001, ratio=ratio)
    b = run(engine, sc, n_p, n_s, defer=1, stamp=9001, ratio=ratio)
    assert b[3] == (1, 1) and a[3][0] == 0
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    assert np.count_nonzero(b[0] != 1.0) > 0
    if copies:
        assert b[2]["voxels_with_overflow"] > 0.45 * b[2]["voxels_with_candidates"]
        assert b[2]["voxels_over8"] > 0.1 * b[2]["voxels_with_candidates"]


@pytest.mark.parametrize("group", [4, 8, 32])
def test_every_particle_group_size(engine, group):
    sc = crowded_scene(0.045, 200, 3000)
    a = run(engine, sc, 200, 3000, defer=0, group=group, stamp=9002)
    b = run(engine, sc, 200, 3000, defer=1, group=group, stamp=9002)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_packed_words_equal_plain_words_on_every_query_path(engine):
    """tiled (immediate), per-particle and wide-record kernels read the packed form; the plain form stays available."""
    sc = crowded_scene(0.045, 96, 2500)
    for n_s in (2500, 700, 100):   # tiled / likelihood_kernel<256> / likelihood_kernel<64>
        ref = run(engine, sc, 96, n_s, defer=0, packed=0, stamp=9003)
        assert ref[3] == (0, 0)
        for parts in (4, 8):
            got = run(engine, sc, 96, n_s, defer=0, packed=1, parts=parts, stamp=9003)
            assert got[3][1] == 1
            np.testing.assert_array_equal(ref[0], got[0])
            np.testing.assert_array_equal(ref[1], got[1])


def test_strict_order_terms_go_through_the_queue_unchanged(engine):
    sc = crowded_scene(0.045, 64, 3000)
    a = run(engine, sc, 64, 3000, defer=0, strict=1, stamp=9004)
    b = run(engine, sc, 64, 3000, defer=1, strict=1, stamp=9004)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_modes_follow_the_map(engine):
    """lik_defer = 1 (default): the queue wherever the records are packed 64-byte ones — a map of centroids is then built with
    those instead of 128-byte records; lik_defer = 2: only where more than lik_defer_min_frac of the voxels overflow."""
    lattice = crowded_scene(0.0, 64, 2048)
    crowded = crowded_scene(0.045, 64, 2048)
    try:
        engine.set_likelihood_params()
        for mode, want in ((1, (1, 1)), (2, (0, 1)), (0, (0, 0))):
            engine.set_option("lik_defer", mode)
            engine.set_map(lattice.map_xyz, lattice.map_label, stamp=9005 + 10 * mode, dist_weight=(1.0, 1.0, 5.0))
            engine.measure_batch(lattice.poses[:64], lattice.scan_lik[:2048])
            assert int(engine.get_option("lik_defer_active")) == want[0]
            engine.set_map(crowded.map_xyz, crowded.map_label, stamp=9006 + 10 * mode, dist_weight=(1.0, 1.0, 5.0))
            engine.measure_batch(crowded.poses[:64], crowded.scan_lik[:2048])
            assert int(engine.get_option("lik_defer_active")) == want[1]
            if mode:
                assert engine.index_stats()["record_parts"] == 4   # never the 128-byte records once the queue is available
    finally:
        engine.set_option("lik_defer", 1)
        engine.set_map(lattice.map_xyz, lattice.map_label, stamp=9007, dist_weight=(1.0, 1.0, 5.0))


def sphere_cluster(centre, n, radius, seed):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return (np.asarray(centre, np.float32) + radius * d).astype(np.float32)


def test_a_voxel_with_more_than_63_candidates_keeps_the_plain_words(engine):
    """150 points on a 2 cm sphere: each is the nearest neighbour of the queries in its direction, so the voxels around the
    cluster keep far more than 63 candidates — the count no longer fits the packed word, the compiler writes the plain
    form, the queue stays off, and the answers are those of the cell scan (lik_index = 0) bit for bit."""
    sc = crowded_scene(0.0, 64, 2048)
    near = sc.map_xyz[np.argmin(np.linalg.norm(sc.map_xyz - (sc.true_pose[:3] + np.array([2.0, 0.0, 0.0], np.float32)), axis=1))]
    sc.map_xyz = np.ascontiguousarray(np.concatenate([sc.map_xyz, sphere_cluster(near, 150, 0.02, 3)], 0), np.float32)
    sc.map_label = np.zeros(len(sc.map_xyz), np.uint32)
    got = run(engine, sc, 64, 2048, defer=1, stamp=9100)
    assert got[3] == (0, 0), got[3]
    try:
        engine.set_option("lik_index", 0)
        engine.set_option("lik_tiled_min", 256)
        engine.set_option("strict_order", 0)   # (like run(): the tiled kernel's fp64 sums on both sides)
        engine.set_map(sc.map_xyz, sc.map_label, stamp=9101, dist_weight=(1.0, 1.0, 5.0))
        want = engine.measure_batch(sc.poses[:64], sc.scan_lik[:2048])
    finally:
        engine.set_option("lik_index", 2)
        engine.set_option("lik_tiled_min", 1024)
        engine.set_option("strict_order", 2)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])


def test_a_map_update_that_does_not_fit_the_packed_words_rebuilds_the_index():
    from mcl_3dl_amd import capi
    sc = crowded_scene(0.0, 48, 1500)
    near = sc.map_xyz[np.argmin(np.linalg.norm(sc.map_xyz - (sc.true_pose[:3] + np.array([2.0, 0.0, 0.0], np.float32)), axis=1))]
    cluster = sphere_cluster(near, 150, 0.02, 4)
    a, b = capi.Engine(0), capi.Engine(0)
    try:
        for e in (a, b):
            e.set_likelihood_params()
            e.set_option("lik_tiled_min", 256)
        a.set_map(sc.map_xyz, None, stamp=1, dist_weight=(1.0, 1.0, 5.0))
        a.measure_batch(sc.poses[:48], sc.scan_lik[:1500])
        assert int(a.get_option("cand_packed_active")) == 1
        n_map, stats = a.map_update(cluster, None, leaf=(0.0005, 0.0005, 0.0005), stamp=2)
        assert stats["outcome"] == 7, stats          # "does not fit the packed words": the next query rebuilds
        merged = a.map_download()[0]
        assert n_map == len(merged) == len(sc.map_xyz) + len(cluster)
        got = a.measure_batch(sc.poses[:48], sc.scan_lik[:1500])
        assert int(a.get_option("cand_packed_active")) == 0 and int(a.get_option("lik_defer_active")) == 0
        b.set_map(merged, None, stamp=3, dist_weight=(1.0, 1.0, 5.0))
        want = b.measure_batch(sc.poses[:48], sc.scan_lik[:1500])
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w)
    finally:
        a.close()
        b.close()


def test_options_from_the_environment(monkeypatch):
    from mcl_3dl_amd import capi
    monkeypatch.setenv("MCL3DL_HIP_OPTIONS", "lik_defer=0, cand_packed=0;strict_order=1")
    e = capi.Engine(0)
    try:
        assert (int(e.get_option("lik_defer")), int(e.get_option("cand_packed")), int(e.get_option("strict_order"))) == (0, 0, 1)
    finally:
        e.close()
    for bad in ("lik_defer=7", "no_such_option=1", "lik_defer", "lik_defer=abc"):
        monkeypatch.setenv("MCL3DL_HIP_OPTIONS", bad)
        with pytest.raises(Exception):
            capi.Engine(0)
    monkeypatch.setenv("MCL3DL_HIP_OPTIONS", "")
    e = capi.Engine(0)
    try:
        assert int(e.get_option("lik_defer")) == 1 and int(e.get_option("cand_packed")) == 1
    finally:
        e.close()


def irregular_map(name, rng):
    if name == "clutter":       # hundreds of points per cubic metre: dozens of candidates per voxel (may exceed 63: plain words)
        return rng.uniform(-2.0, 2.0, (60000, 3))
    if name == "plane":         # a noisy plane: crowded voxels in a sheet
        return np.stack([rng.uniform(-3, 3, 40000), rng.uniform(-3, 3, 40000), rng.normal(0, 0.01, 40000)], 1)
    if name == "duplicates":    # eight exactly coincident copies of each point
        return np.repeat(rng.uniform(-1, 1, (500, 3)), 8, 0)
    return np.concatenate([rng.uniform(-2.0, 2.0, (8000, 3)), np.repeat(rng.uniform(-1, 1, (300, 3)), 8, 0),
                           rng.uniform(-6, 6, (300, 3))], 0)


@pytest.mark.parametrize("name", ["clutter", "plane", "duplicates", "mix"])
@pytest.mark.parametrize("ratio", [0.0, 0.5, 1.0])
def test_irregular_maps_through_the_tiled_kernel(engine, name, ratio):
    """Random clutter, a noisy sheet, exactly coincident points: the tiled kernel with the queue (or, where a voxel exceeds 63
    candidates, with the plain words and immediate rounds) against the cell scan (lik_index = 0), bit for bit."""
    rng = np.random.default_rng(abs(hash(name)) % 1000 + int(ratio * 10))
    m = irregular_map(name, rng).astype(np.float32)
    scan = (m[rng.integers(0, len(m), 4096)] + rng.normal(0, 0.08, (4096, 3))).astype(np.float32)
    poses = np.zeros((24, 7), np.float32)
    poses[:, :3] = rng.normal(0, 0.05, (24, 3))
    poses[:, 3:6] = rng.normal(0, 0.01, (24, 3))
    poses[:, 6] = 1.0
    out = {}
    try:
        engine.set_likelihood_params(match_dist_min=0.2, match_dist_flat=0.0, match_weight=1.0)
        engine.set_option("cand_voxel_ratio", ratio)
        for mode in (0, 2):
            engine.set_option("lik_index", mode)
            engine.set_map(m, None, stamp=9300 + mode, dist_weight=(1.0, 2.0, 5.0) if name == "mix" else None)
            out[mode] = engine.measure_batch(poses, scan)
            if mode == 2:
                st = engine.index_stats()
                assert st["deferred_overflow"] == st["packed_words"]   # the queue whenever the words are packed
    finally:
        engine.set_option("lik_index", 2)
        engine.set_option("cand_voxel_ratio", 0.0)
        engine.set_likelihood_params()
    np.testing.assert_array_equal(out[0][0], out[2][0])
    np.testing.assert_array_equal(out[0][1], out[2][1])
    assert out[2][1].max() > 0.2
