"""SURVEY.md §8f-4 on the GPU: the map from the wire format + map VoxelGrid (src/mcl_3dl.cpp:128-139, 1140-1158), map
updates (pc_map2 = pc_map + VoxelGrid(update), :141-153, 1355-1369) with an INCREMENTAL update of the candidate-voxel index,
and the matched / unmatched output (:761-805) as one device pass."""
import struct

import numpy as np
import pytest

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import cube_map, make_scene
from oracle import pyoracle

pytestmark = pytest.mark.gpu
DW = (1.0, 1.0, 2.0)


@pytest.fixture(scope="module")
def ref():
    if not pyoracle.available("ref"):
        pytest.fail("oracle/_ref is not built")
    return pyoracle.Oracle("ref")


def pc2_bytes(xyz, label, step=20, offs=(0, 4, 8, 16)):
    buf = bytearray(step * len(xyz))
    for i in range(len(xyz)):
        struct.pack_into("<f", buf, i * step + offs[0], xyz[i, 0])
        struct.pack_into("<f", buf, i * step + offs[1], xyz[i, 1])
        struct.pack_into("<f", buf, i * step + offs[2], xyz[i, 2])
        if offs[3] >= 0:
            struct.pack_into("<I", buf, i * step + offs[3], int(label[i]))
    return bytes(buf)


def dense_raw_map(seed=5):
    """A raw (not yet down-sampled) map: the 9.1 m cube sampled densely with noise, labels on one wall."""
    rng = np.random.default_rng(seed)
    base = cube_map(91)
    raw = np.concatenate([base + rng.normal(0, 0.01, base.shape).astype(np.float32) for _ in range(3)], 0).astype(np.float32)
    label = (raw[:, 0] < -4.4).astype(np.uint32) * 2
    return raw, label


def test_map_from_pointcloud2_and_voxel_grid(ref):
    raw, label = dense_raw_map()
    eng = capi.Engine(0)
    try:
        n_map = eng.set_map_pointcloud2(pc2_bytes(raw, label), len(raw), 20, 0, 4, 8, 16, leaf=(0.1, 0.1, 0.1), stamp=5,
                                        dist_weight=DW)
        got_xyz, got_label = eng.map_download()
        want_xyz, want_label = ref.voxel_grid(raw, label, (0.1, 0.1, 0.1))
        assert n_map == len(want_xyz) < len(raw)
        np.testing.assert_array_equal(got_xyz, want_xyz)
        np.testing.assert_array_equal(got_label, want_label)
        assert set(np.unique(want_label)) == {0, 2}
        # the same through the array form, and the engine is a working engine on that map
        assert eng.set_map_downsampled(raw, label, leaf=(0.1, 0.1, 0.1), stamp=6, dist_weight=DW) == n_map
        np.testing.assert_array_equal(eng.map_download()[0], want_xyz)
        sc = make_scene(n=91, n_p=32, n_s=500, seed=2)
        eng.set_likelihood_params()
        lik, ratio, _ = eng.measure_batch(sc.poses, sc.scan_lik)
        o = pyoracle.Oracle("ref")
        o.set_map(want_xyz, want_label, dist_weight=DW)
        o.set_likelihood_params(pyoracle.LikelihoodParams())
        wl, wq = o.likelihood_measure(sc.poses, sc.scan_lik)
        np.testing.assert_allclose(lik, wl, rtol=1e-5)
        np.testing.assert_array_equal(ratio, wq)
    finally:
        eng.close()


def furniture(rng, n, centre):
    """A box-shaped obstacle (points on its faces) near `centre`: what a map update adds."""
    p = rng.uniform(-0.6, 0.6, (n, 3))
    ax = rng.integers(0, 3, n)
    p[np.arange(n), ax] = np.sign(p[np.arange(n), ax]) * 0.6
    return (p + np.asarray(centre)).astype(np.float32)


@pytest.mark.parametrize("n_s,parts", [(300, 0), (2000, 0), (2000, 8)])
def test_incremental_map_update_equals_a_fresh_index(ref, n_s, parts):
    """mapcloud_update: the updated engine (only the touched bricks re-compiled) answers exactly like a fresh engine that
    was given the merged map — likelihoods bit for bit; then a second update REPLACES the first; then the update is removed."""
    sc = make_scene(n=91, n_p=96, n_s=n_s, n_b=32, seed=4, sigma_xyz=(0.4, 0.4, 0.1))
    rng = np.random.default_rng(8)
    inc, fresh = capi.Engine(0), capi.Engine(0)
    try:
        for e in (inc, fresh):
            e.set_likelihood_params()
            e.set_beam_params(num_points=32)
        inc.set_option("cand_record_parts", parts)   # 8: the incremental path on 128-byte records; `fresh` keeps its own choice
        inc.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=DW)
        base_lik, _, _ = inc.measure_batch(sc.poses, sc.scan_lik)   # builds the index
        true_pos = sc.true_pose[:3]
        # update 0: a new surface 0.12 m in front of the walls the scan sees (so the scores must move); update 1: a box in
        # the room; update 2: nothing (the update is withdrawn)
        near = sc.map_xyz[np.linalg.norm(sc.map_xyz - true_pos, axis=1) < 4.0]
        inward = (true_pos - near) / np.linalg.norm(true_pos - near, axis=1, keepdims=True)
        updates = [(near + 0.12 * inward + rng.normal(0, 0.004, near.shape)).astype(np.float32),
                   furniture(rng, 2500, true_pos + np.array([-0.5, 1.8, 0.2])),
                   np.zeros((0, 3), np.float32)]
        leaf = (0.2, 0.2, 0.2)
        for step, upd in enumerate(updates):
            n_map, stats = inc.map_update(upd, None, leaf=leaf, stamp=10 + step)
            upd_ds = ref.voxel_grid(upd, None, leaf)[0] if len(upd) else np.zeros((0, 3), np.float32)
            merged = np.concatenate([sc.map_xyz, upd_ds], 0)
            assert n_map == len(merged)
            np.testing.assert_array_equal(inc.map_download()[0], merged)
            # incremental, not a rebuild: a few dozen bricks of the ~thousands, well under the full build time
            full_bricks = inc.index_stats()["bricks"]
            # incremental for the first two updates; the third may rebuild (outcome 5) once the withdrawn surface's
            # overflow records have piled up as orphans — results must be right either way
            if step < 2:
                assert stats["outcome"] == 0 and 0 < stats["bricks_recompiled"] < full_bricks // 2, (stats, full_bricks)
            else:
                assert stats["outcome"] in (0, 5), stats
            fresh.set_map(merged, None, stamp=100 + step, dist_weight=DW)
            got = inc.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
            want = fresh.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
            for g, w in zip(got, want):
                np.testing.assert_array_equal(g, w)
            if step == 0:
                assert not np.array_equal(got[0], base_lik)   # the update really changes the scores
            elif len(upd) == 0:
                np.testing.assert_array_equal(got[0], base_lik)  # update removed: back to the base map's answers
            print("update %d: %s; full build %.2f ms" % (step, stats, fresh.index_stats()["build_ms"]))
        # radius search (cell grid) and beam status (DDA grid) see the merged map too
        q = (true_pos + rng.normal(0, 1.0, (500, 3))).astype(np.float32)
        np.testing.assert_array_equal(inc.radius_search(q, 0.3)[0], fresh.radius_search(q, 0.3)[0])
    finally:
        inc.close()
        fresh.close()


def test_a_stream_of_replacing_updates_reclaims_its_orphaned_overflow_records():
    """mapcloud_update REPLACES the previous update every time (src/mcl_3dl.cpp:141-153): each update orphans the overflow records
    the previous one appended. They are reclaimed in place (compact_overflow) — every update stays incremental (outcome 0, no
    whole-map rebuild in between) and every answer equals a fresh engine's on the merged map, bit for bit. The map is a cloud of
    voxel-filter centroids (displaced lattice points): a quarter of its voxels keep overflow records."""
    sc = make_scene(n=91, n_p=64, n_s=1500, n_b=0, seed=14, sigma_xyz=(0.3, 0.3, 0.1), map_jitter=0.045)
    rng = np.random.default_rng(18)
    inc, fresh = capi.Engine(0), capi.Engine(0)
    try:
        for e in (inc, fresh):
            e.set_likelihood_params()
        inc.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
        inc.measure_batch(sc.poses, sc.scan_lik)
        true_pos = sc.true_pose[:3]
        near = sc.map_xyz[np.linalg.norm(sc.map_xyz - true_pos, axis=1) < 3.5]
        inward = (true_pos - near) / np.linalg.norm(true_pos - near, axis=1, keepdims=True)
        leaf = (0.1, 0.1, 0.1)
        rebuilds = 0
        for step in range(9):
            upd = (near + (0.08 + 0.015 * step) * inward + rng.normal(0, 0.02, near.shape)).astype(np.float32)
            n_map, stats = inc.map_update(upd, None, leaf=leaf, stamp=20 + step)
            rebuilds += stats["outcome"] != 0
            merged = inc.map_download()[0]
            assert n_map == len(merged)
            fresh.set_map(merged, None, stamp=200 + step, dist_weight=(1.0, 1.0, 1.0))
            got = inc.measure_batch(sc.poses, sc.scan_lik)
            want = fresh.measure_batch(sc.poses, sc.scan_lik)
            np.testing.assert_array_equal(got[0], want[0], err_msg="update %d" % step)
            np.testing.assert_array_equal(got[1], want[1], err_msg="update %d" % step)
        assert rebuilds == 0
        assert inc.get_option("cand_ovf_compactions") >= 1
        # the orphans never outgrow the array they live in (they are reclaimed at half of it)
        assert inc.get_option("cand_ovf_leaked") < inc.index_stats()["overflow_records"]
    finally:
        inc.close()
        fresh.close()


def test_map_updates_ride_on_the_dda_grid_as_an_overlay():
    """The DDA grid keeps its arrays over a stream of map updates (DdaGrid::ov_*: the update's points sorted by voxel, looked up
    behind a voxel's base points; the occupancy bits of the previous update withdrawn, the new ones set): beam scores, per-ray
    status and the collided point's map index equal a fresh engine's on the merged map — for updates that replace each other,
    one that leaves the base map's bounds (rebuild), the one after it (overlay again), an empty one, and with the overlay
    switched off. Labels above filter_label_max make collided update points transparent like base points."""
    sc = make_scene(n=61, n_p=128, n_s=200, n_b=256, seed=24, sigma_xyz=(0.4, 0.4, 0.1))
    rng = np.random.default_rng(28)
    inc, fresh, plain = capi.Engine(0), capi.Engine(0), capi.Engine(0)
    try:
        plain.set_option("dda_overlay", 0)
        for e in (inc, fresh, plain):
            e.set_likelihood_params()
            e.set_beam_params(num_points=256, filter_label_max=1)
        for e in (inc, plain):
            e.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=DW)
            e.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)   # builds the DDA grid on the base map
        true_pos = sc.true_pose[:3]
        lo, hi = sc.map_xyz.min(0), sc.map_xyz.max(0)
        begin = (true_pos + rng.normal(0, 0.3, (3000, 3))).astype(np.float32)
        end = (true_pos + rng.normal(0, 4.0, (3000, 3))).astype(np.float32)
        updates = [furniture(rng, 1500, true_pos + np.array([1.2 * np.cos(k), 1.2 * np.sin(k), 0.1 * k])) for k in range(5)]
        updates.append((rng.uniform(0, 1, (400, 3)) + hi + 2.0).astype(np.float32))      # outside the base bounds: a rebuild
        updates.append(furniture(rng, 1500, true_pos + np.array([-1.0, 0.5, 0.0])))      # inside again
        updates.append(np.zeros((0, 3), np.float32))                                     # withdrawn
        updates.append(furniture(rng, 800, true_pos + np.array([0.3, -1.1, 0.2])))
        applied = 0
        for step, upd in enumerate(updates):
            lab = rng.integers(0, 3, len(upd)).astype(np.uint32)
            before = inc.get_option("dda_overlay_updates")
            inc.map_update(upd, lab, leaf=(0.1, 0.1, 0.1), stamp=10 + step)
            plain.map_update(upd, lab, leaf=(0.1, 0.1, 0.1), stamp=10 + step)
            applied += inc.get_option("dda_overlay_updates") - before
            m_xyz, m_lab = inc.map_download()
            inside = len(upd) == 0 or bool(np.all((m_xyz[len(sc.map_xyz):] >= lo) & (m_xyz[len(sc.map_xyz):] <= hi)))
            fresh.set_map(m_xyz, m_lab, stamp=100 + step, dist_weight=DW)
            want = fresh.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
            w_st, w_hit = fresh.beam_status(begin, end)
            for e in (inc, plain):
                got = e.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
                np.testing.assert_array_equal(got[2], want[2], err_msg="update %d" % step)
                np.testing.assert_array_equal(got[0], want[0], err_msg="update %d" % step)
                st, hit = e.beam_status(begin, end)
                np.testing.assert_array_equal(st, w_st, err_msg="update %d" % step)
                np.testing.assert_array_equal(hit, w_hit, err_msg="update %d" % step)
            if step == 0:
                # the reference's own sizes (one-launch update: compiled without the overlay lookup, so it must step aside)
                w0 = np.full(64, 1.0 / 64, np.float32)
                a = inc.measure_update(sc.poses[:64], w0, sc.scan_lik[:96], sc.scan_beam[:3], sc.scan_beam_label[:3], sc.origins)
                b = fresh.measure_update(sc.poses[:64], w0, sc.scan_lik[:96], sc.scan_beam[:3], sc.scan_beam_label[:3], sc.origins)
                for k in ("weights", "beam", "lik"):
                    np.testing.assert_array_equal(a[k], b[k], err_msg=k)
            # the engine with the overlay holds the update's points there whenever they are inside the base map's bounds
            assert inc.get_option("dda_overlay_points") == (len(m_xyz) - len(sc.map_xyz) if inside else 0), step
            if len(upd) and inside:
                assert np.any(w_hit >= len(sc.map_xyz))   # rays do collide with the update's points
        # every update but the one outside the bounds and the one right after it (its grid had been laid out for the far
        # points) was applied without a rebuild
        assert applied == len(updates) - 2, applied
        assert plain.get_option("dda_overlay_updates") == 0
    finally:
        inc.close()
        fresh.close()
        plain.close()


def test_update_outside_the_grid_falls_back_to_a_rebuild():
    sc = make_scene(n=61, n_p=16, n_s=300, seed=1)
    a, b = capi.Engine(0), capi.Engine(0)
    try:
        for e in (a, b):
            e.set_likelihood_params()
        a.set_map(sc.map_xyz, None, stamp=1)
        a.measure_batch(sc.poses, sc.scan_lik)
        far = (np.random.default_rng(0).uniform(0, 1, (500, 3)) + np.array([30.0, 0, 0])).astype(np.float32)
        n_map, stats = a.map_update(far, None, leaf=(0.2, 0.2, 0.2), stamp=2)
        assert stats["bricks_recompiled"] == 0           # no incremental path: the grid has to be laid out again
        merged = a.map_download()[0]
        assert n_map == len(merged) > len(sc.map_xyz)
        b.set_map(merged, None, stamp=3)
        poses = sc.poses.copy()
        poses[:4, :3] = [30.5, 0.5, 0.5]
        for g, w in zip(a.measure_batch(poses, sc.scan_lik), b.measure_batch(poses, sc.scan_lik)):
            np.testing.assert_array_equal(g, w)
    finally:
        a.close()
        b.close()


def test_matched_unmatched_split(engine):
    # The node builds its kd-tree with max_search_radius = 0.4 but searches with unmatch_output_dist = 0.5: across a 20 m
    # chunk border the reference itself can miss a neighbour between 0.4 and 0.5 m away (chunked_kdtree.h:139-201). The
    # engine returns the global nearest neighbour; the oracle here gets a border margin that covers the radius.
    sc = make_scene(n=91, n_p=4, n_s=3000, seed=6)
    rng = np.random.default_rng(1)
    cloud = np.concatenate([sc.scan_lik, sc.scan_lik[:800] + rng.normal(0, 0.08, (800, 3)).astype(np.float32),
                            rng.uniform(-3, 3, (400, 3)).astype(np.float32)], 0)
    for dw in (DW, None):
        engine.set_map(sc.map_xyz, sc.map_label, stamp=8201, dist_weight=dw)
        engine.set_likelihood_params()
        ref = pyoracle.Oracle("ref", max_search_radius=0.6)  # a fresh kd-tree: its point representation is set once
        ref.set_map(sc.map_xyz, sc.map_label, dist_weight=dw)
        pose = sc.true_pose.copy()
        pose[3:7] *= np.float32(1.7)  # transform() normalises the quaternion
        m, u = engine.match_split(pose, cloud, unmatch_dist=0.5, match_dist=0.1)
        cls, moved = ref.match_split(pose, cloud, 0.5, 0.1)
        np.testing.assert_array_equal(m, moved[cls == 1])
        np.testing.assert_array_equal(u, moved[cls == 2])
        assert len(m) > 1000 and len(u) > 50 and (cls == 0).sum() > 50
    # on the cloud scan_begin left on the device (pc_local_full)
    n_full, _, _ = engine.scan_begin(cloud, None, leaf=(0.1, 0.1, 0.1))
    full, _ = engine.scan_download(0)
    m2, u2 = engine.match_split(pose, None, unmatch_dist=0.5, match_dist=0.1)
    cls2, moved2 = ref.match_split(pose, full, 0.5, 0.1)
    np.testing.assert_array_equal(m2, moved2[cls2 == 1])
    np.testing.assert_array_equal(u2, moved2[cls2 == 2])


def test_matched_unmatched_split_variants(engine):
    """The one-kernel form (both compactions written into page-locked memory, polled completion) against the general form
    (update_zero_copy = 0: scans, count read-backs, compactions, D2H copies); the caller's own page-locked arrays written in
    place; counts without clouds; an output array that is too small."""
    sc = make_scene(n=61, n_p=8, n_s=6000, n_b=0, seed=31, sigma_xyz=(0.3, 0.3, 0.1))
    engine.set_map(sc.map_xyz, sc.map_label, stamp=7301, dist_weight=DW)
    engine.set_likelihood_params()
    rng = np.random.default_rng(3)
    cloud = np.concatenate([sc.scan_lik, rng.uniform(-3, 3, (2500, 3)).astype(np.float32)], 0)   # some far from every map point
    pose = np.asarray(sc.true_pose, np.float32)
    m, u = engine.match_split(pose, cloud, unmatch_dist=0.5, match_dist=0.1)
    assert len(m) > 100 and len(u) > 100
    try:
        engine.set_option("update_zero_copy", 0)
        m0, u0 = engine.match_split(pose, cloud, unmatch_dist=0.5, match_dist=0.1)
    finally:
        engine.set_option("update_zero_copy", 1)
    np.testing.assert_array_equal(m, m0)
    np.testing.assert_array_equal(u, u0)
    hm, hu = engine.host_array((len(cloud), 3)), engine.host_array((len(cloud), 3))
    try:
        hm[:] = -7.0
        n_m, n_u = engine.match_split_into(pose, hm, hu, xyz=cloud)
        assert (n_m, n_u) == (len(m), len(u))
        np.testing.assert_array_equal(hm[:n_m], m)
        np.testing.assert_array_equal(hu[:n_u], u)
        assert np.all(hm[n_m:] == -7.0)
    finally:
        engine.host_free(hm)
        engine.host_free(hu)
    assert engine.match_split_into(pose, None, None, xyz=cloud) == (len(m), len(u))
    small = np.zeros((len(m) - 1, 3), np.float32)
    with pytest.raises(capi.EngineError, match="output capacity"):
        engine.match_split_into(pose, small, np.zeros((len(cloud), 3), np.float32), xyz=cloud)
    # the context is fine afterwards
    m1, u1 = engine.match_split(pose, cloud, unmatch_dist=0.5, match_dist=0.1)
    np.testing.assert_array_equal(m1, m)


def test_device_built_grids_equal_the_host_built_ones():
    """The cell-sorted exact-NN grid and the DDA occupancy / voxel index are counting sorts of the map: built on the device
    (default) and by the sequential host form (option grid_build_host) they must answer every query identically — radius
    search (index AND squared distance), the 27-cell likelihood (lik_index 0), beam scores, per-ray status and the collided
    map point — on a map with labels, a dist_weight and several points per cell."""
    sc = make_scene(n=91, n_p=96, n_s=700, n_b=128, seed=21)
    rng = np.random.default_rng(8)
    # a second, shifted copy of the walls: up to several points per DDA voxel / grid cell, in a known insertion order
    map_xyz = np.concatenate([sc.map_xyz, sc.map_xyz + rng.normal(0, 0.03, sc.map_xyz.shape).astype(np.float32)], 0)
    map_label = np.concatenate([sc.map_label, (np.arange(len(sc.map_xyz)) % 3).astype(np.uint32)])
    queries = (map_xyz[rng.integers(0, len(map_xyz), 5000)] + rng.normal(0, 0.15, (5000, 3))).astype(np.float32)
    begin = np.tile(sc.true_pose[:3] + np.array([0, 0, 0.5], np.float32), (3000, 1)).astype(np.float32)
    end = (begin + rng.normal(0, 4.0, (3000, 3))).astype(np.float32)
    out = {}
    for mode in (0, 1):
        eng = capi.Engine(0)
        try:
            eng.set_option("grid_build_host", mode)
            eng.set_option("lik_index", 0)
            eng.set_map(map_xyz, map_label, stamp=31 + mode, dist_weight=DW)
            eng.set_likelihood_params()
            eng.set_beam_params(num_points=128, filter_label_max=1)
            lik, ratio, beam = eng.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
            idx, sq = eng.radius_search(queries, 0.35)
            st, hit = eng.beam_status(begin, end)
            m, u = eng.match_split(sc.true_pose, sc.scan_lik, unmatch_dist=0.5, match_dist=0.1)
            out[mode] = (lik, ratio, beam, idx, sq, st, hit, m, u)
            assert eng.get_option("grid_build_host") == mode
            if mode == 0:
                assert eng.get_option("lik_grid_build_ms") > 0 and eng.get_option("dda_grid_build_ms") > 0
        finally:
            eng.close()
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(a, b)
    assert (out[0][3] >= 0).sum() > 1000 and len(np.unique(out[0][5])) >= 2 and out[0][6].max() >= len(sc.map_xyz)


def test_the_cell_grid_takes_a_map_update_by_merging(engine):
    """Round 5: the cell-sorted grid behind matched / unmatched, radius_search and lik_index = 0 used to
    be rebuilt whole (keys, histogram, sort of the map, gather, scan) on the first use after every map update. Now the BASE map's
    grid is kept and the update's points are merged into a copy of it — pc_map2 = pc_map + pc_update puts them behind the base
    points of their cell (src/mcl_3dl.cpp:150) — unless the update leaves the bounds the grid was laid out for. Equal, query by
    query, to a fresh engine on the merged map."""
    from mcl_3dl_amd import capi
    sc = make_scene(n=91, n_p=64, n_s=1500, seed=8)
    rng = np.random.default_rng(12)
    half = 91 * 0.1 / 2
    fresh = capi.Engine(0)
    dw = (1.0, 1.0, 5.0)
    try:
        engine.set_map(sc.map_xyz, sc.map_label, stamp=4400, dist_weight=dw)
        engine.set_likelihood_params()
        queries = (sc.scan_lik[:1200] @ np.eye(3, dtype=np.float32) + sc.true_pose[:3]).astype(np.float32)
        engine.radius_search(queries, 0.3)                           # builds the base grid
        r0, m0 = engine.get_option("lik_grid_rebuilds"), engine.get_option("lik_grid_merges")
        for k, inside in enumerate([True, True, False, True, True]):
            span = half - 0.5 if inside else half + 2.0
            pts = rng.uniform(-span, span, (int(rng.integers(1, 400)), 3)).astype(np.float32)
            pts[:, 2] = rng.uniform(-half + 0.1, -half + 2.0, len(pts))
            engine.map_update(pts, rng.integers(0, 3, len(pts)).astype(np.uint32), leaf=(0.2, 0.2, 0.2), stamp=4401 + k)
            mx, ml = engine.map_download()
            fresh.set_map(mx, ml, stamp=4500 + k, dist_weight=dw)
            fresh.set_likelihood_params()
            q = np.concatenate([queries, (mx[-len(pts):] + rng.normal(0, 0.05, (len(pts), 3))).astype(np.float32)])
            for radius in (0.2, 0.5):
                a, b = engine.radius_search(q, radius), fresh.radius_search(q, radius)
                np.testing.assert_array_equal(a[0], b[0])
                np.testing.assert_array_equal(a[1], b[1])
            ms_a = engine.match_split(sc.true_pose, sc.scan_lik)
            ms_b = fresh.match_split(sc.true_pose, sc.scan_lik)
            for x, y in zip(ms_a, ms_b):
                np.testing.assert_array_equal(x, y)
            try:
                engine.set_option("lik_index", 0)
                fresh.set_option("lik_index", 0)
                np.testing.assert_array_equal(engine.measure_batch(sc.poses, sc.scan_lik)[1],
                                              fresh.measure_batch(sc.poses, sc.scan_lik)[1])
            finally:
                engine.set_option("lik_index", 2)
                fresh.set_option("lik_index", 2)
        r1, m1 = engine.get_option("lik_grid_rebuilds"), engine.get_option("lik_grid_merges")
        # five updates: the one that left the bounds rebuilt the base (for the wider bounds), the others merged
        assert r1 - r0 == 1 and m1 - m0 == 4, (r0, r1, m0, m1)
    finally:
        fresh.close()
