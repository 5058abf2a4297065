"""SURVEY.md §8f-2 on the GPU: VoxelGrid down-sampling of the accumulated scan (src/mcl_3dl.cpp:363-367), both models' clip
filter (likelihood.cpp:79-103, beam.cpp:98-122), the uniform sampler's gather (point_cloud_uniform_sampler.h:56-74) and the
scan ordering — `mcl3dl_hip_scan_begin` / `_finish` — against
  * the reference's own filter() + PointCloudUniformSampler compiled into oracle/_ref (engine seeded by the test), and
  * the restated pcl::VoxelGrid of oracle/shims (PCL is not vendored by the reference: parity unpinned at that boundary,
    bit-exact against the restatement).
The prepared scan then feeds the measurement update without a host round trip: results must equal uploading the same
sampled clouds through mcl3dl_hip_measure_batch, bit for bit."""
import struct

import numpy as np
import pytest

from mcl_3dl_amd.synthetic import make_scene
from oracle import pyoracle

pytestmark = pytest.mark.gpu

LEAF = (0.1, 0.1, 0.1)
CLIP_LIK = (0.5, 10.0, -2.0, 2.0)   # clip_near, clip_far, clip_z_min, clip_z_max (parameters.h defaults)
CLIP_BEAM = (0.5, 4.0, -2.0, 2.0)


@pytest.fixture(scope="module")
def ref():
    if not pyoracle.available("ref"):
        pytest.fail("oracle/_ref is not built: run `python -c 'import __graft_entry__ as g; g.build()'` where "
                    "/root/reference exists (the built library travels to the GPU box)")
    o = pyoracle.Oracle("ref")
    o.set_likelihood_params(pyoracle.LikelihoodParams(num_points=500))
    o.set_beam_params(pyoracle.BeamParams(num_points=40))
    return o


def raw_cloud(n=60000, seed=3, n_clouds=3, with_nan=True):
    """An accumulated scan in the robot frame: a dense noisy shell of a room + clutter; label = accumulated-cloud index."""
    rng = np.random.default_rng(seed)
    d = rng.normal(0, 1, (n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.minimum.reduce([np.abs(6.0 / np.maximum(np.abs(d[:, 0]), 1e-3)), np.abs(4.5 / np.maximum(np.abs(d[:, 1]), 1e-3)),
                           np.abs(1.4 / np.maximum(np.abs(d[:, 2]), 1e-3))])
    xyz = (d * r[:, None] + rng.normal(0, 0.01, (n, 3))).astype(np.float32)
    xyz[: n // 10] = rng.uniform(-12, 12, (n // 10, 3)).astype(np.float32)   # clutter, also beyond clip_far
    label = rng.integers(0, n_clouds, n).astype(np.uint32)
    if with_nan:
        xyz[::997, 1] = np.nan
        xyz[5::1201, 0] = np.inf
    return xyz, label


def test_voxel_grid_equals_the_restated_pcl_filter(engine, ref):
    xyz, label = raw_cloud()
    for leaf in (LEAF, (0.25, 0.2, 0.5), (1.0, 1.0, 1.0)):
        n_full, _, _ = engine.scan_begin(xyz, label, leaf=leaf, clip_lik=CLIP_LIK, clip_beam=CLIP_BEAM)
        got_xyz, got_label = engine.scan_download(0)
        want_xyz, want_label = ref.voxel_grid(xyz, label, leaf)
        assert n_full == len(want_xyz) and n_full < len(xyz)
        np.testing.assert_array_equal(got_xyz, want_xyz)      # same leaves, same order, same float centroid
        np.testing.assert_array_equal(got_label, want_label)  # majority label, smallest on a tie
    assert len(np.unique(want_label)) == 3


def test_voxel_grid_edge_cases(engine, ref):
    # leaf so small that the index would overflow 32 bits: PCL hands the input back unchanged
    xyz, label = raw_cloud(n=5000, with_nan=False)
    n_full, _, _ = engine.scan_begin(xyz, label, leaf=(1e-4, 1e-4, 1e-4))
    got, _ = engine.scan_download(0)
    want, _ = ref.voxel_grid(xyz, label, (1e-4, 1e-4, 1e-4))
    assert n_full == len(xyz) == len(want)
    np.testing.assert_array_equal(got, xyz)
    # every point in one leaf; a single point; an empty cloud; no leaf at all (filter skipped)
    one = np.tile(np.array([[0.31, 0.32, 0.33]], np.float32), (1000, 1)) + np.linspace(0, 1e-3, 1000, dtype=np.float32)[:, None]
    n_full, _, _ = engine.scan_begin(one, None, leaf=(1.0, 1.0, 1.0))
    got, _ = engine.scan_download(0)
    want, _ = ref.voxel_grid(one, None, (1.0, 1.0, 1.0))
    assert n_full == 1
    np.testing.assert_array_equal(got, want)
    assert engine.scan_begin(one[:1], None, leaf=LEAF)[0] == 1
    assert engine.scan_begin(np.zeros((0, 3), np.float32), None, leaf=LEAF) == (0, 0, 0)
    assert engine.scan_begin(xyz, label, leaf=None)[0] == len(xyz)


def test_clip_and_uniform_sampling_equal_the_reference_filter(engine, ref):
    """filter() of both reference models with the reference's PointCloudUniformSampler, same seed."""
    xyz, label = raw_cloud()
    n_full, n_lik, n_beam = engine.scan_begin(xyz, label, leaf=LEAF, clip_lik=CLIP_LIK, clip_beam=CLIP_BEAM)
    full, full_label = engine.scan_download(0)
    # clip step alone
    for model, which, cnt in ((0, 1, n_lik), (1, 2, n_beam)):
        want_xyz, want_label, num = ref.clip(model, full, full_label)
        got_xyz, got_label = engine.scan_download(which)
        assert cnt == len(want_xyz) and 0 < cnt < n_full
        np.testing.assert_array_equal(got_xyz, want_xyz)
        np.testing.assert_array_equal(got_label, want_label)
        assert num == (500, 40)[model]
    # sampling: the caller draws the indices with the reference's distribution and engine
    seed_l, seed_b = 4242, 777
    idx_l = ref.uniform_indices(seed_l, n_lik, 500)
    idx_b = ref.uniform_indices(seed_b, n_beam, 40)
    engine.scan_finish(idx_l, idx_b, origins=np.array([[0, 0, 0.5], [0.1, 0, 0.5], [0.2, 0, 0.5]], np.float32))
    got_l, got_ll = engine.scan_download(3)
    got_b, got_bl = engine.scan_download(4)
    want_l, want_ll = ref.filter_uniform(0, full, full_label, seed_l, 4096)
    want_b, want_bl = ref.filter_uniform(1, full, full_label, seed_b, 4096)
    np.testing.assert_array_equal(got_l, want_l)
    np.testing.assert_array_equal(got_ll, want_ll)
    np.testing.assert_array_equal(got_b, want_b)
    np.testing.assert_array_equal(got_bl, want_bl)


def test_prepared_scan_feeds_the_update_without_a_host_round_trip(engine, ref, oracle_kind):
    """scan_begin + scan_finish install the scans on the device; the update that follows equals the update on the same
    sampled clouds uploaded from the host (same ordering keys, same stable order: bit-identical), and the oracle."""
    import torch
    sc = make_scene(n=91, n_p=128, n_s=4000, n_b=600, seed=9)
    # the "accumulated cloud" = the scene's scans with duplicates and clutter, labels = origin index 0
    raw = np.concatenate([sc.scan_lik, sc.scan_beam, sc.scan_lik[::3] + np.float32(0.004)], 0)
    engine.set_map(sc.map_xyz, sc.map_label, stamp=8101, dist_weight=(1.0, 1.0, 3.0))
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=64)
    n_full, n_lik, n_beam = engine.scan_begin(raw, None, leaf=(0.05, 0.05, 0.05), clip_lik=CLIP_LIK, clip_beam=CLIP_BEAM)
    assert n_lik > 1500 and n_beam > 100
    idx_l = ref.uniform_indices(11, n_lik, 1500)   # >= 1024 points: the tiled kernel
    idx_b = ref.uniform_indices(12, n_beam, 64)
    engine.scan_finish(idx_l, idx_b, origins=sc.origins)
    lik_cloud, _ = engine.scan_download(3)
    beam_cloud, beam_label = engine.scan_download(4)
    dev = torch.device("cuda", 0)
    d_pose = torch.from_numpy(sc.poses).to(dev)
    d_lik, d_ratio, d_beam = (torch.empty(len(sc.poses), dtype=torch.float32, device=dev) for _ in range(3))
    engine.measure_device(d_pose, len(sc.poses), d_lik, d_ratio, d_beam)
    engine.synchronize()
    torch.cuda.synchronize()
    got = d_lik.cpu().numpy(), d_ratio.cpu().numpy(), d_beam.cpu().numpy()
    want = engine.measure_batch(sc.poses, lik_cloud, beam_cloud, beam_label, sc.origins)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    o = pyoracle.Oracle(oracle_kind)
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=(1.0, 1.0, 3.0))
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    o.set_beam_params(pyoracle.BeamParams(num_points=64))
    wl, wq = o.likelihood_measure(sc.poses, lik_cloud)
    wb, _ = o.beam_measure(sc.poses, beam_cloud, beam_label, sc.origins)
    np.testing.assert_allclose(got[0], wl, rtol=1e-5)
    np.testing.assert_array_equal(got[1], wq)
    np.testing.assert_array_equal(got[2], wb)
    assert np.count_nonzero(wl) > 64


def test_scan_from_the_wire_format(engine, ref):
    """sensor_msgs/PointCloud2 bytes (x, y, z, intensity, label at odd offsets, point_step 32) decode to the same cloud."""
    xyz, label = raw_cloud(n=20000, with_nan=False)
    step = 32
    buf = bytearray(step * len(xyz))
    for i in range(len(xyz)):
        struct.pack_into("<fff", buf, i * step + 4, *xyz[i])
        struct.pack_into("<f", buf, i * step + 16, 0.5)
        struct.pack_into("<I", buf, i * step + 24, int(label[i]))
    a = engine.scan_begin_pointcloud2(bytes(buf), len(xyz), step, 4, 8, 12, off_label=24, label_override=0xFFFFFFFF, leaf=LEAF)
    got = engine.scan_download(0)
    b = engine.scan_begin(xyz, label, leaf=LEAF)
    want = engine.scan_download(0)
    assert a == b
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    # label_override = 0: every point belongs to accumulated cloud 0 (src/mcl_3dl.cpp:295-298)
    engine.scan_begin_pointcloud2(bytes(buf), len(xyz), step, 4, 8, 12, off_label=24, label_override=0, leaf=LEAF)
    assert not engine.scan_download(0)[1].any()
    from mcl_3dl_amd import capi
    with pytest.raises(capi.EngineError, match="x, y, z"):
        engine.scan_begin_pointcloud2(bytes(buf), len(xyz), step, 4, -1, 12)
    with pytest.raises(capi.EngineError, match="empty"):
        engine.scan_begin_pointcloud2(b"", 0, step, 4, 8, 12)


def test_scan_finish_rejects_bad_draws(engine):
    from mcl_3dl_amd import capi
    xyz, label = raw_cloud(n=5000, with_nan=False)
    _, n_lik, n_beam = engine.scan_begin(xyz, label, leaf=LEAF)
    with pytest.raises(capi.EngineError, match="outside the clipped cloud"):
        engine.scan_finish(np.array([0, n_lik], np.uint32), None)
    with pytest.raises(capi.EngineError, match="origin"):
        engine.scan_finish(np.array([0], np.uint32), np.array([0, 1], np.uint32), origins=np.zeros((1, 3), np.float32))
    far = np.full((100, 3), 50.0, np.float32)   # everything clipped away: the sampler would return an empty cloud
    assert engine.scan_begin(far, None, leaf=LEAF)[1:] == (0, 0)
    engine.scan_finish(np.zeros(0, np.uint32), None)   # empty scans are fine: (1, 0) results downstream
    with pytest.raises(capi.EngineError, match="empty clipped cloud"):
        engine.scan_finish(np.array([0], np.uint32), None)


def test_scans_ordered_on_the_device_equal_scans_ordered_on_the_host(engine):
    """upload_scan orders large scans on the device (option scan_order_device) and small ones on the host: same keys, same
    stable order — the permutation is identical, so every result is, bit for bit (also strict_order, which replays the
    terms in ORIGINAL scan order through that permutation). The scan holds exact duplicates and many points per Morton cell."""
    sc = make_scene(n=91, n_p=50, n_s=5000, n_b=300, seed=14)
    lik = np.concatenate([sc.scan_lik, sc.scan_lik[:700], sc.scan_lik[:50] + np.float32(1e-4)], 0)
    beam = np.concatenate([sc.scan_beam, sc.scan_beam[:40]], 0)
    blab = np.concatenate([sc.scan_beam_label, sc.scan_beam_label[:40]], 0)
    engine.set_map(sc.map_xyz, sc.map_label, stamp=8301, dist_weight=(1.0, 1.0, 2.0))
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=len(beam))
    d = engine.get_option("scan_order_device")
    out = {}
    try:
        for strict in (0, 1):
            engine.set_option("strict_order", strict)
            for where in (0, 1):
                engine.set_option("scan_order_device", where)   # 0 = host, 1 = device for every size
                out[(strict, where)] = engine.measure_batch(sc.poses, lik, beam, blab, sc.origins)
    finally:
        engine.set_option("scan_order_device", d)
        engine.set_option("strict_order", 2)
    for strict in (0, 1):
        for a, b in zip(out[(strict, 0)], out[(strict, 1)]):
            np.testing.assert_array_equal(a, b)
    from mcl_3dl_amd import capi
    engine.set_option("scan_order_device", 1)
    try:
        with pytest.raises(capi.EngineError, match="names origin"):
            engine.measure_batch(sc.poses, lik, beam, np.full(len(beam), 3, np.uint32), sc.origins)
    finally:
        engine.set_option("scan_order_device", d)


@pytest.mark.parametrize("reach", [40.0, 130.0, 400.0])
def test_device_and_host_ordering_agree_when_the_morton_key_drops_bits(engine, reach):
    """The scan ordering keeps the 22 most significant Morton bits the scan's extent can set (cloud_keys.h): a scan wider than
    32 m drops low bits, one wider than 256 m also clamps cells — host and device must still make the same keys (same
    permutation: bit-identical results, also through strict_order's replay in original order)."""
    sc = make_scene(n=91, n_p=40, n_s=5000, n_b=0, seed=15)
    rng = np.random.default_rng(3)
    far = (rng.uniform(-1.0, 1.0, (300, 3)) * np.array([reach, reach, 0.1 * reach])).astype(np.float32)
    lik = np.ascontiguousarray(np.concatenate([sc.scan_lik, far, sc.scan_lik[:200]], 0))
    engine.set_map(sc.map_xyz, sc.map_label, stamp=8302)
    engine.set_likelihood_params()
    d = engine.get_option("scan_order_device")
    out = {}
    try:
        for strict in (0, 1):
            engine.set_option("strict_order", strict)
            for where in (0, 1):
                engine.set_option("scan_order_device", where)
                out[(strict, where)] = engine.measure_batch(sc.poses, lik)
    finally:
        engine.set_option("scan_order_device", d)
        engine.set_option("strict_order", 2)
    for strict in (0, 1):
        for a, b in zip(out[(strict, 0)], out[(strict, 1)]):
            np.testing.assert_array_equal(a, b)
    assert np.count_nonzero(out[(0, 1)][1]) > 0
