"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP engine, called through the C ABI, against the CPU oracle
on the same seeded inputs.

Tolerances (BASELINE.json north_star: "weights within 1e-5 relative of the CPU path"):
  * likelihood score      rtol 1e-5 vs the reference-order float oracle (the GPU sums the same bit-identical float
                          terms in fp64, so the only difference is the reference's own float-summation rounding)
  * match ratio (quality) exact (integer count / N_s)
  * beam score            exact (integer penalty count -> float product table)
  * beam status / hit id  exact
  * normalised weights, entropy  rtol 1e-5
"""
import numpy as np
import pytest

from mcl_3dl_amd.synthetic import make_scene
from oracle import pyoracle

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def make_oracle(kind, sc, dist_weight, beam_kw=None, lik_kw=None):
    o = pyoracle.Oracle(kind)
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=dist_weight)
    o.set_likelihood_params(pyoracle.LikelihoodParams(**(lik_kw or {})))
    o.set_beam_params(pyoracle.BeamParams(**(beam_kw or {})))
    return o


def setup_engine(eng, sc, dist_weight, stamp, beam_kw=None, lik_kw=None):
    eng.set_map(sc.map_xyz, sc.map_label, stamp=stamp, dist_weight=dist_weight)
    eng.set_likelihood_params(**(lik_kw or {}))
    kw = dict(beam_kw or {})
    eng.set_beam_params(**kw)


@pytest.fixture(scope="module")
def scene_c1():
    return make_scene(n=91, n_p=64, n_s=1000, n_b=96)


@pytest.mark.parametrize("dist_weight", [(1.0, 1.0, 1.0), (1.0, 1.0, 5.0), None])
def test_likelihood_c1(engine, oracle_kind, scene_c1, dist_weight):
    """BASELINE config C1 (64 particles x 1k points, 50k-pt cube map), unit and default (1,1,5) dist_weight."""
    sc = scene_c1
    setup_engine(engine, sc, dist_weight, stamp=11)
    lik, ratio, _ = engine.measure_batch(sc.poses, sc.scan_lik)
    o = make_oracle(oracle_kind, sc, dist_weight)
    want_lik, want_q = o.likelihood_measure(sc.poses, sc.scan_lik)
    assert np.all(want_lik > 0)
    np.testing.assert_allclose(lik, want_lik, rtol=RTOL)
    np.testing.assert_array_equal(ratio, want_q)


@pytest.mark.parametrize("short_only", [True, False])
@pytest.mark.parametrize("n_b", [3, 96])
def test_beam_c1(engine, oracle_kind, scene_c1, short_only, n_b):
    sc = scene_c1
    kw = dict(num_points=n_b, add_penalty_short_only_mode=short_only)
    setup_engine(engine, sc, (1.0, 1.0, 1.0), stamp=12, beam_kw=kw)
    sb, sl = sc.scan_beam[:n_b], sc.scan_beam_label[:n_b]
    _, _, beam = engine.measure_batch(sc.poses, None, sb, sl, sc.origins)
    o = make_oracle(oracle_kind, sc, (1.0, 1.0, 1.0), beam_kw=kw)
    want, want_q = o.beam_measure(sc.poses, sb, sl, sc.origins)
    np.testing.assert_array_equal(beam, want)
    assert np.all(want_q == 1.0)
    assert len(np.unique(want)) > 1  # the case must exercise different penalty counts


def test_beam_status_random_rays(engine, oracle_kind, scene_c1):
    """getBeamStatus for explicit rays: status and collided map point identical to the reference."""
    sc = scene_c1
    setup_engine(engine, sc, (1.0, 1.0, 1.0), stamp=13, beam_kw=dict(num_points=8))
    rng = np.random.default_rng(7)
    half = 91 * 0.1 / 2
    begin = rng.uniform(-half * 1.1, half * 1.1, (4000, 3)).astype(np.float32)
    end = (begin + rng.normal(0, 2.0, (4000, 3))).astype(np.float32)
    st, hit = engine.beam_status(begin, end)
    o = make_oracle(oracle_kind, sc, (1.0, 1.0, 1.0), beam_kw=dict(num_points=8))
    want_st, want_hit = o.beam_status(begin, end)
    np.testing.assert_array_equal(st, want_st)
    np.testing.assert_array_equal(hit, want_hit)
    assert set(np.unique(want_st)) >= {0, 1, 2}


def test_beam_label_filter(engine, oracle_kind):
    """Hits on map points with label > filter_label_max are ignored and the ray continues (beam.cpp:168-169)."""
    sc = make_scene(n=61, n_p=32, n_s=64, n_b=64, label_wall=2)
    kw = dict(num_points=64, filter_label_max=1)
    setup_engine(engine, sc, (1.0, 1.0, 1.0), stamp=14, beam_kw=kw)
    _, _, beam = engine.measure_batch(sc.poses, None, sc.scan_beam, sc.scan_beam_label, sc.origins)
    o = make_oracle(oracle_kind, sc, (1.0, 1.0, 1.0), beam_kw=kw)
    want, _ = o.beam_measure(sc.poses, sc.scan_beam, sc.scan_beam_label, sc.origins)
    np.testing.assert_array_equal(beam, want)
    kw2 = dict(num_points=64)
    o2 = make_oracle(oracle_kind, sc, (1.0, 1.0, 1.0), beam_kw=kw2)
    unfiltered, _ = o2.beam_measure(sc.poses, sc.scan_beam, sc.scan_beam_label, sc.origins)
    assert not np.array_equal(unfiltered, want)  # the filter changes the answer in this scene


def test_measure_update_c1(engine, oracle_kind, scene_c1):
    """The whole `pf_->measure(measure_func)` statement (src/mcl_3dl.cpp:398-426): beam x likelihood x odom factor,
    normalisation, entropy, match-ratio min/max."""
    sc = scene_c1
    dw = (1.0, 1.0, 5.0)
    kw = dict(num_points=96)
    setup_engine(engine, sc, dw, stamp=15, beam_kw=kw)
    sigma = 1.0
    # NormalLikelihood<float>(sigma)(|odom_err|), include/mcl_3dl/nd.h:41-58, computed by the caller
    a = np.float32(1.0 / np.sqrt(2.0 * np.pi * sigma * sigma))
    sq2 = np.float32(sigma * sigma * 2.0)
    x = np.sqrt((sc.odom_err[:, 0] * sc.odom_err[:, 0] + sc.odom_err[:, 1] * sc.odom_err[:, 1])
                + sc.odom_err[:, 2] * sc.odom_err[:, 2]).astype(np.float32)
    extra = (a * np.exp((-x * x / sq2).astype(np.float32)).astype(np.float32)).astype(np.float32)
    rng = np.random.default_rng(3)
    w0 = rng.uniform(0.5, 1.5, len(sc.poses)).astype(np.float32)
    w0 /= w0.sum()
    got = engine.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, extra=extra)
    o = make_oracle(oracle_kind, sc, dw, beam_kw=kw)
    want = o.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins,
                            odom_err=sc.odom_err, odom_sigma=sigma)
    np.testing.assert_allclose(got["lik"], want["lik"], rtol=RTOL)
    np.testing.assert_array_equal(got["beam"], want["beam"])
    np.testing.assert_array_equal(got["quality"], want["quality"])
    np.testing.assert_allclose(got["weights"], want["weights"], rtol=RTOL)
    np.testing.assert_allclose(got["entropy"], want["entropy"], rtol=RTOL)
    assert got["match_ratio_min"] == want["match_ratio_min"]
    assert got["match_ratio_max"] == want["match_ratio_max"]
    assert got["restored"] == want["restored"] is False
    np.testing.assert_allclose(got["weights"].sum(dtype=np.float64), 1.0, rtol=1e-6)


def test_empty_scans(engine, oracle_kind, scene_c1):
    """Null/empty cloud -> (likelihood 1, quality 0) for both models (likelihood.cpp:111-114, beam.cpp:130-133)."""
    sc = scene_c1
    setup_engine(engine, sc, (1.0, 1.0, 1.0), stamp=16)
    lik, ratio, beam = engine.measure_batch(sc.poses, np.zeros((0, 3), np.float32))
    assert np.all(lik == 1.0) and np.all(ratio == 0.0) and np.all(beam == 1.0)
    o = make_oracle(oracle_kind, sc, (1.0, 1.0, 1.0))
    wl, wq = o.likelihood_measure(sc.poses, np.zeros((0, 3), np.float32))
    np.testing.assert_array_equal(lik, wl)
    np.testing.assert_array_equal(ratio, wq)


def test_far_away_particles_restore(engine, oracle_kind, scene_c1):
    """Particles far outside the map score 0; if every particle scores 0 the weights are restored (pf.h:274-278)."""
    sc = scene_c1
    setup_engine(engine, sc, (1.0, 1.0, 1.0), stamp=17)
    poses = sc.poses.copy()
    poses[:, :3] += 500.0
    got = engine.measure_update(poses, sc.weights, sc.scan_lik)
    assert np.all(got["lik"] == 0.0) and np.all(got["quality"] == 0.0)
    assert got["restored"] is True
    np.testing.assert_array_equal(got["weights"], sc.weights)
    o = make_oracle(oracle_kind, sc, (1.0, 1.0, 1.0))
    want = o.measure_update(poses, sc.weights, sc.scan_lik, np.zeros((0, 3), np.float32), None, sc.origins)
    assert want["restored"] is True
    np.testing.assert_array_equal(got["weights"], want["weights"])
    # mixed: half the particles far away -> their weights become exactly 0, the rest normalise
    poses2 = sc.poses.copy()
    poses2[::2, :3] += 500.0
    got2 = engine.measure_update(poses2, sc.weights, sc.scan_lik)
    want2 = o.measure_update(poses2, sc.weights, sc.scan_lik, np.zeros((0, 3), np.float32), None, sc.origins)
    assert np.all(got2["weights"][::2] == 0.0)
    np.testing.assert_allclose(got2["weights"], want2["weights"], rtol=RTOL)
    assert got2["match_ratio_min"] == want2["match_ratio_min"] == 0.0


def test_unnormalised_quaternions(engine, oracle_kind, scene_c1):
    """transform() normalises rot_ (state_6dof.h:217) but the beam origin uses the raw rot_ (beam.cpp:145)."""
    sc = scene_c1
    kw = dict(num_points=96)
    setup_engine(engine, sc, (1.0, 1.0, 1.0), stamp=18, beam_kw=kw)
    poses = sc.poses.copy()
    poses[:, 3:] *= np.linspace(0.7, 1.4, len(poses), dtype=np.float32)[:, None]
    lik, ratio, beam = engine.measure_batch(poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
    o = make_oracle(oracle_kind, sc, (1.0, 1.0, 1.0), beam_kw=kw)
    wl, wq = o.likelihood_measure(poses, sc.scan_lik)
    wb, _ = o.beam_measure(poses, sc.scan_beam, sc.scan_beam_label, sc.origins)
    np.testing.assert_allclose(lik, wl, rtol=RTOL)
    np.testing.assert_array_equal(ratio, wq)
    np.testing.assert_array_equal(beam, wb)


def test_ragged_sizes(engine, oracle_kind, scene_c1):
    """Scan sizes that are not multiples of the work-group / wavefront size, single particle, single point."""
    sc = scene_c1
    setup_engine(engine, sc, (1.0, 1.0, 1.0), stamp=19, beam_kw=dict(num_points=5))
    o = make_oracle(oracle_kind, sc, (1.0, 1.0, 1.0), beam_kw=dict(num_points=5))
    for n_p, n_s, n_b in [(1, 1, 1), (3, 63, 2), (7, 129, 5), (64, 257, 5), (5, 999, 0)]:
        lik, ratio, beam = engine.measure_batch(sc.poses[:n_p], sc.scan_lik[:n_s], sc.scan_beam[:n_b],
                                                sc.scan_beam_label[:n_b], sc.origins)
        wl, wq = o.likelihood_measure(sc.poses[:n_p], sc.scan_lik[:n_s])
        wb, _ = o.beam_measure(sc.poses[:n_p], sc.scan_beam[:n_b], sc.scan_beam_label[:n_b], sc.origins)
        np.testing.assert_allclose(lik, wl, rtol=RTOL)
        np.testing.assert_array_equal(ratio, wq)
        np.testing.assert_array_equal(beam, wb)


def test_likelihood_params_change(engine, oracle_kind, scene_c1):
    """refreshParameters(): a different search radius rebuilds the cell index."""
    sc = scene_c1
    lk = dict(match_dist_min=0.35, match_dist_flat=0.1, match_weight=2.5)
    setup_engine(engine, sc, (1.0, 1.0, 2.0), stamp=20, lik_kw=lk)
    lik, ratio, _ = engine.measure_batch(sc.poses, sc.scan_lik)
    o = make_oracle(oracle_kind, sc, (1.0, 1.0, 2.0), lik_kw=lk)
    wl, wq = o.likelihood_measure(sc.poses, sc.scan_lik)
    np.testing.assert_allclose(lik, wl, rtol=RTOL)
    np.testing.assert_array_equal(ratio, wq)
    engine.set_likelihood_params()  # back to the defaults for later tests


@pytest.mark.parametrize("dist_weight", [(1.0, 1.0, 1.0), (1.0, 1.0, 5.0)])
@pytest.mark.parametrize("ratio,phase", [(0.5, 0.5), (0.5, 0.0), (0.25, 0.5), (1.0, 0.3)])
@pytest.mark.parametrize("mode", [2])
def test_candidate_index_equals_cell_scan_bit_for_bit(engine, scene_c1, dist_weight, ratio, phase, mode):
    """The pruned candidate-voxel index (map_compiler.h) must return exactly the nearest distance the 27-cell scan of
    the whole neighbourhood returns: identical per-point terms, identical fp64 sums, identical floats."""
    sc = scene_c1
    rng = np.random.default_rng(11)
    # poses spread far wider than the tracking case so queries land at every offset from the surfaces, incl. off-map
    poses = sc.poses.copy()
    poses[:, :3] += rng.normal(0, 0.6, (len(poses), 3)).astype(np.float32)
    setup_engine(engine, sc, dist_weight, stamp=30)
    try:
        engine.set_option("lik_index", 0)
        lik0, ratio0, _ = engine.measure_batch(poses, sc.scan_lik)
        engine.set_option("lik_index", mode)
        engine.set_option("cand_voxel_ratio", ratio)
        engine.set_option("cand_phase", phase)
        lik1, ratio1, _ = engine.measure_batch(poses, sc.scan_lik)
        st = engine.index_stats()
        assert 0 < st["candidates"] <= st["preliminary"]
    finally:
        engine.set_option("lik_index", 2)
        engine.set_option("cand_voxel_ratio", 0.0)
        engine.set_option("cand_phase", 0.5)
    np.testing.assert_array_equal(lik1, lik0)
    np.testing.assert_array_equal(ratio1, ratio0)
    assert np.count_nonzero(lik0) > len(poses) // 2


@pytest.mark.parametrize("group", [0, 4, 8, 16, 32])
@pytest.mark.parametrize("mode", [0, 2])
def test_tiled_kernel_matches_per_particle_kernel(engine, oracle_kind, group, mode):
    """The tile-major XCD-aware kernel evaluates the same bit-identical terms as the per-particle kernel; only the
    fp64 summation order differs (<= 1 float ulp after rounding). Ragged sizes: 1500 points (5.86 tiles), 100 particles."""
    sc = make_scene(n=91, n_p=100, n_s=1500, seed=5)
    dw = (1.0, 1.0, 3.0)
    setup_engine(engine, sc, dw, stamp=40)
    try:
        engine.set_option("lik_index", mode)
        engine.set_option("lik_tiled", 0)
        lik0, ratio0, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        engine.set_option("lik_tiled", 1)
        engine.set_option("lik_group", group)
        lik1, ratio1, _ = engine.measure_batch(sc.poses, sc.scan_lik)
    finally:
        engine.set_option("lik_index", 2)
        engine.set_option("lik_tiled", 1)
        engine.set_option("lik_group", 0)
    np.testing.assert_array_equal(ratio1, ratio0)
    np.testing.assert_allclose(lik1, lik0, rtol=1.2e-7)
    o = make_oracle(oracle_kind, sc, dw)
    wl, wq = o.likelihood_measure(sc.poses, sc.scan_lik)
    np.testing.assert_allclose(lik1, wl, rtol=RTOL)
    np.testing.assert_array_equal(ratio1, wq)


@pytest.mark.parametrize("shape", ["tiled4", "tiled8", "tiled16", "tiled32", "per_particle", "small"])
@pytest.mark.parametrize("n_p", [100, 37])
@pytest.mark.parametrize("parts", [4, 8])
def test_cooperative_record_fetch_is_bit_identical(engine, shape, n_p, parts):
    """lik_coop = 1 (the four lanes of a quad fetch each 64-byte record together and split its candidates; VALU-trimmed
    transform and sqrt) against lik_coop = 0 (every lane fetches its own record): the SAME terms summed in the SAME
    order, so every kernel family gives equal results bit for bit — also with strict_order, with ragged particle groups
    (37 = 2 x 16 + 5), a ragged last tile (1500 = 5 x 256 + 220), particles outside the map (lanes that discard their
    loads), and voxels with more than four candidates (overflow records: the clutter blob)."""
    n_s = 20 if shape == "small" else 1500
    n_part = 400 if shape == "small" else n_p
    sc = make_scene(n=91, n_p=n_part, n_s=n_s, seed=6, sigma_xyz=(1.5, 1.5, 0.4), sigma_rpy=(0.05, 0.05, 1.0))
    sc.poses[::7, 0] += 30.0  # far outside the grid
    rng = np.random.default_rng(3)
    blob = (sc.true_pose[:3] + np.array([2.0, 0.5, -1.45]) + rng.normal(0, 0.03, (3000, 3))).astype(np.float32)
    map_xyz = np.concatenate([sc.map_xyz, blob], 0)
    scan = np.concatenate([sc.scan_lik, (blob[:40] - sc.true_pose[:3]).astype(np.float32)], 0) if shape != "small" else sc.scan_lik
    res = {}
    d_coop = engine.get_option("lik_coop")
    try:
        engine.set_option("cand_record_parts", parts)   # 64-byte records (4 inline candidates) / 128-byte records (8)
        if shape.startswith("tiled"):
            engine.set_option("lik_group", int(shape[5:]))
        elif shape == "per_particle":
            engine.set_option("lik_tiled", 0)
        for dw in ((1.0, 1.0, 3.0), None):
            engine.set_map(map_xyz, None, stamp=41, dist_weight=dw)
            engine.set_likelihood_params()
            for strict in ((0, 1) if shape.startswith("tiled") else (0,)):
                engine.set_option("strict_order", strict)
                for coop in (0, 1):
                    engine.set_option("lik_coop", coop)
                    res[(dw, strict, coop)] = engine.measure_batch(sc.poses, scan)
            assert engine.index_stats()["record_parts"] == parts
    finally:
        engine.set_option("lik_coop", d_coop)
        engine.set_option("lik_group", 0)
        engine.set_option("lik_tiled", 1)
        engine.set_option("strict_order", 2)
        engine.set_option("cand_record_parts", 0)
    for (dw, strict, coop), r in res.items():
        if coop == 1:
            np.testing.assert_array_equal(r[0], res[(dw, strict, 0)][0])
            np.testing.assert_array_equal(r[1], res[(dw, strict, 0)][1])
    assert np.count_nonzero(res[(None, 0, 0)][0]) > n_part // 4
    assert engine.index_stats()["candidates"] > 0


@pytest.mark.parametrize("flat", [0.05, 0.0, 1e-21])
def test_trimmed_sqrt_at_tiny_and_zero_distances(engine, oracle_kind, flat):
    """lik_coop replaces sqrtf by v_sqrt_f32 + the two residual checks, without the compiler's input scaling below 2^-96:
    distances of exactly zero, denormal and sub-2^-96 squared distances must still give the reference's float."""
    rng = np.random.default_rng(11)
    cloud = rng.uniform(-1.0, 1.0, (4000, 3)).astype(np.float32)
    special = np.array([[0, 0, 0], [0.5, 0.25, -0.125]], np.float32)
    map_xyz = np.concatenate([special, cloud[np.abs(cloud).max(1) > 0.3]], 0)
    tiny = np.array([[0, 0, 0], [1e-20, 0, 0], [3e-23, 0, 0], [0, 1e-15, 0], [2e-16, 1e-16, 0], [0.5, 0.25, -0.125],
                     [0.5, 0.25 + 1.5e-8, -0.125], [1e-7, 0, 0], [0.01, 0.01, 0.01]], np.float32)
    scan = np.concatenate([np.repeat(tiny, 120, 0), rng.uniform(-1, 1, (1200, 3)).astype(np.float32)], 0)  # >= 1024 points: tiled
    poses = np.zeros((8, 7), np.float32)
    poses[:, 6] = 1.0
    poses[1:, :3] = rng.normal(0, 1e-3, (7, 3)).astype(np.float32)
    d_trim = engine.get_option("lik_coop")
    try:
        engine.set_map(map_xyz, None, stamp=43, dist_weight=None)
        engine.set_likelihood_params(match_dist_min=0.2, match_dist_flat=flat, match_weight=5.0)
        out = {}
        for trim in (0, 1):
            engine.set_option("lik_coop", trim)
            out[trim] = engine.measure_batch(poses, scan)
    finally:
        engine.set_option("lik_coop", d_trim)
        engine.set_likelihood_params()
    np.testing.assert_array_equal(out[1][0], out[0][0])
    np.testing.assert_array_equal(out[1][1], out[0][1])
    o = pyoracle.Oracle(oracle_kind)
    o.set_map(map_xyz, None, dist_weight=None)
    o.set_likelihood_params(pyoracle.LikelihoodParams(match_dist_min=0.2, match_dist_flat=flat, match_weight=5.0))
    wl, wq = o.likelihood_measure(poses, scan)
    np.testing.assert_allclose(out[1][0], wl, rtol=RTOL)
    np.testing.assert_array_equal(out[1][1], wq)


def test_sharded_update_protocol_on_one_gpu(engine, oracle_kind, scene_c1):
    """The multi-GPU protocol (particle shards, packed partials, ONE all-reduce(SUM), apply) with the collective emulated
    by a tensor add: two shards of unequal size must reproduce the unsharded update and the CPU reference."""
    import torch
    from mcl_3dl_amd.distributed import shard_bounds
    sc = scene_c1
    dev = torch.device("cuda", 0)
    kw = dict(num_points=96)
    setup_engine(engine, sc, (1.0, 1.0, 1.0), stamp=50, beam_kw=kw)
    engine.upload_scan(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
    n = len(sc.poses)
    rng = np.random.default_rng(8)
    w0 = rng.uniform(0.5, 1.5, n).astype(np.float32)
    w0 /= w0.sum()
    world = 3
    shards = []
    packed_sum = torch.zeros(2 + 2 * world, dtype=torch.float64, device=dev)
    for r in range(world):
        lo, hi = shard_bounds(n, world, r)
        m = hi - lo
        t = dict(pose=torch.from_numpy(sc.poses[lo:hi].copy()).to(dev), w=torch.from_numpy(w0[lo:hi].copy()).to(dev),
                 lik=torch.empty(m, device=dev), ratio=torch.empty(m, device=dev), beam=torch.empty(m, device=dev),
                 pack=torch.zeros(2 + 2 * world, dtype=torch.float64, device=dev), stats=torch.zeros(4, device=dev), m=m)
        engine.measure_device(t["pose"], m, t["lik"], t["ratio"], t["beam"])
        engine.pf_partial_device(t["w"], t["lik"], t["beam"], None, t["ratio"], m, t["pack"], rank=r, world=world)
        engine.synchronize()
        # each rank applies right after the collective in real use; the engine keeps w_new per call, so apply per shard
        # needs the reduced vector first: emulate by computing all partials, then re-running partial before each apply
        packed_sum += t["pack"]
        shards.append(t)
    got_w = []
    for r, t in enumerate(shards):
        engine.pf_partial_device(t["w"], t["lik"], t["beam"], None, t["ratio"], t["m"], t["pack"], rank=r, world=world)
        engine.pf_apply_device(t["w"], t["m"], packed_sum, t["stats"], world=world)
        engine.synchronize()
        got_w.append(t["w"].cpu().numpy())
    got_w = np.concatenate(got_w)
    stats = [t["stats"].cpu().numpy() for t in shards]
    o = make_oracle(oracle_kind, sc, (1.0, 1.0, 1.0), beam_kw=kw)
    want = o.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
    np.testing.assert_allclose(got_w, want["weights"], rtol=RTOL)
    for s in stats:  # every shard derives the same global statistics
        np.testing.assert_allclose(s[0], want["entropy"], rtol=RTOL)
        assert s[1] == np.float32(want["match_ratio_min"]) and s[2] == np.float32(want["match_ratio_max"])
        assert s[3] == 0.0
    whole = engine.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
    np.testing.assert_allclose(got_w, whole["weights"], rtol=1e-6)  # (the whole update adds its 64 weights as the reference does; the sharded protocol in fp64)


@pytest.mark.parametrize("dist_weight", [(1.0, 1.0, 1.0), (1.0, 1.0, 5.0)])
def test_strict_order_is_bit_identical_to_the_reference(engine, oracle_kind, dist_weight):
    """With the "strict_order" option the likelihood terms are added per particle, and the weights over the particles,
    as floats in the reference's own sequential order: likelihoods, match ratios, beam scores AND normalised weights then
    equal the reference's bit for bit (the entropy still goes through the device's log)."""
    sc = make_scene(n=91, n_p=200, n_s=3000, n_b=40, seed=31)
    kw = dict(num_points=40)
    setup_engine(engine, sc, dist_weight, stamp=60, beam_kw=kw)
    sigma = np.float32(0.8)
    nd_a = np.float32(1.0 / np.sqrt(2.0 * np.pi * float(sigma) * float(sigma)))  # NormalLikelihood(0), nd.h:46,51
    extra = np.full(len(sc.poses), nd_a, np.float32)
    rng = np.random.default_rng(4)
    w0 = rng.uniform(0.5, 1.5, len(sc.poses)).astype(np.float32)
    w0 /= w0.sum()
    try:
        engine.set_option("strict_order", 1)
        got = engine.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins,
                                    extra=extra)
    finally:
        engine.set_option("strict_order", 2)
    o = make_oracle(oracle_kind, sc, dist_weight, beam_kw=kw)
    want = o.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins,
                            odom_err=None, odom_sigma=float(sigma))
    np.testing.assert_array_equal(got["lik"], want["lik"])
    np.testing.assert_array_equal(got["quality"], want["quality"])
    np.testing.assert_array_equal(got["beam"], want["beam"])
    np.testing.assert_array_equal(got["weights"], want["weights"])
    np.testing.assert_allclose(got["entropy"], want["entropy"], rtol=1e-6)


@pytest.mark.parametrize("n_s", [1, 3, 8, 13, 32])
@pytest.mark.parametrize("mode", [0, 2])
def test_small_scan_kernel(engine, oracle_kind, n_s, mode):
    """Global-localisation shape: many particles, a handful of points each (likelihood.cpp:63-77: num_points_global = 8).
    The wavefront-sharing kernel gives the same floats as one work-group per particle, and matches the oracle."""
    sc = make_scene(n=91, n_p=1000, n_s=n_s, seed=17 + n_s, sigma_xyz=(1.5, 1.5, 0.2), sigma_rpy=(0.05, 0.05, 3.0))
    dw = (1.0, 1.0, 5.0)
    setup_engine(engine, sc, dw, stamp=70)
    try:
        engine.set_option("lik_index", mode)
        engine.set_option("lik_small", 0)
        lik0, ratio0, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        engine.set_option("lik_small", 1)
        lik1, ratio1, _ = engine.measure_batch(sc.poses, sc.scan_lik)
    finally:
        engine.set_option("lik_index", 2)
        engine.set_option("lik_small", 1)
    np.testing.assert_array_equal(ratio1, ratio0)
    np.testing.assert_allclose(lik1, lik0, rtol=1.2e-7)
    o = make_oracle(oracle_kind, sc, dw)
    wl, wq = o.likelihood_measure(sc.poses, sc.scan_lik)
    np.testing.assert_allclose(lik1, wl, rtol=RTOL)
    np.testing.assert_array_equal(ratio1, wq)


def test_voxel_edge_is_chosen_from_the_map_and_overflow_rounds_are_exact(engine, oracle_kind):
    """A map of voxel-filter centroids (lattice points displaced inside their voxel) fills most candidate voxels beyond the
    four slots of a record: the index then picks the smaller voxel edge, and what still overflows is fetched in further
    cooperative rounds. Results: equal to the non-cooperative kernel bit for bit, and to the reference."""
    sc = make_scene(n=91, n_p=64, n_s=1500, seed=12, map_jitter=0.045)
    dw = (1.0, 1.0, 1.0)
    d_coop = engine.get_option("lik_coop")
    out = {}
    try:
        for ratio in (0.0, 0.5):   # chosen per map / forced large voxels (more overflow rounds)
            engine.set_option("cand_voxel_ratio", ratio)
            setup_engine(engine, sc, dw, stamp=90)
            for coop in (0, 1):
                engine.set_option("lik_coop", coop)
                out[(ratio, coop)] = engine.measure_batch(sc.poses, sc.scan_lik)
            st = engine.index_stats()
            share = st["voxels_with_overflow"] / max(st["voxels_with_candidates"], 1)
            if ratio == 0.0:
                # the jittered map made the index shrink its voxels; the records stay 64-byte ones (packed w words) because
                # the tiled kernel queues its overflow rounds — with the queue switched off they widen to eight inline candidates
                assert abs(st["voxel_ratio"] - 0.36) < 1e-6 and st["record_parts"] == 4 and st["packed_words"] == 1, st
                assert st["voxels_over8"] < 0.05 * st["voxels_with_candidates"], st
                engine.set_option("lik_defer", 0)
                try:
                    setup_engine(engine, sc, dw, stamp=91)
                    out[("wide", 1)] = engine.measure_batch(sc.poses, sc.scan_lik)
                    assert engine.index_stats()["record_parts"] == 8
                finally:
                    engine.set_option("lik_defer", 1)
            else:
                assert abs(st["voxel_ratio"] - 0.5) < 1e-6 and share > 0.25 and st["record_parts"] == 4, st
    finally:
        engine.set_option("cand_voxel_ratio", 0.0)
        engine.set_option("lik_coop", d_coop)
    ref = out[(0.5, 0)]
    for k, v in out.items():
        np.testing.assert_array_equal(v[0], ref[0])
        np.testing.assert_array_equal(v[1], ref[1])
    o = make_oracle(oracle_kind, sc, dw)
    wl, wq = o.likelihood_measure(sc.poses, sc.scan_lik)
    np.testing.assert_allclose(ref[0], wl, rtol=RTOL)
    np.testing.assert_array_equal(ref[1], wq)
    # a lattice map keeps the large voxels
    sc2 = make_scene(n=91, n_p=8, n_s=64, seed=12)
    setup_engine(engine, sc2, dw, stamp=91)
    engine.measure_batch(sc2.poses, sc2.scan_lik)
    st = engine.index_stats()
    assert abs(st["voxel_ratio"] - 0.5) < 1e-6 and st["record_parts"] == 4   # ... and the 64-byte records


def test_prepared_beam_origins_are_bit_identical(engine, oracle_kind):
    """Launches of >= 32 768 rays take the per-(particle, origin) constants (normalised quaternion, begin point, its voxel,
    the within-map flag) from beam_origin_kernel instead of every ray recomputing them: same expressions, same scores —
    with three origins, unnormalised quaternions, particles whose sensor origin lies outside the map, both penalty modes."""
    sc = make_scene(n=91, n_p=300, n_s=64, n_b=160, seed=77, sigma_xyz=(1.0, 1.0, 0.3), sigma_rpy=(0.05, 0.05, 1.0))
    sc.poses[::9, 0] += 25.0          # begin outside the map: LONG for every ray
    sc.poses[1::5, 3:] *= np.float32(1.7)   # raw quaternion not of unit length (beam.cpp:145 rotates the origin with it)
    origins = np.array([[0, 0, 0.5], [0.3, -0.1, 0.4], [-0.2, 0.2, 0.8]], np.float32)
    origin_id = (np.arange(len(sc.scan_beam)) % 3).astype(np.uint32)
    engine.set_map(sc.map_xyz, sc.map_label, stamp=9100, dist_weight=None)
    res = {}
    try:
        for short_only in (True, False):
            engine.set_beam_params(num_points=160, add_penalty_short_only_mode=short_only)
            for prep in (1, 0):
                engine.set_option("beam_prepare", prep)
                res[(short_only, prep)] = engine.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, origin_id, origins)[2]
    finally:
        engine.set_option("beam_prepare", 1)
    for short_only in (True, False):
        np.testing.assert_array_equal(res[(short_only, 1)], res[(short_only, 0)])
    o = make_oracle(oracle_kind, sc, None, beam_kw=dict(num_points=160, add_penalty_short_only_mode=False))
    want, _ = o.beam_measure(sc.poses, sc.scan_beam, origin_id, origins)
    np.testing.assert_array_equal(res[(False, 1)], want)
    assert len(np.unique(want)) > 3


@pytest.mark.parametrize("n_s,n_p", [(2700, 300), (800, 300), (4096, 37), (2048 + 256, 1000), (700, 5)])
def test_tile_to_xcd_mapping_covers_every_pair_once(engine, oracle_kind, n_s, n_p):
    """blockIdx -> (tile, particle group): the tiles of the largest multiple of eight are interleaved over the XCDs, the
    remaining n_tiles % 8 tiles are shared out by particle group. Shapes with both parts (11 and 9 tiles), only the second
    (4 and 3 tiles, the 768-point rule for >= 256 particles), only the first (16 tiles), ragged particle groups and a ragged
    last tile: match ratios equal the per-particle kernel's, likelihoods to one float rounding, both within the gate of the
    reference."""
    sc = make_scene(n=91, n_p=n_p, n_s=n_s, seed=7 + n_s, sigma_xyz=(0.6, 0.6, 0.1), sigma_rpy=(0.03, 0.03, 0.4))
    dw = (1.0, 1.0, 2.0)
    setup_engine(engine, sc, dw, stamp=700 + n_s)
    try:
        engine.set_option("lik_tiled", 0)
        lik0, ratio0, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        engine.set_option("lik_tiled", 1)
        engine.set_option("lik_tiled_min", 512)
        lik1, ratio1, _ = engine.measure_batch(sc.poses, sc.scan_lik)
    finally:
        engine.set_option("lik_tiled", 1)
        engine.set_option("lik_tiled_min", 1024)
    np.testing.assert_array_equal(ratio1, ratio0)
    np.testing.assert_allclose(lik1, lik0, rtol=1.2e-7)
    k = min(n_p, 64)
    o = make_oracle(oracle_kind, sc, dw)
    wl, wq = o.likelihood_measure(sc.poses[:k], sc.scan_lik)
    np.testing.assert_allclose(lik1[:k], wl, rtol=RTOL)
    np.testing.assert_array_equal(ratio1[:k], wq)
    assert np.count_nonzero(wq) == k
