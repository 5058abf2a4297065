"""The whole measurement update as ONE launch (mcl_3dl_amd/csrc/update_kernels.h: likelihood + beam + pf::measure, stage
hand-offs by arrival tickets) against the separate kernels: the same arithmetic in the same association — likelihoods, match
ratios, beam scores, normalised weights, entropy, ratio bounds and the restore rule are bit-identical — at the reference's own
operating sizes (parameters.h:68,98: 64 particles x 96 + 3 points by default) and around every boundary of the path."""
import numpy as np
import pytest
import torch

from mcl_3dl_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu
KEYS = ("lik", "quality", "beam", "weights")


def run_both(engine, sc, n_p, n_s, n_b, extra, lik_index=2, dist_weight=(1.0, 1.0, 5.0), stamp=8100, small_max=8192):
    engine.set_map(sc.map_xyz, sc.map_label, stamp=stamp, dist_weight=dist_weight)
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=max(n_b, 1))
    poses, w0 = sc.poses[:n_p], np.full(n_p, 1.0 / n_p, np.float32)
    ex = None
    if extra:
        ex = np.random.default_rng(n_p + n_s).uniform(0.05, 0.4, n_p).astype(np.float32)
    out = {}
    try:
        engine.set_option("lik_index", lik_index)
        engine.set_option("update_small_max", small_max)   # (the default, 512, is where the path stops paying)
        for one in (0, 1):
            engine.set_option("update_small", one)
            out[one] = engine.measure_update(poses, w0, sc.scan_lik[:n_s], sc.scan_beam[:n_b] if n_b else None,
                                             sc.scan_beam_label[:n_b] if n_b else None, sc.origins, extra=ex)
    finally:
        engine.set_option("update_small", 1)
        engine.set_option("update_small_max", 512)
        engine.set_option("lik_index", 2)
    return out[0], out[1]


def same(a, b):
    for k in KEYS:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a["restored"] == b["restored"]
    if not a["restored"]:
        assert a["entropy"] == b["entropy"]
    assert a["match_ratio_min"] == b["match_ratio_min"] and a["match_ratio_max"] == b["match_ratio_max"]


@pytest.fixture(scope="module")
def scene():
    return make_scene(n=91, n_p=4400, n_s=2000, n_b=256, seed=77)


@pytest.mark.parametrize("n_p,n_s,n_b", [(64, 96, 3), (64, 1000, 0), (64, 1000, 32), (1, 96, 3), (255, 128, 7), (256, 129, 0),
                                         (257, 300, 40), (500, 300, 96), (1000, 512, 0), (100, 700, 256), (4096, 96, 3),
                                         (4400, 64, 5), (3, 1023, 0), (70, 2, 1)])
@pytest.mark.parametrize("extra", [False, True])
def test_one_launch_equals_the_separate_kernels(engine, scene, n_p, n_s, n_b, extra):
    a, b = run_both(engine, scene, n_p, n_s, n_b, extra)
    same(a, b)
    assert np.count_nonzero(b["lik"]) > 0


@pytest.mark.parametrize("lik_index", [0])
@pytest.mark.parametrize("n_p,n_s,n_b", [(64, 96, 3), (300, 400, 20)])
def test_other_indices(engine, scene, lik_index, n_p, n_s, n_b):
    a, b = run_both(engine, scene, n_p, n_s, n_b, True, lik_index=lik_index, stamp=8101)
    same(a, b)


def test_beam_only_and_empty_scans(engine, scene):
    """n_s = 0 -> likelihood (1, 0) (likelihood.cpp:111-114); n_b = 0 -> beam 1 (beam.cpp:130-133)."""
    a, b = run_both(engine, scene, 200, 0, 48, True, stamp=8102)
    same(a, b)
    assert (b["lik"] == 1.0).all() and (b["quality"] == 0.0).all() and len(np.unique(b["beam"])) > 1


def test_restore_rule_and_both_penalty_modes(engine, scene):
    sc = scene
    engine.set_map(sc.map_xyz, sc.map_label, stamp=8103, dist_weight=(1.0, 1.0, 1.0))
    engine.set_likelihood_params()
    far = sc.poses[:300].copy()
    far[:, :3] += 400.0
    w0 = np.full(300, 1 / 300, np.float32)
    for short_only in (True, False):
        engine.set_beam_params(num_points=16, add_penalty_short_only_mode=short_only)
        out = {}
        try:
            for one in (0, 1):
                engine.set_option("update_small", one)
                out[one] = (engine.measure_update(far, w0, sc.scan_lik[:200], sc.scan_beam[:16], sc.scan_beam_label[:16], sc.origins),
                            engine.measure_update(sc.poses[:300], w0, sc.scan_lik[:200], sc.scan_beam[:16], sc.scan_beam_label[:16],
                                                  sc.origins))
        finally:
            engine.set_option("update_small", 1)
        assert out[1][0]["restored"] is True
        np.testing.assert_array_equal(out[1][0]["weights"], w0)   # pf.h:274-278: weights untouched
        same(out[0][0], out[1][0])
        same(out[0][1], out[1][1])


def test_device_resident_entry_point_back_to_back_launches(engine, scene):
    """mcl3dl_hip_update_device: repeated launches reuse the arrival tickets (left at zero by the last work-group)."""
    sc = scene
    n_p, n_s, n_b = 700, 96, 3
    engine.set_map(sc.map_xyz, sc.map_label, stamp=8104, dist_weight=(1.0, 1.0, 1.0))
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=n_b)
    engine.upload_scan(sc.scan_lik[:n_s], sc.scan_beam[:n_b], sc.scan_beam_label[:n_b], sc.origins)
    dev = torch.device("cuda", 0)
    d_pose = torch.from_numpy(np.ascontiguousarray(sc.poses[:n_p])).to(dev)
    w0 = torch.full((n_p,), 1.0 / n_p, device=dev)
    res = {}
    try:
        for tag, one in (("split", 0), ("one", 1)):
            engine.set_option("update_small", one)
            d_w = w0.clone()
            d_lik, d_ratio, d_beam = (torch.empty(n_p, device=dev) for _ in range(3))
            d_stats = torch.zeros(4, device=dev)
            for _ in range(5):
                d_w.copy_(w0)
                torch.cuda.synchronize()  # the engine runs on its own stream
                engine.update_device(d_pose, n_p, d_w, d_stats, d_lik=d_lik, d_ratio=d_ratio, d_beam=d_beam)
                engine.synchronize()
            res[tag] = [t.cpu().numpy().copy() for t in (d_w, d_lik, d_ratio, d_beam, d_stats)]
    finally:
        engine.set_option("update_small", 1)
    for tag in ("one",):
        for x, y in zip(res["split"], res[tag]):
            np.testing.assert_array_equal(x, y, err_msg=tag)


def test_sizes_outside_the_path_keep_the_separate_kernels(engine, scene):
    """Tiled scans (>= 1024 points), small-scan form (<= 32 points x >= 256 particles) and more than update_small_max
    particles are not eligible: same results with the option on or off by construction; the option is still honoured."""
    for n_p, n_s in ((64, 2000), (300, 16), (4400, 96)):
        a, b = run_both(engine, scene, n_p, n_s, 0, False, stamp=8105, small_max=4096)
        same(a, b)
    assert engine.get_option("update_small_max") == 512 and engine.get_option("update_small") == 1
