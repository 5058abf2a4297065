"""mcl3dl_hip_measure_update with its head and tail as one launch each (mcl_3dl_amd/csrc/stage_kernels.h) against the general
path (upload + seven ordering launches + lik_finalize / pf_partial / pf_reduce / pf_apply + one D2H): the same keys, the same
stable order, the same association of every sum — so likelihoods, match ratios, beam scores, normalised weights, entropy, ratio
bounds and the restore rule (include/mcl_3dl/pf.h:252-279) are bit-identical, whether the arrays are read in place from
page-locked memory (update_zero_copy = 1), mirrored by one H2D copy (0) or belong to the caller's own page-locked block."""
import numpy as np
import pytest
import torch

from mcl_3dl_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu
KEYS = ("lik", "quality", "beam", "weights")


def same(a, b):
    for k in KEYS:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a["restored"] == b["restored"]
    if not a["restored"]:
        assert a["entropy"] == b["entropy"]
    assert a["match_ratio_min"] == b["match_ratio_min"] and a["match_ratio_max"] == b["match_ratio_max"]


@pytest.fixture(scope="module")
def scene():
    return make_scene(n=91, n_p=9000, n_s=16384, n_b=2048, seed=4242)


def configure(engine, sc, n_b, stamp, dist_weight=(1.0, 1.0, 5.0)):
    engine.set_map(sc.map_xyz, sc.map_label, stamp=stamp, dist_weight=dist_weight)
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=max(n_b, 1))


def run_modes(engine, sc, n_p, n_s, n_b, extra, modes, origins=None, beam_label=None, scan_lik=None):
    poses, w0 = sc.poses[:n_p], np.full(n_p, 1.0 / n_p, np.float32)
    ex = np.random.default_rng(n_p + n_s).uniform(0.05, 0.4, n_p).astype(np.float32) if extra else None
    lik = sc.scan_lik[:n_s] if scan_lik is None else scan_lik
    org = sc.origins if origins is None else origins
    lab = (sc.scan_beam_label[:n_b] if beam_label is None else beam_label) if n_b else None
    out = []
    try:
        for stage, zero_copy in modes:
            engine.set_option("update_stage", stage)
            engine.set_option("update_zero_copy", zero_copy)
            out.append(engine.measure_update(poses, w0, lik, sc.scan_beam[:n_b] if n_b else None, lab, org, extra=ex))
    finally:
        engine.set_option("update_stage", 1)
        engine.set_option("update_zero_copy", 1)
    return out


ALL_MODES = [(0, 0), (1, 1), (1, 0)]


@pytest.mark.parametrize("n_p,n_s,n_b", [(64, 96, 3), (64, 1000, 32), (700, 300, 0), (513, 2048, 40), (1024, 2049, 0),
                                         (1025, 4096, 7), (4096, 1000, 96), (2000, 8192, 512), (2049, 8193, 0),
                                         (600, 16384, 2048), (700, 2048, 2049), (650, 2049, 2048), (8192, 1024, 3),
                                         (8193, 1500, 3), (1, 96, 3), (3, 1, 1), (4096, 96, 3), (2000, 500, 20), (5000, 700, 0),
                                         (8192, 64, 32), (1500, 767, 33),
                                         (4096, 0, 48), (300, 5, 0)])
@pytest.mark.parametrize("extra", [False, True])
def test_staged_update_equals_the_general_path(engine, scene, n_p, n_s, n_b, extra):
    configure(engine, scene, n_b, stamp=9100 + n_b)
    res = run_modes(engine, scene, n_p, n_s, n_b, extra, ALL_MODES)
    for r in res[1:]:
        same(res[0], r)
    assert np.count_nonzero(res[0]["weights"]) > 0


def test_full_c2_scan_with_all_particles(engine, scene):
    """16 384 points (stage_pack_kernel + the chip-wide sort) x 9000 particles."""
    configure(engine, scene, 0, stamp=9200, dist_weight=(1.0, 1.0, 1.0))
    res = run_modes(engine, scene, 9000, 16384, 0, True, [(0, 0), (1, 1), (1, 0)])
    same(res[0], res[1])
    same(res[0], res[2])
    res = run_modes(engine, scene, 4096, 16384, 0, False, [(0, 0), (1, 1)])
    same(res[0], res[1])


def test_non_finite_points_and_several_origins(engine, scene):
    """Points with NaN / inf coordinates keep their place in the order both paths give them (they never match: the reference
    scores them 0); beam points of several origins are keyed by the range from THEIR origin."""
    sc = scene
    n_s, n_b, n_p = 3000, 300, 500
    lik = sc.scan_lik[:n_s].copy()
    lik[5] = (np.nan, 0.0, 0.0)
    lik[77, 1] = np.inf
    lik[2999, 2] = -np.inf
    lik[1500] = (np.nan, np.nan, np.nan)
    origins = np.array([[0.0, 0.0, 0.5], [0.3, -0.2, 0.6], [-0.4, 0.1, 0.4]], np.float32)
    lab = (np.arange(n_b) % 3).astype(np.uint32)
    configure(engine, sc, n_b, stamp=9300)
    res = run_modes(engine, sc, n_p, n_s, n_b, True, ALL_MODES, origins=origins, beam_label=lab, scan_lik=lik)
    for r in res[1:]:
        same(res[0], r)


def test_bad_origin_is_refused(engine, scene):
    sc = scene
    configure(engine, sc, 8, stamp=9301)
    lab = np.zeros(8, np.uint32)
    lab[3] = 7
    from mcl_3dl_amd.capi import EngineError
    with pytest.raises(EngineError, match="origin"):
        engine.measure_update(sc.poses[:10], np.full(10, 0.1, np.float32), sc.scan_lik[:50], sc.scan_beam[:8], lab, sc.origins)


def test_restore_rule(engine, scene):
    """Every particle far from the map: all likelihoods 0 -> weights restored (pf.h:274-278), entropy NaN, on both paths."""
    sc = scene
    configure(engine, sc, 0, stamp=9302)
    far = sc.poses[:700].copy()
    far[:, :3] += 500.0
    w0 = np.random.default_rng(3).uniform(0.1, 1.0, 700).astype(np.float32)
    outs = []
    try:
        for stage in (0, 1):
            engine.set_option("update_stage", stage)
            outs.append(engine.measure_update(far, w0, sc.scan_lik[:600], None, None, sc.origins))
    finally:
        engine.set_option("update_stage", 1)
    for o in outs:
        assert o["restored"]
        np.testing.assert_array_equal(o["weights"], w0)
    same(outs[0], outs[1])


def test_arrays_in_the_callers_page_locked_block(engine, scene):
    """Poses, weights, scans and the result arrays allocated with mcl3dl_hip_host_alloc are read and written in place."""
    sc = scene
    n_p, n_s, n_b = 3000, 5000, 64
    configure(engine, sc, n_b, stamp=9400)
    ref = run_modes(engine, sc, n_p, n_s, n_b, False, [(0, 0)])[0]
    poses = engine.host_array((n_p, 7))
    w = engine.host_array(n_p)
    lik_xyz = engine.host_array((n_s, 3))
    beam_xyz = engine.host_array((n_b, 3))
    beam_lab = engine.host_array(n_b, np.uint32)
    org = engine.host_array((len(sc.origins), 3))
    out_lik, out_ratio, out_beam = engine.host_array(n_p), engine.host_array(n_p), engine.host_array(n_p)
    poses[:] = sc.poses[:n_p]
    lik_xyz[:] = sc.scan_lik[:n_s]
    beam_xyz[:] = sc.scan_beam[:n_b]
    beam_lab[:] = sc.scan_beam_label[:n_b]
    org[:] = sc.origins
    for rep in range(3):
        w[:] = 1.0 / n_p
        out_lik[:] = -1.0
        ent, rmin, rmax, restored = engine.measure_update_into(poses, w, lik_xyz, beam_xyz, beam_lab, org, out_lik, out_ratio,
                                                               out_beam)
        np.testing.assert_array_equal(w, ref["weights"])
        np.testing.assert_array_equal(out_lik, ref["lik"])
        np.testing.assert_array_equal(out_ratio, ref["quality"])
        np.testing.assert_array_equal(out_beam, ref["beam"])
        assert ent == ref["entropy"] and rmin == ref["match_ratio_min"] and rmax == ref["match_ratio_max"] and not restored
    # mixed: some arrays page-locked, some not
    w2 = np.full(n_p, 1.0 / n_p, np.float32)
    lik2 = np.zeros(n_p, np.float32)
    engine.measure_update_into(poses, w2, np.ascontiguousarray(sc.scan_lik[:n_s]), beam_xyz, beam_lab, org, lik2, out_ratio, None)
    np.testing.assert_array_equal(w2, ref["weights"])
    np.testing.assert_array_equal(lik2, ref["lik"])
    for a in (poses, w, lik_xyz, beam_xyz, beam_lab, org, out_lik, out_ratio, out_beam):
        engine.host_free(a)
