"""pf::measure as ONE kernel (pf_fused_kernel, <= 1024 particles on one GPU; both forms then add the weights in the reference's float
order — float_chain.h — so the comparison covers that recurrence too) against the partial + reduce + apply form:
the same arithmetic in the same association, so weights, entropy, ratio bounds and the restore rule are bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 63, 64, 255, 256, 257, 1000, 4095, 4096])
@pytest.mark.parametrize("with_factors", [True, False])
def test_fused_equals_split(engine, n, with_factors):
    rng = np.random.default_rng(n)
    w0 = rng.uniform(0.2, 1.0, n).astype(np.float32)
    w0 /= w0.sum()
    lik = rng.uniform(0.0, 50.0, n).astype(np.float32)
    lik[rng.uniform(size=n) < 0.2] = 0.0  # dead particles: w = 0 is skipped by the entropy sum (pf.h:267-270)
    beam = rng.uniform(0.2, 1.0, n).astype(np.float32) if with_factors else None
    extra = rng.uniform(0.1, 0.4, n).astype(np.float32) if with_factors else None
    ratio = rng.uniform(0.0, 1.0, n).astype(np.float32) if with_factors else None
    out = {}
    try:
        for fused in (0, 1):
            engine.set_option("pf_fused", fused)
            out[fused] = engine.pf_measure(w0, lik, beam, extra, ratio)
    finally:
        engine.set_option("pf_fused", 1)
    a, b = out[0], out[1]
    np.testing.assert_array_equal(a["weights"], b["weights"])
    assert a["restored"] == b["restored"]
    if not a["restored"]:
        assert a["entropy"] == b["entropy"]
    assert a["match_ratio_min"] == b["match_ratio_min"] and a["match_ratio_max"] == b["match_ratio_max"]


def test_fused_restore_rule(engine):
    """sum <= 0: weights untouched, restored = 1 (pf.h:274-278)."""
    w0 = np.full(300, 1 / 300, np.float32)
    got = engine.pf_measure(w0, np.zeros(300, np.float32))
    assert got["restored"] is True
    np.testing.assert_array_equal(got["weights"], w0)


def test_larger_filters_still_take_the_split_path(engine):
    n = 5000
    rng = np.random.default_rng(1)
    w0 = np.full(n, 1 / n, np.float32)
    lik = rng.uniform(0, 10, n).astype(np.float32)
    got = engine.pf_measure(w0, lik)
    np.testing.assert_allclose(got["weights"].sum(dtype=np.float64), 1.0, rtol=1e-6)


@pytest.mark.parametrize("n_p,n_s,n_b", [(1025, 4352, 0), (4096, 4352, 0), (4100, 5000, 7), (4159, 4608, 0), (16387, 4352, 3)])
def test_the_two_launch_tail_behind_the_tiled_kernel_equals_the_launches_apart(engine, n_p, n_s, n_b):
    """Round 6: behind the tiled likelihood kernel (default mode, 4097 .. 28 146 points, more than 1024 particles) lik_finalize's
    sum and pf_partial's product are ONE launch (lik_pf_partial_kernel: wavefront partials) and pf_reduce runs inside pf_apply.
    Same arithmetic in the same association as the kernels apart — measure_batch (lik_finalize, beam_finalize), then pf_measure
    on its results (pf_partial, reduce inside apply); a one-device group runs the sharded protocol (lik_pf_partial, pf_reduce over
    the wavefront partials into the all-reduce's layout, the collective, pf_apply) — so every output is the same bits:
    likelihoods, ratios, beam scores, weights, entropy, ratio bounds."""
    from mcl_3dl_amd import capi
    from mcl_3dl_amd.synthetic import make_scene
    sc = make_scene(n=91, n_p=n_p, n_s=n_s, n_b=max(n_b, 1), seed=900 + n_b)
    beam = sc.scan_beam[:n_b] if n_b else None
    lab = sc.scan_beam_label[:n_b] if n_b else None
    engine.set_map(sc.map_xyz, sc.map_label, stamp=6600 + n_b, dist_weight=(1.0, 1.0, 5.0))
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=max(n_b, 1))
    extra = np.random.default_rng(n_p).uniform(0.1, 0.4, n_p).astype(np.float32)
    w0 = np.random.default_rng(n_s).uniform(0.0, 1.0, n_p).astype(np.float32)
    w0[::7] = 0.0
    one = engine.measure_update(sc.poses, w0, sc.scan_lik, beam, lab, sc.origins, extra=extra)
    assert engine.get_option("lik_exact") == 0     # (the fp64 tree of the tiled kernel: the path this test is about)
    lik, ratio, bscore = engine.measure_batch(sc.poses, sc.scan_lik, beam, lab, sc.origins)
    apart = engine.pf_measure(w0, lik, bscore, extra, ratio)
    np.testing.assert_array_equal(one["lik"], lik)
    np.testing.assert_array_equal(one["quality"], ratio)
    np.testing.assert_array_equal(one["beam"], bscore)
    np.testing.assert_array_equal(one["weights"], apart["weights"])
    assert one["entropy"] == apart["entropy"] and one["restored"] == apart["restored"] is False
    assert one["match_ratio_min"] == apart["match_ratio_min"] and one["match_ratio_max"] == apart["match_ratio_max"]
    g = capi.Group((0,))
    try:
        g.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 5.0))
        g.set_likelihood_params()
        g.set_beam_params(num_points=max(n_b, 1))
        g.set_option("direct_single", 0)   # the sharded protocol with one rank: partial / reduce / all-reduce / apply apart
        grp = g.measure_update(sc.poses, w0, sc.scan_lik, beam, lab, sc.origins, extra=extra)
    finally:
        g.close()
    for k in ("lik", "quality", "beam", "weights"):
        np.testing.assert_array_equal(one[k], grp[k], err_msg=k)
    assert one["entropy"] == grp["entropy"]
    assert one["match_ratio_min"] == grp["match_ratio_min"] and one["match_ratio_max"] == grp["match_ratio_max"]


def test_beam_counters_stay_consistent_across_the_paths_that_share_them(engine):
    """Round 6: with the update's tail kernel behind it the beam kernel prepares its per-(particle, origin) constants per work-group
    in LDS and finds its penalty counters zeroed by the PREVIOUS update's tail kernel (no beam_origin / fill launch). The counters
    are shared with every other path (measure_batch, other particle counts, pf_partial_kernel / pf_fused_kernel taking the counts
    behind the per-particle likelihood kernels): whatever the order of calls, every beam score equals the one of a plain
    measure_batch."""
    from mcl_3dl_amd.synthetic import make_scene
    sc = make_scene(n=91, n_p=6000, n_s=4500, n_b=600, seed=77)
    engine.set_map(sc.map_xyz, sc.map_label, stamp=6700, dist_weight=(1.0, 1.0, 1.0))
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=600)
    w0 = np.full(6000, np.float32(1.0 / 6000), np.float32)

    def batch(n_p, n_s, n_b):
        return engine.measure_batch(sc.poses[:n_p], sc.scan_lik[:n_s], sc.scan_beam[:n_b], sc.scan_beam_label[:n_b], sc.origins)

    def update(n_p, n_s, n_b):
        return engine.measure_update(sc.poses[:n_p], w0[:n_p], sc.scan_lik[:n_s], sc.scan_beam[:n_b], sc.scan_beam_label[:n_b],
                                     sc.origins)

    want = {}
    for shape in [(4100, 4500, 600), (6000, 4500, 600), (4100, 300, 600), (2000, 4500, 64), (4100, 4500, 64), (700, 300, 600),
                  (1000, 4500, 64), (300, 300, 64)]:
        want[shape] = batch(*shape)
    order = [(4100, 4500, 600), (4100, 4500, 600), (6000, 4500, 600), (4100, 4500, 600), (4100, 300, 600), (4100, 4500, 600),
             (2000, 4500, 64), (4100, 4500, 64), (4100, 4500, 600), (6000, 4500, 600),
             # 513 .. 1024 particles: pf_fused_kernel takes the counts; <= 512: the one-launch update counts for itself
             (700, 300, 600), (700, 300, 600), (4100, 4500, 600), (1000, 4500, 64), (300, 300, 64), (1000, 4500, 64), (6000, 4500, 600)]
    for k, shape in enumerate(order):
        got = update(*shape)
        np.testing.assert_array_equal(got["beam"], want[shape][2], err_msg="update %d %r" % (k, shape))
        np.testing.assert_array_equal(got["lik"], want[shape][0], err_msg="update %d %r" % (k, shape))
        if k % 3 == 1:   # a plain batch in between uses (and dirties) the same counters
            b = batch(*shape)
            np.testing.assert_array_equal(b[2], want[shape][2])
