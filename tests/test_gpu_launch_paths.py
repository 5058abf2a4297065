"""Round 6, the launches around the kernels: above 1024 particles an update is (beam origins,) ONE launch of both models (the beam
kernel's work-groups interleaved with the tiled likelihood kernel's — lik_beam_kernel, particle groups of 4 / 8 / 16 — or with the
per-particle likelihood kernel's — lik_particle_beam_kernel), the first pf::measure kernel — which also adds up the tiled kernel's per-tile partials and turns the beam
model's penalty counts into scores — and pf_apply with the reduction inside. Every one of these is the kernels' own arithmetic in
the same association, so whatever route an update takes the results are the same bits: the merged launch against the kernels
behind each other (overlap_models = 0), against the sharded protocol of a one-rank device group, and — on a sample of particles —
against the reference itself (src/lidar_measurement_model_likelihood.cpp:105-139, src/lidar_measurement_model_beam.cpp:124-155)."""
import numpy as np
import pytest

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_scene
from oracle import pyoracle

pytestmark = pytest.mark.gpu
ND0 = np.float32(1.0 / np.sqrt(2.0 * np.pi))
DW = (1.0, 1.0, 5.0)

SHAPES = [(1500, 4200, 40),    # G = 4: 94 beam work-groups ride with the tiled kernel
          (2048, 8192, 256),   # G = 8
          (4096, 4608, 16),    # G = 16, exactly 256 beam work-groups
          (5000, 6000, 0),     # no beam points: the tail kernel fills the ones
          (1100, 5000, 700),   # more beam work-groups than tiled ones
          (3000, 4100, 5),     # 59 beam work-groups: below the merged launch's threshold, counters cleared by the tail
          (1025, 4097, 64),
          # the per-particle likelihood kernel (caller-order rows below 2048 particles; every scan below 768 points) with the beam
          # kernel's work-groups interleaved: lik_particle_beam_kernel
          (1500, 2048, 48), (4096, 512, 16), (1100, 300, 64), (1900, 4096, 9),
          (4096, 96, 3), (1300, 128, 40), (2500, 33, 7)]   # up to 128 points: 64-thread work-groups, 64 rays per beam work-group


@pytest.fixture(scope="module")
def scene():
    return make_scene(n=91, n_p=5000, n_s=8192, n_b=700, seed=4242)


def run(eng, sc, n_p, n_s, n_b, w0, extra):
    beam = sc.scan_beam[:n_b] if n_b else None
    lab = sc.scan_beam_label[:n_b] if n_b else None
    return eng.measure_update(sc.poses[:n_p], w0, sc.scan_lik[:n_s], beam, lab, sc.origins, extra=extra)


@pytest.mark.parametrize("n_p,n_s,n_b", SHAPES)
def test_every_route_of_a_large_update_gives_the_same_bits(engine, oracle_kind, scene, n_p, n_s, n_b):
    sc = scene
    engine.set_map(sc.map_xyz, sc.map_label, stamp=6800, dist_weight=DW)
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=max(n_b, 1))
    rng = np.random.default_rng(n_p + n_s)
    w0 = rng.uniform(0.1, 1.0, n_p).astype(np.float32)
    extra = np.full(n_p, ND0, np.float32)
    one = run(engine, sc, n_p, n_s, n_b, w0, extra)
    again = run(engine, sc, n_p, n_s, n_b, w0, extra)          # (steady state: the counters the first update left zeroed)
    try:
        engine.set_option("overlap_models", 0)
        apart = run(engine, sc, n_p, n_s, n_b, w0, extra)
    finally:
        engine.set_option("overlap_models", 1)
    g = capi.Group((0,))
    try:
        g.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=DW)
        g.set_likelihood_params()
        g.set_beam_params(num_points=max(n_b, 1))
        g.set_option("direct_single", 0)
        beam = sc.scan_beam[:n_b] if n_b else None
        lab = sc.scan_beam_label[:n_b] if n_b else None
        grp = g.measure_update(sc.poses[:n_p], w0, sc.scan_lik[:n_s], beam, lab, sc.origins, extra=extra)
    finally:
        g.close()
    for other, name in ((again, "second update"), (apart, "overlap_models = 0"), (grp, "one-rank group")):
        for k in ("lik", "quality", "beam", "weights"):
            np.testing.assert_array_equal(one[k], other[k], err_msg="%s: %s" % (name, k))
        assert one["entropy"] == other["entropy"], name
        assert one["match_ratio_min"] == other["match_ratio_min"] and one["match_ratio_max"] == other["match_ratio_max"], name
    # ... and the reference on a sample of the particles
    sel = np.unique(np.concatenate([np.arange(8), np.linspace(0, n_p - 1, 40).astype(int)]))
    o = pyoracle.Oracle(oracle_kind)
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=DW)
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    o.set_beam_params(pyoracle.BeamParams(num_points=max(n_b, 1)))
    want_lik, want_q = o.likelihood_measure(sc.poses[:n_p][sel], sc.scan_lik[:n_s])
    np.testing.assert_array_equal(one["quality"][sel], want_q)
    np.testing.assert_allclose(one["lik"][sel], want_lik, rtol=1e-5)   # (4097 .. 28 146 points: the fp64 tree)
    if n_b:
        want_b = o.beam_measure(sc.poses[:n_p][sel], sc.scan_beam[:n_b], sc.scan_beam_label[:n_b], sc.origins)[0]
        np.testing.assert_array_equal(one["beam"][sel], want_b)
    else:
        assert (one["beam"] == 1.0).all()
    # pf::measure on the engine's own factors: float product, the sum within float rounding of the fp64 tree
    wn = w0 * (((np.float32(1.0) * one["beam"]) * one["lik"]) * extra)
    np.testing.assert_allclose(one["weights"], wn / np.float32(wn.sum(dtype=np.float64)), rtol=3e-7)
