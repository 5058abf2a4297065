"""strict_order = 3: the reference's float recurrence `score_like += dist * match_weight`
(src/lidar_measurement_model_likelihood.cpp:124-134) runs INSIDE the tiled likelihood kernel, over the scan in the ENGINE's
order (mcl3dl_hip_scan_order) — no N_s x N_p term array, no replay pass. The contract these tests hold it to:

    engine likelihood (strict_order = 3)  ==  reference measure() on the cloud { scan[order[0]], scan[order[1]], ... }

bit for bit (assert_array_equal), at every scan size and particle count, for every index / kernel family, with overflow
records (map of centroids), through the device call, the host-buffer update and the progressive batch.
"""
import numpy as np
import pytest

from mcl_3dl_amd.synthetic import make_scene
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def make_oracle(kind, sc, dist_weight, beam_kw=None):
    o = pyoracle.Oracle(kind)
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=dist_weight)
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    o.set_beam_params(pyoracle.BeamParams(**(beam_kw or {})))
    return o


def setup_engine(eng, sc, dist_weight, stamp, beam_kw=None):
    eng.set_map(sc.map_xyz, sc.map_label, stamp=stamp, dist_weight=dist_weight)
    eng.set_likelihood_params()
    eng.set_beam_params(**(beam_kw or {}))


class chain_mode:
    def __init__(self, eng, **opts):
        self.eng, self.opts = eng, dict(opts, strict_order=3)
        self.saved = {}

    def __enter__(self):
        for k, v in self.opts.items():
            self.saved[k] = self.eng.get_option(k)
            self.eng.set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            self.eng.set_option(k, v)


def check_order(order, n_s):
    assert order.shape == (n_s,)
    assert np.array_equal(np.sort(order), np.arange(n_s, dtype=np.uint32)), "scan_order is not a permutation"


# (particles, points): one tile, a ragged last tile, fewer than eight tiles, partial last row of tiles, ragged particle groups
SHAPES = [(5, 1), (64, 255), (64, 1000), (33, 2049), (300, 2500), (1000, 4100), (257, 6000)]


@pytest.mark.parametrize("n_p,n_s", SHAPES)
@pytest.mark.parametrize("dist_weight", [(1.0, 1.0, 1.0), (1.0, 1.0, 5.0)])
def test_chain_is_the_reference_on_the_scan_in_engine_order(engine, oracle_kind, n_p, n_s, dist_weight):
    sc = make_scene(n=91, n_p=n_p, n_s=n_s, seed=100 + n_s)
    setup_engine(engine, sc, dist_weight, stamp=300 + n_s)
    with chain_mode(engine):
        lik, ratio, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        order = engine.scan_order(n_s)
    check_order(order, n_s)
    o = make_oracle(oracle_kind, sc, dist_weight)
    want_lik, want_q = o.likelihood_measure(sc.poses, sc.scan_lik[order])
    assert np.any(want_lik > 0)
    np.testing.assert_array_equal(lik, want_lik)
    np.testing.assert_array_equal(ratio, want_q)
    # and the default mode (fp64 sums, caller's order) stays within its tolerance of the same points
    lik0, ratio0, _ = engine.measure_batch(sc.poses, sc.scan_lik)
    np.testing.assert_allclose(lik0, want_lik, rtol=3e-5)
    np.testing.assert_array_equal(ratio0, want_q)


@pytest.mark.parametrize("opts", [dict(lik_group=4), dict(lik_group=8), dict(lik_group=16), dict(lik_coop=0),
                                  dict(lik_index=0), dict(lik_defer=0), dict(cand_packed=0)])
def test_every_kernel_family_chains_the_same_bits(engine, oracle_kind, opts):
    sc = make_scene(n=91, n_p=150, n_s=3000, seed=7)
    dw = (1.0, 1.0, 5.0)
    setup_engine(engine, sc, dw, stamp=340)
    with chain_mode(engine, **opts):
        lik, ratio, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        order = engine.scan_order(len(sc.scan_lik))
    o = make_oracle(oracle_kind, sc, dw)
    want_lik, want_q = o.likelihood_measure(sc.poses, sc.scan_lik[order])
    np.testing.assert_array_equal(lik, want_lik)
    np.testing.assert_array_equal(ratio, want_q)


def test_chain_on_a_map_of_centroids(engine, oracle_kind):
    """Overflow records (deferred rounds parked in the term slots) in front of the chain."""
    sc = make_scene(n=91, n_p=200, n_s=4096, seed=11, map_jitter=0.045)
    dw = (1.0, 1.0, 1.0)
    setup_engine(engine, sc, dw, stamp=350)
    with chain_mode(engine):
        lik, ratio, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        order = engine.scan_order(len(sc.scan_lik))
    o = make_oracle(oracle_kind, sc, dw)
    want_lik, want_q = o.likelihood_measure(sc.poses, sc.scan_lik[order])
    np.testing.assert_array_equal(lik, want_lik)
    np.testing.assert_array_equal(ratio, want_q)


def test_chain_through_the_host_buffer_update_and_the_progressive_batch(engine, oracle_kind):
    sc = make_scene(n=91, n_p=1500, n_s=2600, n_b=40, seed=5)
    dw = (1.0, 1.0, 5.0)
    kw = dict(num_points=40)
    setup_engine(engine, sc, dw, stamp=360, beam_kw=kw)
    rng = np.random.default_rng(9)
    w0 = rng.uniform(0.5, 1.5, len(sc.poses)).astype(np.float32)
    w0 /= w0.sum()
    with chain_mode(engine):
        got = engine.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        order = engine.scan_order(len(sc.scan_lik))
        lik_b, ratio_b, beam_b = engine.measure_batch_begin(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label,
                                                            sc.origins, slice_particles=400)
        assert engine.measure_batch_wait(0) >= 400
        engine.measure_batch_end()
        order_b = engine.scan_order(len(sc.scan_lik))
        # twice in a row: the hand-off words of the first launch must not satisfy the second
        lik_c, _, _ = engine.measure_batch(sc.poses[::-1].copy(), sc.scan_lik)
    np.testing.assert_array_equal(order, order_b)
    o = make_oracle(oracle_kind, sc, dw, beam_kw=kw)
    want = o.measure_update(sc.poses, w0, sc.scan_lik[order], sc.scan_beam, sc.scan_beam_label, sc.origins)
    np.testing.assert_array_equal(got["lik"], want["lik"])
    np.testing.assert_array_equal(got["quality"], want["quality"])
    np.testing.assert_array_equal(got["beam"], want["beam"])
    np.testing.assert_allclose(got["weights"], want["weights"], rtol=1e-5)  # (weights: fp64 tree here, float recurrence there)
    np.testing.assert_array_equal(lik_b, want["lik"])
    np.testing.assert_array_equal(ratio_b, want["quality"])
    np.testing.assert_array_equal(beam_b, want["beam"])
    np.testing.assert_array_equal(lik_c[::-1], want["lik"])


def test_chain_at_the_headline_size(engine, oracle_kind):
    """C2: 4096 particles x 16 384 points (64 tiles, 256 particle groups: the size at which a row of tiles lasts about as
    long as its eight hand-offs). Every particle evaluated on the GPU, a sample of them by the reference."""
    sc = make_scene(n=408, n_p=4096, n_s=16384, seed=12345)
    dw = (1.0, 1.0, 1.0)
    setup_engine(engine, sc, dw, stamp=370)
    with chain_mode(engine):
        lik, ratio, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        lik2, ratio2, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        order = engine.scan_order(len(sc.scan_lik))
    np.testing.assert_array_equal(lik, lik2)
    sample = np.r_[0:8, 2040:2056, 4088:4096]
    o = make_oracle(oracle_kind, sc, dw)
    want_lik, want_q = o.likelihood_measure(sc.poses[sample], sc.scan_lik[order])
    np.testing.assert_array_equal(lik[sample], want_lik)
    np.testing.assert_array_equal(ratio[sample], want_q)
    lik0, _, _ = engine.measure_batch(sc.poses, sc.scan_lik)
    np.testing.assert_allclose(lik0, lik, rtol=2e-5)


def test_scan_order_errors(engine):
    sc = make_scene(n=91, n_p=8, n_s=100, seed=3)
    setup_engine(engine, sc, (1.0, 1.0, 1.0), stamp=380)
    engine.measure_batch(sc.poses, sc.scan_lik)
    with pytest.raises(RuntimeError):
        engine.scan_order(99)
    check_order(engine.scan_order(100), 100)


@pytest.mark.parametrize("n_p,n_s", [(40, 300), (64, 1000), (300, 5000), (700, 16384)])
def test_presorted_scans_are_installed_as_they_are(engine, oracle_kind, n_p, n_s):
    """Option scan_presorted: a caller that holds its scan in the engine's order (mcl3dl_hip_scan_order_host) skips the ordering
    launches; strict_order = 3 then sums in the CALLER's order — bit-identical to the reference on the caller's own array. A
    scan that is not in that order is evaluated in the order it has, at every size (same bits as the reference again)."""
    from mcl_3dl_amd import capi
    sc = make_scene(n=91, n_p=n_p, n_s=n_s, seed=200 + n_s)
    dw = (1.0, 1.0, 5.0)
    setup_engine(engine, sc, dw, stamp=390 + n_s)
    o = make_oracle(oracle_kind, sc, dw)
    held = np.ascontiguousarray(sc.scan_lik[capi.scan_order_host(sc.scan_lik)])
    with chain_mode(engine, scan_presorted=1):
        lik_held, q_held, _ = engine.measure_batch(sc.poses, held)
        np.testing.assert_array_equal(engine.scan_order(n_s), np.arange(n_s, dtype=np.uint32))
        lik_raw, q_raw, _ = engine.measure_batch(sc.poses, sc.scan_lik)          # NOT ordered: summed as it is
        np.testing.assert_array_equal(engine.scan_order(n_s), np.arange(n_s, dtype=np.uint32))
        w0 = np.full(n_p, 1.0 / n_p, np.float32)
        upd = engine.measure_update(sc.poses, w0, held)
    want_held, want_q = o.likelihood_measure(sc.poses, held)
    want_raw, _ = o.likelihood_measure(sc.poses, sc.scan_lik)
    np.testing.assert_array_equal(lik_held, want_held)
    np.testing.assert_array_equal(q_held, want_q)
    np.testing.assert_array_equal(lik_raw, want_raw)
    np.testing.assert_array_equal(q_raw, want_q)
    np.testing.assert_array_equal(upd["lik"], want_held)
    # and without the option the engine orders the held scan itself — to the same order
    with chain_mode(engine):
        lik2, _, _ = engine.measure_batch(sc.poses, held)
        np.testing.assert_array_equal(engine.scan_order(n_s), np.arange(n_s, dtype=np.uint32))
    np.testing.assert_array_equal(lik2, want_held)


@pytest.mark.parametrize("n_p,n_s,jitter", [(5, 1, 0.0), (64, 255, 0.0), (64, 1000, 0.0), (33, 2049, 0.0), (300, 2500, 0.0),
                                            (1000, 4100, 0.0), (257, 6000, 0.0), (1024, 16384, 0.0), (3000, 9000, 0.0),
                                            (200, 4096, 0.045), (777, 12345, 0.045)])
def test_four_tiles_per_work_group_chain_the_same_bits(engine, oracle_kind, n_p, n_s, jitter):
    """likelihood_chain_multi.h (option chain_ppl: 0 = by size — four tiles per work-group up to 1536 particles from 16 tiles
    on —, 1 = never, 4 = always): a quarter of the hand-offs, the same adds in the same order. Against the one-tile form, the
    reference on the scan in engine order (a sample of the particles at the larger sizes), twice in a row (tags), and with
    the ragged ends: one tile, a last super-tile of one to three tiles, a last row of fewer than eight super-tiles."""
    sc = make_scene(n=91, n_p=n_p, n_s=n_s, seed=4000 + n_s, **({"map_jitter": jitter} if jitter else {}))
    dw = (1.0, 1.0, 5.0) if n_s % 2 else (1.0, 1.0, 1.0)
    setup_engine(engine, sc, dw, stamp=420 + n_s)
    with chain_mode(engine, chain_ppl=1):
        lik1, q1, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        order = engine.scan_order(n_s)
    with chain_mode(engine, chain_ppl=4):
        lik4, q4, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        lik4b, _, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        np.testing.assert_array_equal(engine.scan_order(n_s), order)
        w0 = np.full(n_p, 1.0 / n_p, np.float32)
        upd = engine.measure_update(sc.poses, w0, sc.scan_lik)
    with chain_mode(engine, chain_ppl=0):
        lik0, q0, _ = engine.measure_batch(sc.poses, sc.scan_lik)
    for got_lik, got_q in ((lik4, q4), (lik4b, q4), (lik0, q0), (upd["lik"], upd["quality"])):
        np.testing.assert_array_equal(got_lik, lik1)
        np.testing.assert_array_equal(got_q, q1)
    sample = np.arange(n_p) if n_p * n_s <= 2_000_000 else np.unique(np.r_[0:6, n_p // 2 - 3:n_p // 2 + 3, n_p - 6:n_p])
    o = make_oracle(oracle_kind, sc, dw)
    want_lik, want_q = o.likelihood_measure(sc.poses[sample], sc.scan_lik[order])
    np.testing.assert_array_equal(lik4[sample], want_lik)
    np.testing.assert_array_equal(q4[sample], want_q)
