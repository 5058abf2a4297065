"""The float-order replay in the CALLER's order (strict_order 1, and 2 from strict_auto_min points) on long scans, option
strict_chunk > 0: the scan is ordered in chunks of the caller's order (Morton order inside a chunk), chunk c + 1 is evaluated
while chunk c is replayed on a stream of its own, the term buffer holds two chunks instead of the whole scan, and the result is
the reference's float bit for bit — the same bits as the one-piece replay (strict_chunk = 0, the default: the chunked form
measured 9 % slower at C5, profiles/r05g_chunked_replay.txt; it is the low-memory form)."""
import numpy as np
import pytest

from mcl_3dl_amd.synthetic import make_scene
from oracle import pyoracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    return make_scene(n=91, n_p=300, n_s=40000, n_b=40, seed=15, lik_clip=(0.5, 10.0, -3.0, 3.0))


def oracle_for(kind, sc, dw):
    o = pyoracle.Oracle(kind)
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=dw)
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    o.set_beam_params(pyoracle.BeamParams(num_points=40))
    return o


@pytest.mark.parametrize("n_s,chunk", [(40000, 16384), (40000, 4096), (32768, 16384), (9000, 2048), (33000, 1024)])
def test_chunked_replay_is_the_reference_in_the_callers_order(engine, oracle_kind, scene, n_s, chunk):
    sc = scene
    dw = (1.0, 1.0, 5.0)
    scan = np.ascontiguousarray(sc.scan_lik[:n_s])
    engine.set_map(sc.map_xyz, sc.map_label, stamp=8800, dist_weight=dw)
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=40)
    try:
        engine.set_option("strict_order", 1)
        engine.set_option("strict_chunk", chunk)
        lik, ratio, beam = engine.measure_batch(sc.poses, scan, sc.scan_beam, sc.scan_beam_label, sc.origins)
        assert int(engine.get_option("scan_chunk_in_use")) == chunk
        order = engine.scan_order(n_s)
        engine.set_option("strict_chunk", 0)
        lik1, ratio1, _ = engine.measure_batch(sc.poses, scan)
        assert int(engine.get_option("scan_chunk_in_use")) == 0
    finally:
        engine.set_option("strict_order", 2)
        engine.set_option("strict_chunk", 0)
    # the engine's order: a permutation that keeps every chunk of the caller's order together
    assert np.array_equal(np.sort(order), np.arange(n_s, dtype=np.uint32))
    assert np.array_equal(order // chunk, np.arange(n_s) // chunk)
    o = oracle_for(oracle_kind, sc, dw)
    want_lik, want_q = o.likelihood_measure(sc.poses, scan)
    want_beam, _ = o.beam_measure(sc.poses, sc.scan_beam, sc.scan_beam_label, sc.origins)
    np.testing.assert_array_equal(lik, want_lik)
    np.testing.assert_array_equal(ratio, want_q)
    np.testing.assert_array_equal(beam, want_beam)
    np.testing.assert_array_equal(lik1, want_lik)
    np.testing.assert_array_equal(ratio1, want_q)


def test_the_automatic_mode_chunks_from_twice_the_chunk_size_and_other_modes_take_the_same_scan(engine, oracle_kind, scene):
    sc = scene
    dw = (1.0, 1.0, 1.0)
    engine.set_map(sc.map_xyz, sc.map_label, stamp=8810, dist_weight=dw)
    engine.set_likelihood_params()
    engine.set_option("strict_chunk", 16384)
    o = oracle_for(oracle_kind, sc, dw)
    want_lik, want_q = o.likelihood_measure(sc.poses[:64], sc.scan_lik)
    # default: strict_order 2 replays from strict_auto_min = 28 147 points, in chunks of 16 384
    lik, ratio, _ = engine.measure_batch(sc.poses[:64], sc.scan_lik)
    assert int(engine.get_option("scan_chunk_in_use")) == 16384
    np.testing.assert_array_equal(lik, want_lik)
    np.testing.assert_array_equal(ratio, want_q)
    # a progressive batch over the same (chunk-ordered) scan, and the host-buffer update
    got = engine.measure_batch_begin(sc.poses, sc.scan_lik, slice_particles=100)
    engine.measure_batch_end()
    np.testing.assert_array_equal(got[0][:64], want_lik)
    w0 = np.full(64, 1.0 / 64, np.float32)
    upd = engine.measure_update(sc.poses[:64], w0, sc.scan_lik)
    np.testing.assert_array_equal(upd["lik"], want_lik)
    # uploaded once, then evaluated under every mode: fp64 sums, the in-kernel chain (engine order), the one-piece modes
    import torch
    dev = torch.device("cuda", 0)
    engine.upload_scan(sc.scan_lik)
    d_pose = torch.from_numpy(np.ascontiguousarray(sc.poses[:64])).to(dev)
    d_l, d_q = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    torch.cuda.synchronize()
    try:
        for mode in (0, 3, 2, 1):
            engine.set_option("strict_order", mode)
            engine.measure_device(d_pose, 64, d_l, d_q, None)
            engine.synchronize()
            np.testing.assert_array_equal(d_q.cpu().numpy(), want_q)
            if mode == 3:
                order = engine.scan_order(len(sc.scan_lik))
                w3, _ = o.likelihood_measure(sc.poses[:64], np.ascontiguousarray(sc.scan_lik[order]))
                np.testing.assert_array_equal(d_l.cpu().numpy(), w3)
            elif mode == 0:
                np.testing.assert_allclose(d_l.cpu().numpy(), want_lik, rtol=2e-5)
            else:
                np.testing.assert_array_equal(d_l.cpu().numpy(), want_lik)
    finally:
        engine.set_option("strict_order", 2)
        engine.set_option("strict_chunk", 0)
