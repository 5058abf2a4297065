"""The one-launch sort of arrays with keys of at most 16 bits (mcl_3dl_amd/csrc/sort_kernels.h: rs_sort16_kernel; 2049 .. 32 768
elements — the scan ordering of a 16 384-point likelihood scan; option sort_one_launch, OFF by default: it costs what the
launches per 8-bit pass cost, profiles/r05p_sort16_one_launch.txt): every work-group holds the whole key distribution in LDS and
places its own 1024 elements. Held to: the stable order numpy gives, the order the passes give, and — through the paths that
use it — the same scan order and the same bits of every result."""
import numpy as np
import pytest

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [2049, 3071, 4096, 10000, 16384, 20001, 32767, 32768])
@pytest.mark.parametrize("end_bit,spread", [(16, 65536), (16, 40), (12, 65536), (5, 65536), (1, 65536), (16, 1)])
def test_pairs_come_out_in_the_stable_order(engine, n, end_bit, spread):
    rng = np.random.default_rng(n + 131 * end_bit + spread)
    keys = rng.integers(0, spread, n).astype(np.uint32)
    keys |= rng.integers(0, 2, n).astype(np.uint32) << 20           # a bit above end_bit must not order anything
    vals = rng.permutation(n).astype(np.uint32)
    masked = keys & np.uint32((1 << end_bit) - 1)
    order = np.argsort(masked, kind="stable")
    for one_launch in (1, 0):
        engine.set_option("sort_one_launch", one_launch)
        try:
            ok, ov = engine.sort_pairs(keys, vals, end_bit=end_bit)
            ok2, ov2 = engine.sort_pairs(keys, None, end_bit=end_bit)
        finally:
            engine.set_option("sort_one_launch", 0)
        np.testing.assert_array_equal(ok, keys[order])
        np.testing.assert_array_equal(ov, vals[order])
        np.testing.assert_array_equal(ov2, order.astype(np.uint32))


def test_wider_keys_and_larger_arrays_take_the_passes(engine):
    rng = np.random.default_rng(5)
    engine.set_option("sort_one_launch", 1)
    for n, end_bit in ((16384, 17), (16384, 32), (32769, 16), (70000, 16)):
        keys = rng.integers(0, 1 << min(end_bit, 31), n).astype(np.uint32)
        order = np.argsort(keys & np.uint32((1 << end_bit) - 1 if end_bit < 32 else 0xffffffff), kind="stable")
        try:
            ok, ov = engine.sort_pairs(keys, None, end_bit=end_bit)
        finally:
            engine.set_option("sort_one_launch", 0 if (n, end_bit) == (70000, 16) else 1)
        np.testing.assert_array_equal(ov, order.astype(np.uint32))


@pytest.mark.parametrize("n_s", [2049, 5000, 16384, 30000])
def test_scan_order_and_results_do_not_depend_on_the_sort(engine, n_s):
    """upload (device ordering from 4096 points), the host-buffer update and the host's scan_order_host agree on the order;
    likelihoods are the same bits whichever sort ordered the scan, in the fp64 mode and with the in-kernel float sums."""
    sc = make_scene(n=91, n_p=300, n_s=n_s, n_b=64, seed=900 + n_s)
    engine.set_map(sc.map_xyz, sc.map_label, stamp=8800 + n_s, dist_weight=(1.0, 1.0, 5.0))
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=64)
    w0 = np.full(len(sc.poses), 1.0 / len(sc.poses), np.float32)
    want_order = capi.scan_order_host(sc.scan_lik)
    got = {}
    saved = engine.get_option("strict_order")
    try:
        for one_launch in (1, 0):
            engine.set_option("sort_one_launch", one_launch)
            for mode in (0, 3):
                engine.set_option("strict_order", mode)
                r = engine.measure_update(sc.poses, w0, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
                np.testing.assert_array_equal(engine.scan_order(n_s), want_order)
                lik, ratio, beam = engine.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
                np.testing.assert_array_equal(engine.scan_order(n_s), want_order)
                np.testing.assert_array_equal(lik, r["lik"])
                got[(one_launch, mode)] = r
    finally:
        engine.set_option("sort_one_launch", 0)
        engine.set_option("strict_order", saved)
    for mode in (0, 3):
        for k in ("lik", "quality", "beam", "weights"):
            np.testing.assert_array_equal(got[(1, mode)][k], got[(0, mode)][k], err_msg="%s mode %d" % (k, mode))
