"""CPU tests of the resampling row: the plain-C restatement of pf::resample / resizeParticle — including
its restatement of libstdc++'s std::sort, which decides who leads a tie group of weight-0 particles — against the real
pf.h (oracle/_ref when built) and against the committed outputs of the real pf.h (tests/golden/resample.npz)."""
import os

import numpy as np
import pytest

import resample_cases as rc
from oracle import pyoracle

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resample.npz"))


def test_upstream_resample_kats():
    """test/src/test_pf.cpp:210-289: ResampleFirstAndLastParticle — engine seed 12345, zero sigma; initial_p is the
    draw that engine makes (recorded from the real engine in the golden file)."""
    port = pyoracle.Oracle("port")
    small = np.float32(1.0e-06)
    cases = [("first", [small, 0.2, 0.2, 0.2, np.float32(0.4) - small], [1.0, 2.0, 3.0, 4.0, 4.0]),
             ("last", [0.2, 0.2, 0.2, np.float32(0.4) - small, small], [0.0, 1.0, 2.0, 3.0, 3.0])]
    for name, probs, expected in cases:
        w = np.array(probs, np.float32)
        src, dup = port.resample_plan(w, 5, 0, float(GOLD["kat_initial_p_" + name]))
        assert [float(v) for v in src] == expected  # states are 0..4, so the source index IS the expected state


def test_flat_likelihood_resample_is_identity():
    """test/src/test_pf.cpp:190-208: uniform weights -> every particle keeps its state (zero sigma: noise adds 0)."""
    port = pyoracle.Oracle("port")
    w = np.full(10, 0.1, np.float32)
    pstep = port.resample_pstep(w, 10)
    src, dup = port.resample_plan(w, 10, 0, np.float32(pstep * 0.5))
    assert src.tolist() == list(range(10))


@pytest.mark.parametrize("n,dead", rc.CASES)
def test_port_reproduces_reference_goldens(n, dead):
    port = pyoracle.Oracle("port")
    s, w = rc.make_case(n, dead)
    key = "n%d_d%d" % (n, dead)
    src, dup = port.resample_plan(w, n, 0, float(GOLD[key + "_initial_p"]))
    np.testing.assert_array_equal(src, GOLD[key + "_source"])
    np.testing.assert_array_equal(dup, GOLD[key + "_dup"])
    np.testing.assert_array_equal(port.resample_apply(s, src, dup, GOLD[key + "_noise"]), GOLD[key + "_states"])
    for n_out in rc.resize_targets(n):
        s2, d2 = port.resample_plan(w, n_out, 1, 0.0)
        assert not d2.any()
        np.testing.assert_array_equal(port.resample_apply(s, s2, d2, np.zeros((0, 13))), GOLD[key + "_resize%d" % n_out])
    if dead:
        assert (w[src] == 0).sum() > 0  # the reference really does resample dead particles out of tie groups


@pytest.mark.skipif(not pyoracle.available("ref"), reason="oracle/_ref not built here")
@pytest.mark.parametrize("n,dead", [(300, 0), (300, 120), (5000, 2500)])
def test_port_equals_live_reference(n, dead):
    ref, port = pyoracle.Oracle("ref"), pyoracle.Oracle("port")
    s, w = rc.make_case(n, dead)
    for seed in (1, 12345):
        want, wout = ref.resample(s, w, seed, rc.SIGMA6)
        pstep = port.resample_pstep(w, n)
        ip, _ = ref.resample_draws(seed, pstep, rc.SIGMA6, 0)
        src, dup = port.resample_plan(w, n, 0, ip)
        _, noise = ref.resample_draws(seed, pstep, rc.SIGMA6, int(dup.sum()))
        np.testing.assert_array_equal(port.resample_apply(s, src, dup, noise), want)
        assert np.all(wout == np.float32(1.0 / n))
