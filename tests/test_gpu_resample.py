"""GPU tests of the resampling row: mcl3dl_hip_resample_begin / plan / apply against the committed outputs
of the real pf.h (tests/golden/resample.npz; bit-exact, including who leads the tie groups of weight-0 particles) and,
where oracle/_ref is present, against the live reference with other seeds."""
import os

import numpy as np
import pytest

import resample_cases as rc
from oracle import pyoracle

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resample.npz"))


def test_upstream_resample_kats(engine):
    """test/src/test_pf.cpp:210-289 (ResampleFirstAndLastParticle) and :190-208 (flat likelihood = identity)."""
    small = np.float32(1.0e-06)
    states = np.zeros((5, 13), np.float32)
    states[:, 0] = np.arange(5)
    states[:, 6] = 1.0
    for name, probs, expected in (("first", [small, 0.2, 0.2, 0.2, np.float32(0.4) - small], [1, 2, 3, 4, 4]),
                                  ("last", [0.2, 0.2, 0.2, np.float32(0.4) - small, small], [0, 1, 2, 3, 3])):
        engine.resample_begin(np.array(probs, np.float32))
        src, dup, nd = engine.resample_plan(0, float(GOLD["kat_initial_p_" + name]))
        noise = np.zeros((nd, 13), np.float32)
        noise[:, 6] = 1.0  # zero sigma: the noise state is the identity
        out = engine.resample_apply(states, noise)
        assert out[:, 0].tolist() == [float(v) for v in expected]
    pstep = engine.resample_begin(np.full(10, 0.1, np.float32))
    src, dup, nd = engine.resample_plan(0, pstep * 0.5)
    assert src.tolist() == list(range(10))


@pytest.mark.parametrize("n,dead", rc.CASES)
def test_against_reference_goldens(engine, n, dead):
    s, w = rc.make_case(n, dead)
    key = "n%d_d%d" % (n, dead)
    engine.resample_begin(w)
    src, dup, nd = engine.resample_plan(0, float(GOLD[key + "_initial_p"]))
    np.testing.assert_array_equal(src, GOLD[key + "_source"])
    np.testing.assert_array_equal(dup, GOLD[key + "_dup"])
    assert nd == len(GOLD[key + "_noise"])
    np.testing.assert_array_equal(engine.resample_apply(s, GOLD[key + "_noise"]), GOLD[key + "_states"])
    for n_out in rc.resize_targets(n):
        engine.resample_begin(w, n_out)
        s2, d2, nd2 = engine.resample_plan(1)
        assert nd2 == 0
        np.testing.assert_array_equal(engine.resample_apply(s), GOLD[key + "_resize%d" % n_out])


@pytest.mark.parametrize("n,dead", [(300, 120), (20000, 9000), (262144, 0)])
def test_against_live_reference(engine, n, dead):
    if not pyoracle.available("ref"):
        pytest.skip("oracle/_ref not on this box; the golden test above covers the reference")
    ref = pyoracle.Oracle("ref")
    s, w = rc.make_case(n, dead)
    seed = 777
    want, _ = ref.resample(s, w, seed, rc.SIGMA6)
    pstep = engine.resample_begin(w)
    ip, _ = ref.resample_draws(seed, pstep, rc.SIGMA6, 0)
    src, dup, nd = engine.resample_plan(0, ip)
    _, noise = ref.resample_draws(seed, pstep, rc.SIGMA6, nd)
    np.testing.assert_array_equal(engine.resample_apply(s, noise), want)


def test_error_paths(engine):
    from mcl_3dl_amd import capi
    e = capi.Engine(0)
    with pytest.raises(capi.EngineError, match="before resample_begin"):
        e._rs_n_out = 4
        e.resample_plan(0, 0.0)
    e.resample_begin(np.full(4, 0.25, np.float32))
    with pytest.raises(capi.EngineError, match="before resample_plan"):
        e.resample_apply(np.zeros((4, 13), np.float32))
    src, dup, nd = e.resample_plan(0, 0.0)
    assert nd > 0
    with pytest.raises(capi.EngineError, match="need noise"):
        e.resample_apply(np.zeros((4, 13), np.float32), None)
    e.close()


@pytest.mark.parametrize("n,dead", [(1000, 300), (2048, 900), (5, 0)])
def test_device_resident_and_sliced_apply(engine, n, dead):
    """Weights and states stay in device memory; the output is produced in two slices, as two GPUs would
    (mcl_3dl_amd/distributed.py:sharded_resample), and must equal the reference's single pass bit for bit."""
    import torch
    from mcl_3dl_amd.distributed import EngineResampleOps, shard_bounds, sharded_resample
    key = "n%d_d%d" % (n, dead)
    dev = torch.device("cuda:0")
    s, w = rc.make_case(n, dead)
    d_w, d_s = torch.from_numpy(w).to(dev), torch.from_numpy(s).to(dev)
    ops = EngineResampleOps(engine)
    # world size 1 through the distributed entry point
    new_s, new_w, (src, dup) = sharded_resample(ops, d_s, d_w, n, lambda pstep: float(GOLD[key + "_initial_p"]),
                                                lambda nd: GOLD[key + "_noise"][:nd])
    np.testing.assert_array_equal(new_s.cpu().numpy(), GOLD[key + "_states"])
    assert torch.all(new_w == np.float32(1.0 / n))
    # two output slices from the same plan
    parts = []
    for r in range(2):
        lo, hi = shard_bounds(n, 2, r)
        parts.append(ops.apply_slice(d_s, GOLD[key + "_noise"], lo, hi - lo).cpu().numpy())
    np.testing.assert_array_equal(np.concatenate(parts), GOLD[key + "_states"])
    from mcl_3dl_amd import capi
    with pytest.raises(capi.EngineError, match="outside"):
        ops.apply_slice(d_s, GOLD[key + "_noise"], n - 1, 5)


@pytest.mark.parametrize("n,zeros", [(1, 0), (300, 0), (4096, 0), (4097, 0), (50000, 0), (262144, 0), (5000, 40), (4096, 4096 - 3)])
def test_device_resident_begin_runs_the_references_prefix_recurrence(engine, n, zeros):
    """resample_begin_device: the float recurrence `accum += p.probability_` (pf.h:193-197) over weights that live on the device
    (4 bytes per particle come to the host, where the recurrence runs): pstep is the reference's float."""
    import torch
    rng = np.random.default_rng(n + zeros)
    w = rng.uniform(0.1, 1.0, n).astype(np.float32)
    if zeros:
        w[rng.choice(n, zeros, replace=False)] = 0.0
    w /= w.sum(dtype=np.float64).astype(np.float32)
    d_w = torch.from_numpy(w).cuda()
    pstep = engine.resample_begin_device(d_w, n)
    src, dup, nd = engine.resample_plan(0, 0.37 * pstep)
    assert len(src) == n
    acc = np.float32(0)
    for x in w:
        acc = np.float32(acc + x)
    assert pstep == np.float32(acc / np.float32(n))
