"""GPU tests of the "next" row: expectationBiased / max / maxBiased / covariance against the oracle
(the reference's pf.h + ParticleWeightedMeanQuat + State6DOF::covElement compiled unmodified when oracle/_ref is built).
Tolerances: the reference sums in float sequentially, the GPU in fp64 trees; rpy uses device atan2f/asinf (<= 2 ulp)."""
import numpy as np
import pytest

from mcl_3dl_amd.synthetic import make_scene
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def weights_for(n, seed, dead=0):
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.0, 1.0, n).astype(np.float32) ** 4
    if dead:
        w[rng.integers(0, n, dead)] = 0.0
    w /= w.sum(dtype=np.float64)
    return w.astype(np.float32), rng.uniform(1e-6, 1.0, n).astype(np.float32)


@pytest.mark.parametrize("n", [1, 64, 1000, 4096, 65537])
def test_expectation_and_max(engine, oracle_kind, n):
    sc = make_scene(n=41, n_p=n, n_s=4, seed=n)
    w, bias = weights_for(n, n, dead=n // 10)
    o = pyoracle.Oracle(oracle_kind)
    for b in (None, bias):
        mean, total, im, ib = engine.expectation(sc.poses, w, b)
        want, wim, wib = o.expectation(sc.poses, w, b)
        np.testing.assert_allclose(mean[:3], want[:3], rtol=2e-5, atol=2e-6)
        # Quat(front, up) (quat.h:61-80) forms each component as sqrt(max(0, 1 +- xv.x +- yv.y +- zv.z)) / 2 from float
        # vectors: near the identity that argument is a difference of ~1.0 floats, i.e. a multiple of 6e-8, so the small
        # components of the REFERENCE's own result are quantised to ~1.2e-4 (0, 1.2e-4, 1.7e-4, ...). A 1-ulp change of
        # the summed front/up vectors (float-sequential vs fp64 sums) moves them by one such step; the rotation itself
        # is resolved to ~2.4e-4 rad by that formula. Large components agree to float precision.
        big = np.abs(want[3:]) > 0.1
        np.testing.assert_allclose(mean[3:][big], want[3:][big], rtol=3e-5)  # float-sequential sums over up to 65 537 terms
        angle = 2.0 * np.arccos(min(1.0, abs(float(np.dot(mean[3:].astype(np.float64), want[3:].astype(np.float64))))))
        assert angle < 5e-4
        assert abs(np.linalg.norm(mean[3:]) - 1.0) < 1e-6
        assert im == wim and ib == wib
        assert abs(total - (w.astype(np.float64) * (1.0 if b is None else b.astype(np.float64))).sum()) < 1e-5


def test_max_returns_first_maximum(engine):
    poses = make_scene(n=41, n_p=300, n_s=4).poses
    w = np.full(300, 0.001, np.float32)
    w[[17, 130, 131, 299]] = 0.05          # four equal maxima: the reference keeps the first (strict <)
    bias = np.ones(300, np.float32)
    bias[17] = 0.5                          # maxBiased then moves to the next one
    _, _, im, ib = engine.expectation(poses, w, bias)
    assert im == 17 and ib == 130


@pytest.mark.parametrize("n", [64, 4096])
def test_covariance(engine, oracle_kind, n):
    sc = make_scene(n=41, n_p=n, n_s=4, seed=3 * n, sigma_rpy=(0.05, 0.05, 0.4))
    w, _ = weights_for(n, 7 * n)
    o = pyoracle.Oracle(oracle_kind)
    want_cov, want_mean = o.covariance(sc.poses, w)
    # centred on the reference's own expectation(1.0) so that only the covariance path is compared
    cov = engine.covariance(sc.poses, w, want_mean)
    scale = np.sqrt(np.outer(np.diag(want_cov), np.diag(want_cov)))
    np.testing.assert_allclose(cov, want_cov, rtol=2e-4, atol=0)
    assert np.max(np.abs(cov - want_cov) / scale) < 5e-5
    np.testing.assert_array_equal(cov, cov.T)
    # subset = what pf::covariance does for random_sample_ratio < 1 (indices drawn by the caller's RNG)
    rng = np.random.default_rng(1)
    sub = rng.permutation(n)[: n // 3].astype(np.uint32)
    cov_sub = engine.covariance(sc.poses, w, want_mean, subset=sub)
    ws = w[sub] / w[sub].sum(dtype=np.float64)
    want_sub, _ = o.covariance(sc.poses[sub], ws.astype(np.float32))
    # the oracle re-centres on the subset's own mean; compare the position block after shifting (parallel-axis theorem)
    m_sub = (sc.poses[sub, :3].astype(np.float64) * ws[:, None]).sum(0)
    shift = np.outer(m_sub - want_mean[:3], m_sub - want_mean[:3])
    np.testing.assert_allclose(cov_sub[:3, :3], want_sub[:3, :3] + shift, rtol=2e-3, atol=1e-7)


def test_yaw_wraparound(engine, oracle_kind):
    """covElement wraps rpy differences into [-pi, pi] (state_6dof.h:175-179): particles straddling yaw = +-pi."""
    from mcl_3dl_amd.synthetic import quat_from_rpy
    rng = np.random.default_rng(5)
    n = 500
    yaw = np.pi + rng.normal(0, 0.2, n)
    yaw = (yaw + np.pi) % (2 * np.pi) - np.pi
    poses = np.concatenate([rng.normal(0, 0.1, (n, 3)), quat_from_rpy(np.stack([np.zeros(n), np.zeros(n), yaw], 1))],
                           1).astype(np.float32)
    w = np.full(n, 1.0 / n, np.float32)
    o = pyoracle.Oracle(oracle_kind)
    want_cov, want_mean = o.covariance(poses, w)
    cov = engine.covariance(poses, w, want_mean)
    np.testing.assert_allclose(cov[5, 5], want_cov[5, 5], rtol=1e-4)
    assert 0.02 < cov[5, 5] < 0.08  # ~0.2^2, not ~pi^2
