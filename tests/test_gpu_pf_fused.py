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
