"""Seeded resampling cases shared by tests/golden/make_golden.py, the CPU tests and the GPU tests."""
import numpy as np

SIGMA6 = np.array([0.1, 0.1, 0.05, 0.01, 0.01, 0.05], np.float32)
SEED = 12345
# (n, number of weight-0 particles): ties in the accumulated probability only arise with dead particles
CASES = [(5, 0), (64, 10), (1000, 300), (2048, 0), (2048, 900)]


def make_case(n, dead):
    rng = np.random.default_rng(1000 * n + dead)
    s = rng.normal(0, 1, (n, 13)).astype(np.float32)
    s[:, 3:7] /= np.linalg.norm(s[:, 3:7], axis=1, keepdims=True)
    w = (rng.uniform(0, 1, n) ** 3).astype(np.float32)
    if dead:
        w[rng.integers(0, n, dead)] = 0
    w = (w / w.sum(dtype=np.float64)).astype(np.float32)
    return s, w


def resize_targets(n):
    return [n // 3 + 1, 2 * n + 5]
