"""One-off sweep (run on the GPU box): mcl3dl_hip_resample_* against the live reference (oracle/_ref) over random sizes,
dead-particle fractions, weight shapes and seeds; resizeParticle against the reference too. Prints the first mismatch."""
import sys

import numpy as np
import torch  # noqa: F401  (initialises the HIP runtime before the engine library)

sys.path.insert(0, ".")
from mcl_3dl_amd import capi  # noqa: E402
from oracle import pyoracle  # noqa: E402

SIGMA6 = np.array([0.1, 0.1, 0.05, 0.01, 0.01, 0.05], np.float32)
ref = pyoracle.Oracle("ref")
eng = capi.Engine(0)
rng = np.random.default_rng(2024)
bad = 0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    n = int(rng.choice([1, 2, 3, 7, 64, 257, 1000, 4096, 20000]))
    s = rng.normal(0, 1, (n, 13)).astype(np.float32)
    s[:, 3:7] /= np.linalg.norm(s[:, 3:7], axis=1, keepdims=True)
    shape = rng.integers(0, 4)
    w = {0: rng.uniform(0, 1, n), 1: rng.uniform(0, 1, n) ** 6, 2: np.ones(n), 3: rng.exponential(1, n)}[int(shape)]
    w = w.astype(np.float32)
    dead = rng.random() < 0.5
    if dead and n > 1:
        w[rng.integers(0, n, max(1, n // int(rng.integers(2, 10))))] = 0
    if w.sum() == 0:
        w[0] = 1
    w = (w / w.sum(dtype=np.float64)).astype(np.float32)
    seed = int(rng.integers(1, 1 << 30))
    want, _ = ref.resample(s, w, seed, SIGMA6)
    pstep = eng.resample_begin(w)
    ip, _ = ref.resample_draws(seed, pstep, SIGMA6, 0)
    src, dup, nd = eng.resample_plan(0, ip)
    _, noise = ref.resample_draws(seed, pstep, SIGMA6, nd)
    got = eng.resample_apply(s, noise)
    ok = np.array_equal(got, want)
    n_out = int(rng.choice([1, max(1, n // 3), n + 5, 2 * n + 1]))
    want2, _ = ref.resize(s, w, n_out)
    eng.resample_begin(w, n_out)
    eng.resample_plan(1)
    got2 = eng.resample_apply(s)
    ok2 = np.array_equal(got2, want2)
    if not (ok and ok2):
        bad += 1
        print("MISMATCH case", case, "n", n, "shape", shape, "dead", dead, "seed", seed, "resample ok", ok, "resize ok", ok2,
              "n_out", n_out)
        if bad > 5:
            break
print("done, mismatches:", bad)
