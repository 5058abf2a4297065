"""GPU tests of the candidate-voxel index on IRREGULAR maps (the benchmark maps are lattices): the pruned per-voxel
candidate sets must give exactly the nearest distance the exhaustive 27-cell scan gives, on dense random clutter
(many candidates per voxel -> overflow records), jittered planes, exactly coincident points, isolated points, and with
a strongly anisotropic dist_weight. Ground truth a second time: the stand-alone radius search."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
IDENTITY = np.array([[0, 0, 0, 0, 0, 0, 1]], np.float32)


def maps(rng):
    clutter = rng.uniform(-2.0, 2.0, (60000, 3))                                  # ~940 pts/m^3: dozens per voxel
    plane = np.stack([rng.uniform(-3, 3, 40000), rng.uniform(-3, 3, 40000), rng.normal(0, 0.01, 40000)], 1)
    dup = np.repeat(rng.uniform(-1, 1, (500, 3)), 8, 0)                           # 8 exactly coincident copies each
    sparse = rng.uniform(-6, 6, (300, 3))                                         # isolated points
    mix = np.concatenate([clutter[:8000], plane[:8000], dup, sparse], 0)
    return {"clutter": clutter, "plane": plane, "duplicates": dup, "sparse": sparse, "mix": mix}


@pytest.mark.parametrize("name", ["clutter", "plane", "duplicates", "sparse", "mix"])
@pytest.mark.parametrize("dist_weight", [None, (1.0, 2.0, 5.0)])
def test_candidate_index_on_irregular_maps(engine, name, dist_weight):
    rng = np.random.default_rng(hash(name) % 1000)
    m = maps(rng)[name].astype(np.float32)
    # queries: near map points (most find a neighbour), plus uniformly random ones (many do not)
    near = m[rng.integers(0, len(m), 30000)] + rng.normal(0, 0.08, (30000, 3)).astype(np.float32)
    far = rng.uniform(m.min() - 1, m.max() + 1, (5000, 3)).astype(np.float32)
    queries = np.concatenate([near, far], 0).astype(np.float32)
    engine.set_map(m, None, stamp=200, dist_weight=dist_weight)
    engine.set_likelihood_params(match_dist_min=0.2, match_dist_flat=0.0, match_weight=1.0)
    # one "particle" (identity pose) per 5000-query chunk, each query its own 1-point-per-lane scan
    try:
        results = {}
        for mode, ratio, phase in ((0, 0.5, 0.5), (2, 0.5, 0.5), (2, 0.25, 0.0), (2, 1.0, 0.3)):
            engine.set_option("lik_index", mode)
            engine.set_option("cand_voxel_ratio", ratio)
            engine.set_option("cand_phase", phase)
            engine.set_option("lik_tiled", 0)
            out = []
            for c in range(0, len(queries), 5000):
                lik, rat, _ = engine.measure_batch(IDENTITY, queries[c:c + 5000])
                out.append((lik[0], rat[0]))
            results[(mode, ratio, phase)] = np.array(out)
            if mode:
                st = engine.index_stats()
                assert st["candidates"] > 0
    finally:
        engine.set_option("lik_index", 2)
        engine.set_option("cand_voxel_ratio", 0.0)
        engine.set_option("cand_phase", 0.5)
        engine.set_option("lik_tiled", 1)
        engine.set_likelihood_params()
    base = results[(0, 0.5, 0.5)]
    for key, val in results.items():
        np.testing.assert_array_equal(val, base, err_msg=str(key))
    assert base[:, 1].max() > 0.2  # the case exercises matches ...
    # ... and the sums ARE those of the stand-alone radius search: (r - d) over the found, added as the reference adds them —
    # float, sequentially, in the order of the queries (round 6: the per-particle kernel's caller-order rows, the default)
    idx, sq = engine.radius_search(queries, 0.2)
    want = 0.0
    for c in range(0, len(queries), 5000):
        f = idx[c:c + 5000] >= 0
        d = np.sqrt(sq[c:c + 5000][f]).astype(np.float32)
        terms = (np.float32(0.2) - d).astype(np.float32)
        assert base[c // 5000, 0] == np.cumsum(terms, dtype=np.float32)[-1] if len(terms) else base[c // 5000, 0] == 0.0
        assert base[c // 5000, 1] == np.float32(f.sum()) / np.float32(5000)
