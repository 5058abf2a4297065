"""Seeded differential fuzzing of the HIP engine against the CPU oracle: every seed draws its own map shape, labels,
dist_weight, likelihood and beam parameters (grids, tolerances, label filter, penalty mode), scan sizes, origins and
poses (including un-normalised quaternions and particles far outside the map), then compares

  * likelihood score      bit-identical in the DEFAULT mode (round 6: every scan of at most 4096 points is added up in the
                          caller's order, as floats — strict_order 2) and with strict_order 1 / 3; the opt-out fp64 tree
                          (strict_order 0): rtol 1e-5 (north_star), 3e-5 on scans > 512 points (see below)
  * match ratio           exact
  * beam score            exact                                 * beam status + collided map point   exact
  * the fused update      weights bit-identical (default and strict_order 1) / rtol 3e-5 (strict_order 0), entropy rtol 1e-5,
                          match-ratio min/max and `restored` exact

The parity tests elsewhere use the reference's default parameters on cube maps; this file is the guard for every
other combination the plugin surface accepts (parameters.h:64-132).

dist_weight components stay >= 1: below 1 the reference's ChunkedKdtree can miss a neighbour that sits across a chunk
border (its overlap margin is max_search_radius in RAW coordinates, chunked_kdtree.h:139-201, while the search radius
is in weighted ones), the engine returns the global nearest neighbour; DESIGN.md §4 lists this as a declared deviation."""
import os

import numpy as np
import pytest

from oracle import pyoracle

pytestmark = pytest.mark.gpu
RTOL = 1e-5
# the default suite runs seeds 0..95 (~7 s); MCL3DL_FUZZ_FIRST / MCL3DL_FUZZ_LAST widen the sweep (seeds 96..599 were run
# once for round 1: 504 further configurations, all green)
FUZZ_FIRST = int(os.environ.get("MCL3DL_FUZZ_FIRST", "0"))
FUZZ_LAST = int(os.environ.get("MCL3DL_FUZZ_LAST", "96"))
# MCL3DL_FUZZ_OFFSET=x,y,z moves every scene (map and particles) that far from the origin: large world coordinates,
# where float32 positions are coarse (1e5 m: 8 mm steps) — parity must hold there too (seeds 0..95 run once with
# 1e5,-2e5,50: green)
FUZZ_OFFSET = np.array([float(v) for v in os.environ.get("MCL3DL_FUZZ_OFFSET", "0,0,0").split(",")], np.float64)


def draw_map(rng):
    kind = rng.integers(0, 4)
    ext = rng.uniform(2.0, 15.0)
    if kind == 0:  # lattice planes (floor + two walls), like the reference's test scenes
        s = rng.choice([0.05, 0.1, 0.2])
        g = np.arange(-ext / 2, ext / 2, s) + 0.5 * s
        a, b = np.meshgrid(g, g, indexing="ij")
        floor = np.stack([a.ravel(), b.ravel(), np.full(a.size, -0.9)], 1)
        wall1 = np.stack([np.full(a.size, ext / 2), a.ravel(), b.ravel() * 0.3], 1)
        wall2 = np.stack([a.ravel(), np.full(a.size, -ext / 2), b.ravel() * 0.3], 1)
        m = np.concatenate([floor, wall1, wall2], 0)
    elif kind == 1:  # uniform clutter
        m = rng.uniform(-ext / 2, ext / 2, (int(rng.integers(500, 20000)), 3))
    elif kind == 2:  # noisy planes
        n = int(rng.integers(2000, 20000))
        m = np.stack([rng.uniform(-ext / 2, ext / 2, n), rng.uniform(-ext / 2, ext / 2, n), rng.normal(0, 0.03, n)], 1)
        m = np.concatenate([m, m[:, [2, 0, 1]] + [ext / 3, 0, 0]], 0)
    else:  # clusters + isolated points + exact duplicates
        c = rng.uniform(-ext / 2, ext / 2, (30, 3))
        m = np.concatenate([c[rng.integers(0, 30, 6000)] + rng.normal(0, 0.15, (6000, 3)),
                            rng.uniform(-ext, ext, (200, 3)), np.repeat(rng.uniform(-1, 1, (50, 3)), 4, 0)], 0)
    if len(m) > 30000:
        m = m[rng.choice(len(m), 30000, replace=False)]
    label = rng.integers(0, 4, len(m)).astype(np.uint32) if rng.random() < 0.6 else np.zeros(len(m), np.uint32)
    return m.astype(np.float32), label, ext


def draw_case(seed):
    rng = np.random.default_rng(900 + seed)
    m, label, ext = draw_map(rng)
    dist_weight = [(1.0, 1.0, 1.0), (1.0, 1.0, 5.0), (2.0, 1.5, 3.0), None][rng.integers(0, 4)]
    r = float(rng.choice([0.05, 0.1, 0.2, 0.35, 0.6]))
    lik_kw = dict(match_dist_min=r, match_dist_flat=float(rng.choice([0.0, 0.25 * r, 0.9 * r, 1.5 * r])),
                  match_weight=float(rng.choice([1.0, 5.0, 0.3])))
    n_b = int(rng.choice([0, 1, 3, 64, 300]))
    beam_kw = dict(map_grid_x=float(rng.choice([0.05, 0.1, 0.3])), map_grid_y=float(rng.choice([0.05, 0.1, 0.3])),
                   map_grid_z=float(rng.choice([0.05, 0.1, 0.3])), dda_grid_size=float(rng.choice([0.1, 0.2, 0.5])),
                   ray_angle_half=float(rng.choice([0.0, 0.25, 1.0])) * np.pi / 180.0,
                   hit_range=float(rng.choice([0.1, 0.3, 0.8])), beam_likelihood_min=float(rng.choice([0.05, 0.2, 0.6])),
                   num_points=max(n_b, 1), ang_total_ref=float(rng.choice([np.pi / 6, 0.1, np.pi / 2])),
                   filter_label_max=int(rng.choice([0xFFFFFFFF, 0, 2])),
                   add_penalty_short_only_mode=bool(rng.integers(0, 2)))
    n_p = int(rng.choice([1, 5, 64, 300]))
    n_s = int(rng.choice([0, 1, 31, 33, 257, 1500, 3000])) if n_b else int(rng.choice([1, 31, 33, 257, 1500, 3000]))
    n_o = int(rng.integers(1, 4))
    # a "true" pose inside the map; scan = nearby map points seen from it + noise, plus some random points
    pos0 = rng.uniform(-ext / 4, ext / 4, 3)
    near = m[rng.integers(0, len(m), max(n_s, 1))] - pos0 + rng.normal(0, 0.03, (max(n_s, 1), 3))
    scan_lik = np.where(rng.random((max(n_s, 1), 1)) < 0.8, near, rng.uniform(-ext, ext, (max(n_s, 1), 3)))[:n_s]
    nearb = m[rng.integers(0, len(m), max(n_b, 1))] - pos0 + rng.normal(0, 0.05, (max(n_b, 1), 3))
    scan_beam = nearb[:n_b] * rng.choice([1.0, 0.7, 1.3], (n_b, 1))  # some end short of / beyond the surface
    scan_beam_label = rng.integers(0, n_o, n_b).astype(np.uint32)
    origins = rng.normal(0, 0.3, (n_o, 3))
    poses = np.zeros((n_p, 7), np.float32)
    poses[:, :3] = pos0 + rng.normal(0, 0.15, (n_p, 3))
    q = rng.normal(0, 0.05, (n_p, 4))
    q[:, 3] += 1.0
    if rng.random() < 0.5:
        q /= np.linalg.norm(q, axis=1, keepdims=True)  # otherwise left un-normalised (state_6dof.h:217 vs beam.cpp:145)
    poses[:, 3:] = q
    if n_p > 4:
        poses[-1, :3] += 10 * ext  # a particle far outside the map: likelihood 0, rays never enter the grid
        poses[-2, :3] = m.min(0) - 0.01  # and one just outside the map's bounding box
    if np.any(FUZZ_OFFSET != 0):
        m = (m.astype(np.float64) + FUZZ_OFFSET).astype(np.float32)
        poses[:, :3] = (poses[:, :3].astype(np.float64) + FUZZ_OFFSET).astype(np.float32)
    return dict(map=m, label=label, dist_weight=dist_weight, lik_kw=lik_kw, beam_kw=beam_kw,
                scan_lik=scan_lik.astype(np.float32), scan_beam=scan_beam.astype(np.float32),
                scan_beam_label=scan_beam_label, origins=origins.astype(np.float32), poses=poses, ext=ext, rng=rng)


def make_oracle(kind, c):
    # the node sizes the chunk overlap from the models: max(match_dist_min, 4 * max map grid)
    # (src/mcl_3dl.cpp:1320-1327, likelihood.h:61-64, beam.cpp:60-63)
    kw = c["beam_kw"]
    o = pyoracle.Oracle(kind, max_search_radius=max(c["lik_kw"]["match_dist_min"],
                                                    4.0 * max(kw["map_grid_x"], kw["map_grid_y"], kw["map_grid_z"])))
    o.set_map(c["map"], c["label"], dist_weight=c["dist_weight"])
    o.set_likelihood_params(pyoracle.LikelihoodParams(**c["lik_kw"]))
    o.set_beam_params(pyoracle.BeamParams(**c["beam_kw"]))
    return o


def setup_engine(eng, c, stamp):
    eng.set_map(c["map"], c["label"], stamp=stamp, dist_weight=c["dist_weight"])
    eng.set_likelihood_params(**c["lik_kw"])
    kw = dict(c["beam_kw"])
    kw["map_grid"] = (kw.pop("map_grid_x"), kw.pop("map_grid_y"), kw.pop("map_grid_z"))
    eng.set_beam_params(**kw)


@pytest.mark.parametrize("seed", range(FUZZ_FIRST, FUZZ_LAST))
def test_random_configuration(engine, oracle_kind, seed):
    c = draw_case(seed)
    try:
        setup_engine(engine, c, stamp=5000 + seed)
        o = make_oracle(oracle_kind, c)
        has_beam = len(c["scan_beam"]) > 0
        want_lik, want_q = o.likelihood_measure(c["poses"], c["scan_lik"])
        # strict_order: the reference's float summation order -> bit-identical scores
        engine.set_option("strict_order", 1)
        lik, ratio, beam = engine.measure_batch(c["poses"], c["scan_lik"], c["scan_beam"] if has_beam else None,
                                                c["scan_beam_label"] if has_beam else None, c["origins"])
        np.testing.assert_array_equal(lik, want_lik)
        np.testing.assert_array_equal(ratio, want_q)
        # strict_order = 3: the same recurrence inside the likelihood kernel, over the scan in the engine's order — bit-identical
        # to the reference on the scan permuted that way (round 5: no term array, no replay pass)
        if len(c["scan_lik"]):
            engine.set_option("strict_order", 3)
            lik3, ratio3, _ = engine.measure_batch(c["poses"], c["scan_lik"])
            order = engine.scan_order(len(c["scan_lik"]))
            want3, want_q3 = o.likelihood_measure(c["poses"], np.ascontiguousarray(c["scan_lik"][order]))
            np.testing.assert_array_equal(lik3, want3)
            np.testing.assert_array_equal(ratio3, want_q3)
            np.testing.assert_array_equal(want_q3, want_q)
        # the DEFAULT (strict_order 2): scans of at most strict_exact_max = 4096 points — all of this file's — are added up as
        # the reference adds them, by the per-particle kernels' LDS rows (float_chain.h) or the replay: the reference's bits
        engine.set_option("strict_order", 2)
        assert len(c["scan_lik"]) <= engine.get_option("strict_exact_max")
        lik, ratio, beam = engine.measure_batch(c["poses"], c["scan_lik"], c["scan_beam"] if has_beam else None,
                                                c["scan_beam_label"] if has_beam else None, c["origins"])
        np.testing.assert_array_equal(lik, want_lik)
        np.testing.assert_array_equal(ratio, want_q)
        # the opt-out (strict_order 0): the same float terms summed in fp64. What is left is the rounding of the REFERENCE's
        # own float running sum (bounded by n_s * 2^-24 relative, ~sqrt(n_s) * 2^-24 typical): inside north_star's 1e-5 for
        # the scan sizes of BASELINE.json's configs (tests/test_gpu_parity.py, test_gpu_fullsize.py), up to ~2e-5 on the
        # longest random scans here (match_dist_flat = 0.9 r makes most terms EQUAL: their roundings do not cancel)
        engine.set_option("strict_order", 0)
        lik, ratio, beam = engine.measure_batch(c["poses"], c["scan_lik"], c["scan_beam"] if has_beam else None,
                                                c["scan_beam_label"] if has_beam else None, c["origins"])
        np.testing.assert_allclose(lik, want_lik, rtol=RTOL if len(c["scan_lik"]) <= 512 else 3e-5, atol=0)
        np.testing.assert_array_equal(ratio, want_q)
        if has_beam:
            want_beam, _ = o.beam_measure(c["poses"], c["scan_beam"], c["scan_beam_label"], c["origins"])
            np.testing.assert_array_equal(beam, want_beam)
        # explicit rays through the same grid
        rng = c["rng"]
        lo, hi = c["map"].min(0), c["map"].max(0)
        begin = rng.uniform(lo - 0.3, hi + 0.3, (1500, 3)).astype(np.float32)
        end = (begin + rng.normal(0, 0.25 * c["ext"], (1500, 3))).astype(np.float32)
        st, hit = engine.beam_status(begin, end)
        want_st, want_hit = o.beam_status(begin, end)
        np.testing.assert_array_equal(st, want_st)
        np.testing.assert_array_equal(hit, want_hit)
        # the fused update with random prior weights
        w0 = rng.uniform(0.1, 1.0, len(c["poses"])).astype(np.float32)
        w0 /= w0.sum()
        want = o.measure_update(c["poses"], w0, c["scan_lik"], c["scan_beam"], c["scan_beam_label"], c["origins"])
        # the node's odometry-error factor with a zero error vector: NormalLikelihood(sigma = 1)(0) (nd.h:41-58)
        extra = np.full(len(w0), np.float32(1.0 / np.sqrt(2.0 * np.pi)), np.float32)
        for strict in (1, 2, 0):
            engine.set_option("strict_order", strict)
            got = engine.measure_update(c["poses"], w0, c["scan_lik"], c["scan_beam"] if has_beam else None,
                                        c["scan_beam_label"] if has_beam else None, c["origins"], extra=extra)
            assert got["restored"] == want["restored"]
            if strict:  # (the default too: at most 300 particles here — pf::measure's float sum runs inside the update)
                np.testing.assert_array_equal(got["weights"], want["weights"])
            else:
                np.testing.assert_allclose(got["weights"], want["weights"], rtol=3e-5, atol=0)
            if not want["restored"]:
                np.testing.assert_allclose(got["entropy"], want["entropy"], rtol=RTOL, atol=1e-6)
            assert got["match_ratio_min"] == want["match_ratio_min"]
            assert got["match_ratio_max"] == want["match_ratio_max"]
    finally:
        engine.set_option("strict_order", 2)
        engine.set_likelihood_params()
        engine.set_beam_params()
