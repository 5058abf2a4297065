"""Soak test of the one-launch update (mcl_3dl_amd/csrc/update_kernels.h): its stages are handed over between work-groups by
arrival tickets, and the failure mode of a wrong hand-off is a RARE stale read — a handful of launches prove nothing. Here:
>= 20 000 back-to-back launches over random shapes (N_p <= 512, N_s <= 2048, N_b <= 256: the reference's operating range,
parameters.h:68,98) while a second stream keeps HBM and the L2s busy with large copies, every result compared bit for bit with
what the separate kernels (likelihood / beam / pf_partial / pf_reduce / pf_apply, no hand-off inside a launch) give for the same
input; then the same shapes with the memory-model-conformant tickets (option update_small_conformant)."""
import numpy as np
import pytest
import torch

from mcl_3dl_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    return make_scene(n=91, n_p=512, n_s=2048, n_b=256, seed=31337)


def _hammer(side, bufs):
    """Keep ~40 ms of 512 MB device copies queued on the side stream."""
    with torch.cuda.stream(side):
        for _ in range(24):
            bufs[1].copy_(bufs[0], non_blocking=True)
            bufs[0].copy_(bufs[1], non_blocking=True)


def _soak(engine, sc, n_shapes, reps, conformant, seed):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    bufs = [torch.empty(128 << 20, dtype=torch.float32, device=dev) for _ in range(2)]
    bufs[0].zero_()
    engine.set_map(sc.map_xyz, sc.map_label, stamp=6600 + conformant, dist_weight=(1.0, 1.0, 5.0))
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=256)
    launches = 0
    bad = []
    try:
        engine.set_option("update_small_conformant", conformant)
        for shape in range(n_shapes):
            n_p = int(rng.integers(1, 513))
            n_s = int(rng.choice([0, 1, 3, 96, 128, 129, 300, 700, 2048, int(rng.integers(1, 700))]))
            n_b = int(rng.choice([0, 3, 3, 48, 256, int(rng.integers(1, 257))]))
            if n_s == 0 and n_b == 0:
                n_b = 3
            first = int(rng.integers(0, 512 - n_p + 1))
            poses = np.ascontiguousarray(sc.poses[first:first + n_p])
            w0 = rng.uniform(0.05, 1.0, n_p).astype(np.float32)
            extra = rng.uniform(0.1, 0.4, n_p).astype(np.float32) if shape % 3 == 0 else None
            lik_pts = np.ascontiguousarray(sc.scan_lik[rng.permutation(2048)[:n_s]])
            beam_pts = np.ascontiguousarray(sc.scan_beam[rng.permutation(256)[:n_b]]) if n_b else None
            beam_lab = np.zeros(n_b, np.uint32) if n_b else None
            engine.set_option("update_small", 0)
            want = engine.measure_update(poses, w0, lik_pts, beam_pts, beam_lab, sc.origins, extra=extra)
            engine.set_option("update_small", 1)
            out_lik, out_ratio, out_beam = (np.zeros(n_p, np.float32) for _ in range(3))
            w = np.empty(n_p, np.float32)
            for rep in range(reps):
                if rep % 16 == 0 and side.query():
                    _hammer(side, bufs)
                w[:] = w0
                ent, rmin, rmax, restored = engine.measure_update_into(poses, w, lik_pts, beam_pts, beam_lab, sc.origins,
                                                                       out_lik, out_ratio, out_beam, extra=extra)
                launches += 1
                ok = (np.array_equal(w, want["weights"]) and np.array_equal(out_lik, want["lik"]) and
                      np.array_equal(out_ratio, want["quality"]) and np.array_equal(out_beam, want["beam"]) and
                      restored == want["restored"] and (restored or ent == want["entropy"]) and
                      rmin == want["match_ratio_min"] and rmax == want["match_ratio_max"])
                if not ok:
                    bad.append((shape, rep, n_p, n_s, n_b))
                    break
    finally:
        engine.set_option("update_small", 1)
        engine.set_option("update_small_conformant", 0)
        torch.cuda.synchronize()
    assert not bad, "one-launch update differs from the separate kernels at (shape, repetition, n_p, n_s, n_b) = %s" % bad[:5]
    return launches


def test_twenty_thousand_one_launch_updates_under_memory_pressure(engine, scene):
    n = _soak(engine, scene, n_shapes=250, reps=84, conformant=0, seed=1)
    assert n >= 20000
    print("soak: %d one-launch updates, all bit-identical to the separate kernels" % n)


def test_conformant_tickets_give_the_same_results(engine, scene):
    n = _soak(engine, scene, n_shapes=60, reps=40, conformant=1, seed=2)
    assert n >= 2000
