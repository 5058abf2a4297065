#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref, built from /root/reference by oracle/Makefile).

Run in the build container (where /root/reference exists):   python tests/golden/make_golden.py
The GPU box has no /root/reference; the fixtures let the tests there (and the C port here) be checked against outputs
of the reference itself.  Inputs are regenerated from seeds by mcl_3dl_amd.synthetic; only outputs (+ the seeds and
parameters) are stored, so the files stay small.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import kats  # noqa: E402
from mcl_3dl_amd.synthetic import make_scene  # noqa: E402
from oracle import pyoracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

SCENES = {
    # name: (scene kwargs, dist_weight, beam kwargs)
    "c1_unit": (dict(n=91, n_p=64, n_s=1000, n_b=96, seed=12345), (1.0, 1.0, 1.0), dict(num_points=96)),
    "c1_default_weight": (dict(n=91, n_p=64, n_s=1000, n_b=96, seed=12345), (1.0, 1.0, 5.0), dict(num_points=96)),
    "label_wall": (dict(n=61, n_p=32, n_s=300, n_b=64, seed=7, label_wall=2), (1.0, 1.0, 2.0),
                   dict(num_points=64, filter_label_max=1, add_penalty_short_only_mode=False)),
}


def rays_for(sc, seed, n=3000):
    rng = np.random.default_rng(seed)
    half = sc.meta["n"] * sc.meta["spacing"] / 2
    begin = rng.uniform(-half * 1.05, half * 1.05, (n, 3)).astype(np.float32)
    end = (begin + rng.normal(0, 2.0, (n, 3))).astype(np.float32)
    return begin, end


def main():
    assert pyoracle.available("ref"), "oracle/_ref is not built (needs /root/reference): make -C oracle ref"
    for name, (skw, dw, bkw) in SCENES.items():
        sc = make_scene(**skw)
        o = pyoracle.Oracle("ref")
        o.set_map(sc.map_xyz, sc.map_label, dist_weight=dw)
        o.set_likelihood_params(pyoracle.LikelihoodParams())
        o.set_beam_params(pyoracle.BeamParams(**bkw))
        lik, quality = o.likelihood_measure(sc.poses, sc.scan_lik)
        beam, _ = o.beam_measure(sc.poses, sc.scan_beam, sc.scan_beam_label, sc.origins)
        begin, end = rays_for(sc, 99)
        status, hit = o.beam_status(begin, end)
        upd = o.measure_update(sc.poses, sc.weights, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins,
                               sc.odom_err, 0.5)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            scene_kwargs=np.array(repr(skw)), dist_weight=np.array(dw, np.float32), beam_kwargs=np.array(repr(bkw)),
            input_checksum=np.array([sc.map_xyz.sum(dtype=np.float64), sc.scan_lik.sum(dtype=np.float64),
                                     sc.poses.sum(dtype=np.float64), sc.scan_beam.sum(dtype=np.float64)]),
            lik=lik, quality=quality, beam=beam, status=status.astype(np.int8), hit=hit,
            upd_weights=upd["weights"], upd_entropy=np.float32(upd["entropy"]),
            upd_ratio=np.array([upd["match_ratio_min"], upd["match_ratio_max"]], np.float32), odom_sigma=np.float32(0.5))
        print(name, "lik[0:3]", lik[:3], "beam[0:3]", beam[:3], "entropy", upd["entropy"])
    # upstream fixture sweep (test/src/test_beam_likelihood.cpp:81-210): 2 modes x 6 hit ranges with the DDA raycaster
    raw_pc, pc_map = kats.beam_wall_fixture()
    out = {}
    for mode in (0, 1):
        for k, hr in enumerate((0.0, 0.2, 0.4, 0.6, 0.8, 1.0)):
            o = pyoracle.Oracle("ref", 10.0, 1.0)
            o.set_map(pc_map, None, dist_weight=None)
            o.set_beam_params(pyoracle.BeamParams(num_points=len(raw_pc), hit_range=hr, dda_grid_size=0.1,
                                                  add_penalty_short_only_mode=bool(mode), clip_z_min=-0.3,
                                                  clip_z_max=4.1))
            pc = raw_pc[(raw_pc[:, 2] > -0.3) & (raw_pc[:, 2] < 4.1)]
            xs = (0.1 * np.arange(-50, 50)).astype(np.float32)
            poses = np.zeros((len(xs), 7), np.float32)
            poses[:, 0] = xs
            poses[:, 6] = 1.0
            liks = np.zeros(len(xs), np.float32)
            for i in range(len(xs)):  # origins = {pos} per pose, as the upstream loop builds them
                liks[i] = o.beam_measure(poses[i:i + 1], pc, np.zeros(len(pc), np.uint32), poses[i:i + 1, :3])[0][0]
            end = np.stack([xs, np.zeros_like(xs), np.zeros_like(xs)], 1)
            st, _ = o.beam_status(np.zeros_like(end), end)
            out["lik_m%d_h%d" % (mode, k)] = liks
            out["status_m%d_h%d" % (mode, k)] = st.astype(np.int8)
    np.savez_compressed(os.path.join(HERE, "beam_wall_fixture.npz"), **out)
    print("beam_wall_fixture: ", "".join("s*lt"[s] for s in out["status_m1_h2"]))


def resample_goldens():
    """pf::resample (seed 12345) and pf::resizeParticle through the real pf.h: expected states + the random numbers the
    resample consumed (so that a path that keeps the caller's RNG outside can be replayed exactly)."""
    import resample_cases as rc
    ref = pyoracle.Oracle("ref")
    port = pyoracle.Oracle("port")
    out = {}
    for n, dead in rc.CASES:
        s, w = rc.make_case(n, dead)
        want, _ = ref.resample(s, w, rc.SEED, rc.SIGMA6)
        pstep = port.resample_pstep(w, n)
        ip, _ = ref.resample_draws(rc.SEED, pstep, rc.SIGMA6, 0)
        src, dup = port.resample_plan(w, n, 0, ip)
        _, noise = ref.resample_draws(rc.SEED, pstep, rc.SIGMA6, int(dup.sum()))
        assert np.array_equal(port.resample_apply(s, src, dup, noise), want)
        key = "n%d_d%d" % (n, dead)
        out[key + "_states"] = want
        out[key + "_initial_p"] = np.float32(ip)
        out[key + "_noise"] = noise
        out[key + "_source"] = src
        out[key + "_dup"] = dup
        for n_out in rc.resize_targets(n):
            out[key + "_resize%d" % n_out] = ref.resize(s, w, n_out)[0]
        print("resample", key, "duplicates", int(dup.sum()), "dead particles picked", int((w[src] == 0).sum()))
    # the draw behind the upstream KAT test/src/test_pf.cpp:210-289 (engine seed 12345, pstep = sum(probs) / 5)
    small = np.float32(1.0e-06)
    for name, probs in (("first", [small, 0.2, 0.2, 0.2, np.float32(0.4) - small]),
                        ("last", [0.2, 0.2, 0.2, np.float32(0.4) - small, small])):
        pstep = port.resample_pstep(np.array(probs, np.float32), 5)
        out["kat_initial_p_" + name] = np.float32(ref.resample_draws(12345, pstep, np.zeros(6, np.float32), 0)[0])
    np.savez_compressed(os.path.join(HERE, "resample.npz"), **out)


def surface_golden():
    """Output of tests/cpp/surface_check.cpp built against the reference's own classes (oracle/Makefile, target `ref`):
    getMaxSearchRange / setGlobalLocalizationStatus / filter / refreshParameters / getSinTotalRef / getFilterLabelMax."""
    import subprocess
    exe = os.path.join(HERE, "..", "..", "oracle", "_ref", "surface_ref.bin")
    subprocess.run([exe, os.path.join(HERE, "surface_ref.dat")], check=True)
    print("surface", os.path.getsize(os.path.join(HERE, "surface_ref.dat")), "bytes")


if __name__ == "__main__":
    main()
    resample_goldens()
    surface_golden()
