"""Loads tests/golden/*.npz (outputs of the real reference, see tests/golden/make_golden.py) and regenerates the
seeded inputs they belong to."""
import ast
import os

import numpy as np

from mcl_3dl_amd.synthetic import make_scene

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE_FIXTURES = ["c1_unit", "c1_default_weight", "label_wall"]


def load(name):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    skw = ast.literal_eval(str(g["scene_kwargs"]))
    bkw = ast.literal_eval(str(g["beam_kwargs"]))
    sc = make_scene(**skw)
    chk = np.array([sc.map_xyz.sum(dtype=np.float64), sc.scan_lik.sum(dtype=np.float64),
                    sc.poses.sum(dtype=np.float64), sc.scan_beam.sum(dtype=np.float64)])
    # the fixture is only meaningful if the seeded generator reproduces the inputs it was made from
    np.testing.assert_array_equal(chk, g["input_checksum"], err_msg="synthetic inputs changed: regenerate the goldens")
    return g, sc, tuple(float(v) for v in g["dist_weight"]), bkw


def rays_for(sc, seed, n=3000):
    rng = np.random.default_rng(seed)
    half = sc.meta["n"] * sc.meta["spacing"] / 2
    begin = rng.uniform(-half * 1.05, half * 1.05, (n, 3)).astype(np.float32)
    end = (begin + rng.normal(0, 2.0, (n, 3))).astype(np.float32)
    return begin, end


def odom_factor(odom_err, sigma):
    """NormalLikelihood<float>(sigma)(odom_err_integ_lin.norm()), include/mcl_3dl/nd.h:41-58 — what the caller of the
    C ABI passes as `extra`."""
    e = odom_err.astype(np.float32)
    x = np.sqrt((e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]).astype(np.float32)
    a = np.float32(1.0 / np.sqrt(2.0 * np.pi * float(sigma) * float(sigma)))
    sq2 = np.float32(float(sigma) * float(sigma) * 2.0)
    return (a * np.exp((-x * x / sq2).astype(np.float32)).astype(np.float32)).astype(np.float32)
