"""CPU tests: the plain-C restatement against the committed outputs of the real reference (tests/golden/)."""
import numpy as np
import pytest

import golden_util
import kats
from oracle import pyoracle


@pytest.mark.parametrize("name", golden_util.SCENE_FIXTURES)
def test_port_reproduces_reference_goldens(name):
    g, sc, dw, bkw = golden_util.load(name)
    o = pyoracle.Oracle("port")
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=dw)
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    o.set_beam_params(pyoracle.BeamParams(**bkw))
    lik, quality = o.likelihood_measure(sc.poses, sc.scan_lik)
    np.testing.assert_array_equal(lik, g["lik"])
    np.testing.assert_array_equal(quality, g["quality"])
    beam, _ = o.beam_measure(sc.poses, sc.scan_beam, sc.scan_beam_label, sc.origins)
    np.testing.assert_array_equal(beam, g["beam"])
    begin, end = golden_util.rays_for(sc, 99)
    st, hit = o.beam_status(begin, end)
    np.testing.assert_array_equal(st, g["status"])
    np.testing.assert_array_equal(hit, g["hit"])
    upd = o.measure_update(sc.poses, sc.weights, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins,
                           sc.odom_err, float(g["odom_sigma"]))
    np.testing.assert_array_equal(upd["weights"], g["upd_weights"])
    assert np.float32(upd["entropy"]) == g["upd_entropy"]
    assert np.float32(upd["match_ratio_min"]) == g["upd_ratio"][0]
    assert np.float32(upd["match_ratio_max"]) == g["upd_ratio"][1]


def test_port_reproduces_beam_wall_fixture():
    """The upstream fixture of test/src/test_beam_likelihood.cpp:81-210 (2 modes x 6 hit ranges, DDA raycaster)."""
    g = np.load(golden_util.os.path.join(golden_util.HERE, "golden", "beam_wall_fixture.npz"))
    raw_pc, pc_map = kats.beam_wall_fixture()
    pc = raw_pc[(raw_pc[:, 2] > -0.3) & (raw_pc[:, 2] < 4.1)]
    xs = (0.1 * np.arange(-50, 50)).astype(np.float32)
    end = np.stack([xs, np.zeros_like(xs), np.zeros_like(xs)], 1)
    for mode in (0, 1):
        for k, hr in enumerate((0.0, 0.2, 0.4, 0.6, 0.8, 1.0)):
            o = pyoracle.Oracle("port", 10.0, 1.0)
            o.set_map(pc_map, None, dist_weight=None)
            o.set_beam_params(pyoracle.BeamParams(num_points=len(raw_pc), hit_range=hr, dda_grid_size=0.1,
                                                  add_penalty_short_only_mode=bool(mode)))
            st, _ = o.beam_status(np.zeros_like(end), end)
            np.testing.assert_array_equal(st, g["status_m%d_h%d" % (mode, k)])
            for i in (0, 30, 55, 70, 71, 75, 99):
                pose = np.array([[xs[i], 0, 0, 0, 0, 0, 1]], np.float32)
                lik = o.beam_measure(pose, pc, np.zeros(len(pc), np.uint32), pose[:, :3])[0][0]
                assert lik == g["lik_m%d_h%d" % (mode, k)][i]
