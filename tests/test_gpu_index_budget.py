"""Footprint of the candidate-voxel index: `cand_start` was 32 x the map at unit weight and 117 x
at the shipped dist_weight (z x 5, src/parameters.cpp:108-110), with no bound. Round 5:
  * voxels may be BOXES whose edges follow the dist_weight axis by axis (option cand_aniso; every step of the candidate
    proof is per axis): the records of a z x 5 map shrink to those of the unit-weight map;
  * option index_budget_bytes (default: a quarter of the device's memory): cubes while they fit, boxes when they do not,
    coarser voxels after that, a clean error when even 1.5 r voxels do not fit.
Whatever the voxel shape, results are the same bits (the minimum over a superset of the candidates is the same minimum)."""
import numpy as np
import pytest

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_scene
from oracle import pyoracle

pytestmark = pytest.mark.gpu
DW = (1.0, 1.0, 5.0)


@pytest.fixture(scope="module")
def scene():
    return make_scene(n=91, n_p=120, n_s=2500, n_b=0, seed=41)


def measure(engine, sc, stamp, **opts):
    saved = {k: engine.get_option(k) for k in opts}
    try:
        for k, v in opts.items():
            engine.set_option(k, v)
        engine.set_option("strict_order", 1)
        engine.set_map(sc.map_xyz, sc.map_label, stamp=stamp, dist_weight=DW)
        engine.set_likelihood_params()
        lik, ratio, _ = engine.measure_batch(sc.poses, sc.scan_lik)
        info = dict(bytes=engine.get_option("index_record_bytes"), boxes=int(engine.get_option("cand_aniso_active")),
                    edges=[engine.get_option("cand_edge_ratio_" + a) for a in "xyz"], stats=engine.index_stats())
        return lik, ratio, info
    finally:
        engine.set_option("strict_order", 2)
        for k, v in saved.items():
            engine.set_option(k, v)


def test_boxes_and_coarser_voxels_give_the_same_bits_in_less_memory(engine, oracle_kind, scene):
    sc = scene
    o = pyoracle.Oracle(oracle_kind)
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=DW)
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    want_lik, want_q = o.likelihood_measure(sc.poses, sc.scan_lik)
    cubes = measure(engine, sc, 7001, cand_aniso=0, index_budget_bytes=0)
    boxes = measure(engine, sc, 7002, cand_aniso=1, index_budget_bytes=0)
    boxes3 = measure(engine, sc, 7003, cand_aniso=1, cand_aniso_max=3, index_budget_bytes=0)
    assert cubes[2]["boxes"] == 0 and cubes[2]["edges"] == pytest.approx([0.5, 0.5, 0.5])
    assert boxes[2]["boxes"] == 1 and boxes[2]["edges"] == pytest.approx([0.5, 0.5, 2.5])
    assert boxes3[2]["edges"] == pytest.approx([0.5, 0.5, 1.5])
    assert boxes[2]["bytes"] < 0.4 * cubes[2]["bytes"], (boxes[2]["bytes"], cubes[2]["bytes"])
    # a budget between the two: the automatic mode goes from cubes to boxes; below the boxes: coarser voxels as well
    mid = 0.5 * (boxes[2]["bytes"] + cubes[2]["bytes"])
    auto_mid = measure(engine, sc, 7004, cand_aniso=2, index_budget_bytes=mid)
    auto_low = measure(engine, sc, 7005, cand_aniso=2, index_budget_bytes=0.5 * boxes[2]["bytes"])
    iso_low = measure(engine, sc, 7006, cand_aniso=0, index_budget_bytes=0.3 * cubes[2]["bytes"])
    assert auto_mid[2]["boxes"] == 1 and auto_mid[2]["edges"][0] == pytest.approx(0.5) and auto_mid[2]["bytes"] <= mid
    assert auto_low[2]["boxes"] == 1 and auto_low[2]["edges"][0] > 0.51 and auto_low[2]["bytes"] <= 0.5 * boxes[2]["bytes"]
    assert iso_low[2]["boxes"] == 0 and iso_low[2]["edges"][0] > 0.51 and iso_low[2]["bytes"] <= 0.3 * cubes[2]["bytes"]
    for name, (lik, ratio, info) in dict(cubes=cubes, boxes=boxes, boxes3=boxes3, auto_mid=auto_mid, auto_low=auto_low,
                                         iso_low=iso_low).items():
        np.testing.assert_array_equal(ratio, want_q, err_msg=name)
        np.testing.assert_array_equal(lik, want_lik, err_msg=name)


def test_default_budget_is_a_quarter_of_the_device_and_small_maps_keep_cubes(engine, scene):
    lik, ratio, info = measure(engine, scene, 7010)
    assert info["boxes"] == 0 and info["edges"] == pytest.approx([0.5, 0.5, 0.5])
    budget = engine.get_option("index_budget_in_use")
    assert 30e9 < budget < 100e9, budget     # 288 GB of HBM3E: 72 GB


def test_an_index_that_cannot_fit_its_budget_is_a_clean_error(engine, scene):
    sc = scene
    with pytest.raises(capi.EngineError, match="budget"):
        measure(engine, sc, 7020, index_budget_bytes=100000.0)
    # ... and the engine goes on: the next map (budget lifted by measure()'s finally) compiles and evaluates
    lik, ratio, info = measure(engine, sc, 7021)
    assert np.count_nonzero(lik) > 0 and info["bytes"] > 100000


def test_map_update_on_an_index_of_boxes_equals_a_fresh_one(engine, scene):
    sc = scene
    rng = np.random.default_rng(3)
    fresh = capi.Engine(0)
    try:
        for obj in (engine, fresh):
            obj.set_option("cand_aniso", 1)
        engine.set_map(sc.map_xyz, sc.map_label, stamp=7030, dist_weight=DW)
        engine.set_likelihood_params()
        engine.measure_batch(sc.poses[:4], sc.scan_lik[:64])     # builds the index
        half = 91 * 0.1 / 2
        for k in range(3):
            pts = rng.uniform(-half + 0.5, half - 0.5, (150, 3)).astype(np.float32)
            pts[:, 2] = rng.uniform(-half + 0.2, -half + 1.5, 150)
            engine.map_update(pts, None, leaf=(0.2, 0.2, 0.2), stamp=7031 + k)
            mx, ml = engine.map_download()
            fresh.set_map(mx, ml, stamp=7040 + k, dist_weight=DW)
            fresh.set_likelihood_params()
            a = engine.measure_batch(sc.poses, sc.scan_lik)
            b = fresh.measure_batch(sc.poses, sc.scan_lik)
            assert int(engine.get_option("cand_aniso_active")) == 1
            np.testing.assert_array_equal(a[1], b[1])
            np.testing.assert_allclose(a[0], b[0], rtol=2e-6)
    finally:
        engine.set_option("cand_aniso", 2)
        fresh.close()
