"""GPU test of the drop-in boundary: tests/cpp/adapter_demo.bin is a miniature of the reference node's call
site compiled against the DROP-IN C++ headers (same class names as the reference: LidarMeasurementModelLikelihood,
LidarMeasurementModelBeam, pf::ParticleFilter) — code written for the reference's plugin surface, running on the HIP
engine through the C ABI.  Its output is compared with the CPU oracle's measure_update on the same scene."""
import os
import struct
import subprocess

import numpy as np
import pytest

from mcl_3dl_amd.synthetic import make_scene
from oracle import pyoracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "tests", "cpp", "adapter_demo.bin")


def need_binary(path):
    """The adapter programs compile against the reference's own headers (pf.h, chunked_kdtree.h, ...), so they are built
    where /root/reference exists (__graft_entry__.build()) and travel to the GPU box as files. Missing on a box that has
    no reference tree: nothing to run. Missing where the reference IS present: the build is broken."""
    if os.path.exists(path):
        return
    if os.path.exists("/root/reference/include/mcl_3dl/pf.h"):
        pytest.fail(os.path.relpath(path, ROOT) + " missing although /root/reference is present: run __graft_entry__.build()")
    pytest.skip(os.path.relpath(path, ROOT) + " was not shipped and cannot be built here (no reference headers on this box)")


def write_scene(path, sc, dist_weight, beam_num_points, short_only, filter_label_max, sigma):
    with open(path, "wb") as f:
        f.write(struct.pack("<8Q", len(sc.map_xyz), len(sc.poses), len(sc.scan_lik), len(sc.scan_beam),
                            len(sc.origins), beam_num_points, int(short_only), filter_label_max))
        dw = dist_weight if dist_weight is not None else (1.0, 1.0, 1.0)
        f.write(struct.pack("<5f", dw[0], dw[1], dw[2], 0.0 if dist_weight is None else 1.0, sigma))
        for a, dt in ((sc.map_xyz, np.float32), (sc.map_label, np.uint32), (sc.poses, np.float32),
                      (sc.odom_err, np.float32), (sc.weights, np.float32), (sc.scan_lik, np.float32),
                      (sc.scan_beam, np.float32), (sc.scan_beam_label, np.uint32), (sc.origins, np.float32)):
            f.write(np.ascontiguousarray(a, dtype=dt).tobytes())


@pytest.mark.parametrize("dist_weight", [(1.0, 1.0, 5.0), None])
def test_node_call_site_through_drop_in_classes(tmp_path, oracle_kind, dist_weight):
    need_binary(DEMO)
    sc = make_scene(n=91, n_p=96, n_s=777, n_b=40, seed=21, label_wall=2)
    sigma, flmax = 0.6, 1
    scene, result = str(tmp_path / "scene.bin"), str(tmp_path / "result.bin")
    write_scene(scene, sc, dist_weight, 40, True, flmax, sigma)
    proc = subprocess.run([DEMO, scene, result], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    raw = np.fromfile(result, dtype=np.float32)
    n = len(sc.poses)
    w, lik, beam, quality, tail = raw[:n], raw[n:2 * n], raw[2 * n:3 * n], raw[3 * n:4 * n], raw[4 * n:4 * n + 6]
    np.testing.assert_array_equal(raw[4 * n + 6:].reshape(-1, 3), sc.scan_lik)   # filter() leaves the sampler's order alone

    o = pyoracle.Oracle(oracle_kind)
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=dist_weight)
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    o.set_beam_params(pyoracle.BeamParams(num_points=40, filter_label_max=flmax))
    want = o.measure_update(sc.poses, sc.weights, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins,
                            sc.odom_err, sigma)
    np.testing.assert_allclose(lik, want["lik"], rtol=1e-5)
    np.testing.assert_array_equal(beam, want["beam"])
    np.testing.assert_array_equal(quality, want["quality"])
    np.testing.assert_allclose(w, want["weights"], rtol=1e-5)
    np.testing.assert_allclose(tail[0], want["entropy"], rtol=1e-5)
    assert tail[1] == np.float32(want["match_ratio_min"]) and tail[2] == np.float32(want["match_ratio_max"])
    # the call outside pf::measure (batch of one) gives the same answer as the batched one for particle 0
    assert tail[3] == lik[0] and tail[4] == quality[0]
    st, _ = o.beam_status(sc.poses[:1, :3], sc.poses[:1, :3] + np.array([[3.0, 0.5, -0.2]], np.float32))
    assert int(tail[5]) == int(st[0])


@pytest.mark.parametrize("n_p,n_s", [(96, 777), (2000, 5000)])
def test_node_call_site_in_engine_order_is_bit_identical(tmp_path, oracle_kind, n_p, n_s):
    """MCL3DL_HIP_ENGINE_ORDER=1: the likelihood model's filter() hands the node its sampled cloud in the engine's scan
    order and the likelihood kernel runs the reference's float recurrence over it (strict_order = 3). The reference's own
    classes, given the cloud the node holds, return the same floats bit for bit — the node's source untouched, no N_s x N_p
    term array, no replay pass."""
    from mcl_3dl_amd import capi
    need_binary(DEMO)
    dist_weight = (1.0, 1.0, 5.0)
    sc = make_scene(n=91, n_p=n_p, n_s=n_s, n_b=40, seed=22, label_wall=2)
    sigma, flmax = 0.6, 1
    scene, result = str(tmp_path / "scene.bin"), str(tmp_path / "result.bin")
    write_scene(scene, sc, dist_weight, 40, True, flmax, sigma)
    proc = subprocess.run([DEMO, scene, result], capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, MCL3DL_HIP_ENGINE_ORDER="1"))
    assert proc.returncode == 0, proc.stdout + proc.stderr
    raw = np.fromfile(result, dtype=np.float32)
    n = len(sc.poses)
    w, lik, beam, quality, tail = raw[:n], raw[n:2 * n], raw[2 * n:3 * n], raw[3 * n:4 * n], raw[4 * n:4 * n + 6]
    held = raw[4 * n + 6:].reshape(-1, 3)
    order = capi.scan_order_host(sc.scan_lik)
    assert np.array_equal(np.sort(order), np.arange(n_s, dtype=np.uint32)) and not np.array_equal(order, np.arange(n_s))
    np.testing.assert_array_equal(held, sc.scan_lik[order])
    np.testing.assert_array_equal(capi.scan_order_host(held), np.arange(n_s, dtype=np.uint32))  # an ordered cloud stays put
    o = pyoracle.Oracle(oracle_kind)
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=dist_weight)
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    o.set_beam_params(pyoracle.BeamParams(num_points=40, filter_label_max=flmax))
    want = o.measure_update(sc.poses, sc.weights, held, sc.scan_beam, sc.scan_beam_label, sc.origins, sc.odom_err, sigma)
    np.testing.assert_array_equal(lik, want["lik"])
    np.testing.assert_array_equal(beam, want["beam"])
    np.testing.assert_array_equal(quality, want["quality"])
    np.testing.assert_array_equal(w, want["weights"])     # pf.h normalises on the host in this route: the reference's own code
    assert tail[3] == lik[0] and tail[4] == quality[0]


def test_rest_of_the_plugin_surface(tmp_path):
    """getMaxSearchRange, refreshParameters, setGlobalLocalizationStatus, filter, getSinTotalRef, getFilterLabelMax of the
    drop-in classes against the reference's own classes: tests/cpp/surface_check.cpp is built against both, the two
    record streams must be byte-identical."""
    exe = os.path.join(ROOT, "tests", "cpp", "surface_gpu.bin")
    need_binary(exe)
    got_path = str(tmp_path / "surface_gpu.out")
    proc = subprocess.run([exe, got_path], capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    got = open(got_path, "rb").read()
    golden = open(os.path.join(ROOT, "tests", "golden", "surface_ref.dat"), "rb").read()
    assert len(golden) > 10000
    assert got == golden
    live = os.path.join(ROOT, "oracle", "_ref", "surface_ref.bin")
    if os.path.exists(live):  # the reference's classes, run on this box
        ref_path = str(tmp_path / "surface_ref.out")
        subprocess.run([live, ref_path], check=True, timeout=300)
        assert open(ref_path, "rb").read() == got
