"""A device group's particles RESIDENT on its GPUs (mcl_3dl_amd/csrc/api_group_state.inl): a whole filter iteration for N GPUs
behind the C ABI — pf::measure, expectationBiased / max / covariance, resample / resizeParticle (include/mcl_3dl/pf.h:187-225,
252-279, 294-390, 399-436) — without a per-update pose upload. The GPU box has one MI355X, so
  * N = 1 (direct, and through the sharded path with RCCL running with one rank: ncclAllReduce AND ncclAllGather) must equal
    the plain context bit for bit;
  * N = 2, 3, 5 contexts on the SAME device (collective = host) must reproduce the unsharded filter: likelihood / ratio /
    beam bit-identical, weights and moments to the fp64 sums' association (2e-7), and — given the SAME weights — the very same
    resampling plan and new generation."""
import numpy as np
import pytest

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu
DW = (1.0, 1.0, 5.0)
N_P = 203


@pytest.fixture(scope="module")
def scene():
    return make_scene(n=91, n_p=N_P, n_s=1200, n_b=48, seed=77)


@pytest.fixture(scope="module")
def start(scene):
    rng = np.random.default_rng(5)
    st = np.zeros((N_P, 13), np.float32)
    st[:, :7] = scene.poses
    st[:, 7:] = rng.normal(0.0, 0.05, (N_P, 6)).astype(np.float32)
    w0 = rng.uniform(0.5, 1.5, N_P).astype(np.float32)
    w0 /= w0.sum()
    extra = rng.uniform(0.2, 1.0, N_P).astype(np.float32)
    bias = rng.uniform(0.5, 1.0, N_P).astype(np.float32)
    return st, w0, extra, bias


def configure(obj, sc):
    obj.set_map(sc.map_xyz, sc.map_label, stamp=7100, dist_weight=DW)
    obj.set_likelihood_params()
    obj.set_beam_params(num_points=48)


def noise_for(n_dup, seed):
    rng = np.random.default_rng(seed)
    nz = np.zeros((n_dup, 13), np.float32)
    nz[:, :3] = rng.normal(0, 0.05, (n_dup, 3))
    q = rng.normal(0, 0.02, (n_dup, 4))
    q[:, 3] = 1.0
    nz[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    nz[:, 7:] = rng.normal(0, 0.01, (n_dup, 6))
    return nz.astype(np.float32)


def reference_iteration(engine, sc, st, w, extra, bias, n_out, mode):
    """The same steps on ONE context with host arrays."""
    upd = engine.measure_update(st[:, :7], w, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, extra=extra)
    return upd


@pytest.mark.parametrize("devices,collective,direct", [([0], None, 1), ([0], None, 0), ([0, 0], "host", 1),
                                                       ([0, 0, 0], "host", 1), ([0] * 5, "host", 1)])
def test_filter_iterations_over_resident_shards(engine, scene, start, devices, collective, direct):
    sc = scene
    st, w0, extra, bias = start
    configure(engine, sc)
    exact = len(devices) == 1
    g = capi.Group(devices, collective=collective)
    try:
        configure(g, sc)
        g.set_option("direct_single", direct)
        g.upload_state(st, w0)
        assert g.resident() == N_P
        cur_st, cur_w = st, w0
        for it, (n_out, mode) in enumerate([(0, 0), (150, 1), (260, 1), (0, 0)]):
            n = len(cur_st)
            ex = extra[:n] if n <= N_P else np.resize(extra, n)
            bi = bias[:n] if n <= N_P else np.resize(bias, n)
            # ---- pf::measure
            got = g.update_resident(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, extra=ex)
            want = engine.measure_update(cur_st[:, :7], cur_w, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins, extra=ex)
            for k in ("lik", "quality", "beam"):
                np.testing.assert_array_equal(got[k], want[k], err_msg="%s iteration %d" % (k, it))
            assert got["restored"] == want["restored"] and got["match_ratio_min"] == want["match_ratio_min"]
            if exact:
                np.testing.assert_array_equal(got["weights"], want["weights"])
                assert got["entropy"] == want["entropy"]
            else:
                np.testing.assert_allclose(got["weights"], want["weights"], rtol=1e-6)  # (one GPU adds up to 1024 weights as the reference does, float by float; shards add theirs in an fp64 all-reduce)
                np.testing.assert_allclose(got["entropy"], want["entropy"], rtol=1e-6)
            # the weights stayed on the devices
            st_dev, w_dev = g.download_state()
            np.testing.assert_array_equal(st_dev, cur_st)
            np.testing.assert_array_equal(w_dev, got["weights"])
            # ---- expectationBiased + max, covariance (reference: one context on the group's own weights)
            mean, total, im, ib = g.expectation(bi)
            wmean, wtotal, wim, wib = engine.expectation(cur_st[:, :7], w_dev, bi)
            assert (im, ib) == (wim, wib)
            if exact:
                np.testing.assert_array_equal(mean, wmean)
                assert total == wtotal
            else:
                np.testing.assert_allclose(mean, wmean, rtol=2e-6, atol=1e-7)
                np.testing.assert_allclose(total, wtotal, rtol=2e-7)
            cov = g.covariance(wmean)
            wcov = engine.covariance(cur_st[:, :7], w_dev, wmean)
            if exact:
                np.testing.assert_array_equal(cov, wcov)
            else:
                np.testing.assert_allclose(cov, wcov, rtol=2e-5, atol=1e-9)
            # ---- resample / resizeParticle: same weights in -> same plan, same new generation, bit for bit
            pstep = g.resample_begin(n_out)
            wpstep = engine.resample_begin(w_dev, n_out or None)
            assert pstep == wpstep
            src, dup, n_dup = g.resample_plan(mode, 0.37 * pstep)
            wsrc, wdup, wn_dup = engine.resample_plan(mode, 0.37 * pstep)
            np.testing.assert_array_equal(src, wsrc)
            np.testing.assert_array_equal(dup, wdup)
            assert n_dup == wn_dup
            nz = noise_for(n_dup, 100 + it)
            g.resample_apply(nz)
            want_st = engine.resample_apply(cur_st, nz)
            new_st, new_w = g.download_state()
            np.testing.assert_array_equal(new_st, want_st)
            n_new = n_out or n
            assert g.resident() == n_new and len(new_st) == n_new
            np.testing.assert_array_equal(new_w, np.full(n_new, np.float32(1.0) / np.float32(n_new), np.float32))
            cur_st, cur_w = new_st, new_w
        stats = g.collective_stats()
        if len(devices) == 1 and direct == 1:
            assert stats == dict(rccl=0, host=0)
        elif len(devices) == 1:
            assert stats["rccl"] == 8 and stats["host"] == 0   # four all-reduces + four all-gathers, one rank each
        else:
            assert stats["host"] == 8 and stats["rccl"] == 0
    finally:
        g.close()


def test_more_devices_than_particles_and_call_order(scene, start):
    sc = scene
    st, w0, extra, bias = start
    g = capi.Group([0, 0, 0, 0], collective="host")
    try:
        configure(g, sc)
        with pytest.raises(capi.EngineError, match="resident"):
            g.update_resident(sc.scan_lik)
        g.upload_state(st[:3], None)           # three particles over four contexts: one shard is empty
        got = g.update_resident(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        assert abs(float(got["weights"].astype(np.float64).sum()) - 1.0) < 1e-6
        mean, total, im, ib = g.expectation()
        assert 0 <= im < 3 and abs(total - 1.0) < 1e-6
        with pytest.raises(capi.EngineError, match="before"):
            g.resample_plan(0, 0.0)
        pstep = g.resample_begin(0)
        src, dup, n_dup = g.resample_plan(0, 0.5 * pstep)
        g.resample_apply(noise_for(n_dup, 9))
        new_st, new_w = g.download_state()
        assert new_st.shape == (3, 13) and np.isfinite(new_st).all()
    finally:
        g.close()


@pytest.mark.parametrize("devices,collective", [([0], None), ([0, 0, 0], "host")])
def test_non_resident_calls_between_resident_ones_do_not_swap_the_poses(scene, start, devices, collective):
    """update_resident / expectation / covariance read the poses from the context's pose buffer, which
    mcl3dl_hip_group_upload_poses, group_measure_batch and the non-resident group_measure_update also write — with the same
    particle count the resident calls used to evaluate the FOREIGN poses and report success. They now re-derive the poses
    from the resident states whenever somebody else has written the buffer."""
    sc = scene
    st, w0, extra, bias = start
    other = make_scene(n=91, n_p=N_P, n_s=1200, n_b=48, seed=78).poses   # same count, different poses
    g = capi.Group(devices, collective=collective)
    h = capi.Group(devices, collective=collective)
    try:
        for obj in (g, h):
            configure(obj, sc)
            obj.upload_state(st, w0)
        args = (sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        # h: resident calls only; g: the same calls with foreign-pose traffic in between each of them
        want = h.update_resident(*args, extra=extra)
        g.measure_update(other, w0.copy(), *args)
        got = g.update_resident(*args, extra=extra)
        for k in ("lik", "quality", "beam", "weights"):
            np.testing.assert_array_equal(got[k], want[k])
        want_m = h.expectation(bias)
        g.upload_poses(other)
        got_m = g.expectation(bias)
        np.testing.assert_array_equal(got_m[0], want_m[0])
        assert got_m[1:] == want_m[1:]
        want_c = h.covariance(want_m[0])
        g.measure_batch(other, sc.scan_lik)
        got_c = g.covariance(want_m[0])
        np.testing.assert_array_equal(got_c, want_c)
        # ... and a second resident update still starts from the weights the first one left
        want2 = h.update_resident(*args)
        g.measure_update(other, w0.copy(), *args)
        got2 = g.update_resident(*args)
        np.testing.assert_array_equal(got2["weights"], want2["weights"])
    finally:
        g.close()
        h.close()
