"""GPU tests of mcl3dl_hip_update_device: the fused device-resident update (measure + pf::measure) and its hipGraph
replay. The graph must give the very bits the three separate calls give, survive new scans / poses / weights of the
same size (those are data, not launch parameters), and start over when anything that shapes the launches changes."""
import numpy as np
import pytest

from mcl_3dl_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu


def separate_calls(eng, torch, d_pose, n_p, d_w, d_extra):
    dev = d_pose.device
    lik = torch.empty(n_p, dtype=torch.float32, device=dev)
    ratio = torch.empty_like(lik)
    beam = torch.empty_like(lik)
    pack = torch.zeros(4, dtype=torch.float64, device=dev)
    stats = torch.zeros(4, dtype=torch.float32, device=dev)
    w = d_w.clone()
    torch.cuda.synchronize()  # the engine runs on its own stream
    eng.measure_device(d_pose, n_p, lik, ratio, beam)
    eng.pf_partial_device(w, lik, beam, d_extra, ratio, n_p, pack, 0, 1)
    eng.pf_apply_device(w, n_p, pack, stats, 1)
    eng.synchronize()
    return w.cpu().numpy(), lik.cpu().numpy(), ratio.cpu().numpy(), beam.cpu().numpy(), stats.cpu().numpy()


@pytest.mark.parametrize("n_b", [0, 48])
def test_graph_replay_equals_separate_calls(engine, n_b):
    import torch
    dev = torch.device("cuda:0")
    sc = make_scene(n=91, n_p=128, n_s=600, n_b=n_b, seed=5)
    n_p = len(sc.poses)
    engine.set_map(sc.map_xyz, sc.map_label, stamp=7100 + n_b, dist_weight=(1.0, 1.0, 5.0))
    engine.set_likelihood_params()
    engine.set_beam_params(num_points=max(n_b, 1))
    rng = np.random.default_rng(1)
    d_pose = torch.from_numpy(sc.poses).to(dev)
    d_w = torch.empty(n_p, dtype=torch.float32, device=dev)
    d_extra = torch.from_numpy(rng.uniform(0.2, 0.4, n_p).astype(np.float32)).to(dev)
    d_lik, d_ratio, d_beam = (torch.empty(n_p, dtype=torch.float32, device=dev) for _ in range(3))
    d_stats = torch.zeros(4, dtype=torch.float32, device=dev)
    engine.set_option("use_graph", 1)
    before = engine.graph_stats()
    try:
        for it in range(6):
            # a new scan of the same size, moved particles, new prior weights: data only
            sel = rng.permutation(len(sc.scan_lik))
            scan = (sc.scan_lik[sel] + rng.normal(0, 0.005, sc.scan_lik.shape)).astype(np.float32)
            engine.upload_scan(scan, sc.scan_beam if n_b else None, sc.scan_beam_label if n_b else None, sc.origins)
            poses = sc.poses.copy()
            poses[:, :3] += rng.normal(0, 0.05, (n_p, 3)).astype(np.float32)
            d_pose.copy_(torch.from_numpy(poses))
            w0 = rng.uniform(0.5, 1.5, n_p).astype(np.float32)
            w0 /= w0.sum()
            d_w.copy_(torch.from_numpy(w0))
            torch.cuda.synchronize()
            want = separate_calls(engine, torch, d_pose, n_p, d_w, d_extra)
            engine.update_device(d_pose, n_p, d_w, d_stats, d_extra=d_extra, d_lik=d_lik, d_ratio=d_ratio, d_beam=d_beam)
            engine.synchronize()
            got = (d_w.cpu().numpy(), d_lik.cpu().numpy(), d_ratio.cpu().numpy(), d_beam.cpu().numpy(),
                   d_stats.cpu().numpy())
            for g, w_ in zip(got, want):
                np.testing.assert_array_equal(g, w_, err_msg="iteration %d" % it)
        st = engine.graph_stats()
        assert st["captures"] - before["captures"] == 1
        assert st["replays"] - before["replays"] == 5  # call 1 eager, call 2 captures + replays, calls 3-6 replay
        # a different scan size changes the launch geometry: eager, then a fresh capture
        engine.upload_scan(sc.scan_lik[:333], sc.scan_beam if n_b else None, sc.scan_beam_label if n_b else None,
                           sc.origins)
        for it in range(3):
            d_w.copy_(torch.from_numpy(w0))
            torch.cuda.synchronize()
            want = separate_calls(engine, torch, d_pose, n_p, d_w, d_extra)
            engine.update_device(d_pose, n_p, d_w, d_stats, d_extra=d_extra, d_lik=d_lik, d_ratio=d_ratio, d_beam=d_beam)
            engine.synchronize()
            np.testing.assert_array_equal(d_w.cpu().numpy(), want[0])
            np.testing.assert_array_equal(d_stats.cpu().numpy(), want[4])
        st2 = engine.graph_stats()
        assert st2["captures"] - st["captures"] == 1 and st2["replays"] - st["replays"] == 2
        # graphs off: same answers, no replays
        engine.set_option("use_graph", 0)
        d_w.copy_(torch.from_numpy(w0))
        torch.cuda.synchronize()
        engine.update_device(d_pose, n_p, d_w, d_stats, d_extra=d_extra)
        engine.synchronize()
        np.testing.assert_array_equal(d_w.cpu().numpy(), want[0])
        assert engine.graph_stats()["replays"] == st2["replays"]
    finally:
        engine.set_option("use_graph", 0)
        engine.set_likelihood_params()
        engine.set_beam_params()


def test_update_device_argument_errors(engine):
    import torch
    from mcl_3dl_amd import capi
    dev = torch.device("cuda:0")
    e = capi.Engine(0)
    d = torch.zeros(7 * 4, dtype=torch.float32, device=dev)
    with pytest.raises(capi.EngineError, match="no scan"):
        e.update_device(d, 4, d, d)
    with pytest.raises(capi.EngineError, match="null"):
        e.update_device(None, 4, d, d)
    e.close()
