"""pf::measure as ONE launch between pf_fused_max (one work-group) and 16 384 particles (update_kernels.h:pf_ticket_kernel: the
split form's partial kernel, then — behind an arrival-ticket tree — its reduce and its apply by the last work-group) against the
three launches: the same arithmetic in the same association, so weights, entropy, ratio bounds and the restore rule are
bit-identical (pf.h:252-279)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def inputs(n, seed, dead=False):
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.2, 1.0, n).astype(np.float32)
    w /= w.sum()
    lik = np.zeros(n, np.float32) if dead else rng.uniform(0.0, 3.0, n).astype(np.float32)
    beam = rng.uniform(0.1, 1.0, n).astype(np.float32)
    extra = rng.uniform(0.5, 1.5, n).astype(np.float32)
    ratio = rng.uniform(0.0, 1.0, n).astype(np.float32)
    return w, lik, beam, extra, ratio


def both(engine, n, seed, use=(True, True, True), dead=False):
    w, lik, beam, extra, ratio = inputs(n, seed, dead)
    out = {}
    try:
        for on in (0, 1):
            engine.set_option("pf_ticket", on)
            out[on] = engine.pf_measure(w, lik, beam if use[0] else None, extra if use[1] else None, ratio if use[2] else None)
    finally:
        engine.set_option("pf_ticket", 1)
    return out[0], out[1], w


def same(a, b):
    np.testing.assert_array_equal(a["weights"], b["weights"])
    assert a["restored"] == b["restored"]
    if not a["restored"]:
        assert a["entropy"] == b["entropy"]
    assert a["match_ratio_min"] == b["match_ratio_min"] and a["match_ratio_max"] == b["match_ratio_max"]


@pytest.mark.parametrize("n", [1025, 1500, 2048, 4096, 4097, 8192, 10000, 16383, 16384, 16385, 40000])
def test_one_launch_equals_three(engine, n):
    a, b, _ = both(engine, n, n)
    same(a, b)
    assert abs(float(b["weights"].sum()) - 1.0) < 1e-4


@pytest.mark.parametrize("use", [(False, False, False), (True, False, True), (False, True, False)])
def test_optional_factors(engine, use):
    a, b, _ = both(engine, 4096, 7, use=use)
    same(a, b)


def test_dead_filter_restores_the_weights(engine):
    a, b, w = both(engine, 5000, 3, dead=True)
    same(a, b)
    assert b["restored"] is True
    np.testing.assert_array_equal(b["weights"], w)


def test_back_to_back_launches_reuse_the_tickets(engine):
    """The last work-group leaves the counters at zero: a hundred launches in a row, every fifth on another size."""
    ref = {}
    for n in (4096, 9000):
        engine.set_option("pf_ticket", 0)
        ref[n] = engine.pf_measure(*inputs(n, 11))
    engine.set_option("pf_ticket", 1)
    for k in range(100):
        n = 9000 if k % 5 == 0 else 4096
        same(engine.pf_measure(*inputs(n, 11)), ref[n])


def test_device_resident_update_and_graph_replay(engine):
    from mcl_3dl_amd.synthetic import make_scene
    sc = make_scene(n=61, n_p=4096, n_s=600, n_b=0, seed=2)
    engine.set_map(sc.map_xyz, sc.map_label, stamp=9200)
    engine.set_likelihood_params()
    engine.upload_scan(sc.scan_lik, None, None, sc.origins)
    dev = torch.device("cuda", 0)
    d_pose = torch.from_numpy(np.ascontiguousarray(sc.poses)).to(dev)
    w0 = torch.full((4096,), 1.0 / 4096, device=dev)
    res = {}
    try:
        for tag, on, graph in (("three", 0, 0), ("one", 1, 0), ("graph", 1, 1)):
            engine.set_option("pf_ticket", on)
            engine.set_option("use_graph", graph)
            d_w = w0.clone()
            d_stats = torch.zeros(4, device=dev)
            for _ in range(5):
                d_w.copy_(w0)
                torch.cuda.synchronize()
                engine.update_device(d_pose, 4096, d_w, d_stats)
                engine.synchronize()
            res[tag] = (d_w.cpu().numpy().copy(), d_stats.cpu().numpy().copy())
    finally:
        engine.set_option("pf_ticket", 1)
        engine.set_option("use_graph", 0)
    for tag in ("one", "graph"):
        np.testing.assert_array_equal(res["three"][0], res[tag][0])
        np.testing.assert_array_equal(res["three"][1], res[tag][1])
