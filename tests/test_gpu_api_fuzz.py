"""Stateful API-sequence fuzz. tests/test_gpu_fuzz.py randomises PARAMETERS; what found the
round-4 hazard (a lazy structure build recycling staging memory under a live update: profiles/r04ad_poll_all.txt) was the
ORDER of calls. Here every sequence draws >= 30 calls from

    set_map / map_update (inside and outside the base map's bounds) / set_likelihood_params / set_beam_params /
    set_option (EVERY key the library accepts — round 6: 38 of them, the fault-injection hook aside — with two to five values
    each: summation modes and their thresholds, kernel selection, index forms, staging, polling, ...) /
    measure_batch / measure_batch_begin.._wait.._end with calls in between and batches abandoned until a later _end /
    measure_update on pageable and on page-locked arrays / scan_begin + scan_finish + measure_device /
    resample_begin + _plan / expectation / the same update through a device group of two contexts

in the call order of the node (src/mcl_3dl.cpp:378-452, 1355-1369) shuffled, and checks EVERY result against the reference
(oracle/_ref, or the C port): match ratios, beam scores and resampling plans exactly, likelihoods exactly in the float-order
modes (strict_order 1: caller's order, 3: the engine's order) and within the fp64 mode's tolerance otherwise.

MCL3DL_FUZZ_SEQUENCES (default 300) x MCL3DL_FUZZ_CALLS (default 30); a failure prints the seed and the calls so far.
"""
import os

import numpy as np
import pytest

from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_scene
from oracle import pyoracle

pytestmark = pytest.mark.gpu

N_SEQ = int(os.environ.get("MCL3DL_FUZZ_SEQUENCES", "300"))
N_CALLS = int(os.environ.get("MCL3DL_FUZZ_CALLS", "30"))
CHUNK = 10
# the oracle's measure_update always applies the odometry factor of src/mcl_3dl.cpp:400,422-424; with no odometry error it is
# NormalLikelihood(0) = 1 / sqrt(2 pi sigma^2) (nd.h:41-58), sigma = 1: the engine is handed the same constant as `extra`
ND0 = np.float32(1.0 / np.sqrt(2.0 * np.pi))
RTOL_FP64 = 3e-5   # the fp64 mode against the reference's sequential float sum (tests/test_gpu_fuzz.py holds the same gate)

# EVERY key mcl3dl_hip_set_option accepts (tests/test_abi.py::test_every_option_is_in_the_api_fuzz_pool keeps this list and
# api_support.inl in step) except the fault-injection hook test_late_structures, which MCL3DL_FUZZ_INJECT switches on
DEFAULTS = dict(poll_sync=2, strict_order=2, update_stage=1, update_zero_copy=1, update_small=1, pf_fused=1,
                lik_coop=1, lik_defer=1, overlap_models=1, batch_slice=0, scan_order_device=4096, lik_tiled_min=1024,
                poll_spin_us=2000, lik_index=2, cand_aniso=2, chain_ppl=0,
                beam_prepare=1, cand_aniso_max=8, cand_bound=1, cand_packed=1, cand_phase=0.5, cand_prune_coop=1, cand_record_parts=0,
                cand_voxel_ratio=0.0, dda_overlay=1, grid_build_host=0, index_budget_bytes=-1, lik_group=0, lik_small=1,
                lik_tiled=1, scan_presorted=0, strict_auto_max_bytes=0, strict_auto_min=28147, strict_chunk=0,
                strict_exact_max=4096, timing_mask=0xFFFFFFFF, update_small_conformant=0, update_small_max=512)
CHOICES = dict(poll_sync=(0, 1, 2), strict_order=(0, 1, 2, 2, 3), update_stage=(0, 1), update_zero_copy=(0, 1),
               update_small=(0, 1), pf_fused=(0, 1), lik_coop=(0, 1), lik_defer=(0, 1, 2),
               overlap_models=(0, 1), batch_slice=(0, 64, 128), scan_order_device=(0, 512, 4096),
               lik_tiled_min=(256, 1024), poll_spin_us=(0, 50, 2000), lik_index=(2, 2, 0), cand_aniso=(2, 1, 0),
               chain_ppl=(0, 1, 4),
               beam_prepare=(1, 0), cand_aniso_max=(8, 3), cand_bound=(1, 0), cand_packed=(1, 0), cand_phase=(0.5, 0.0, 0.3), cand_prune_coop=(1, 0),
               cand_record_parts=(0, 4, 8), cand_voxel_ratio=(0.0, 0.5, 0.3), dda_overlay=(1, 0), grid_build_host=(0, 1),
               index_budget_bytes=(-1, 0, 2e7), lik_group=(0, 4, 8, 16, 32), lik_small=(1, 0), lik_tiled=(1, 0),
               scan_presorted=(0, 1), strict_auto_max_bytes=(0, 1 << 18), strict_auto_min=(28147, 1500), strict_chunk=(0, 1024),
               strict_exact_max=(4096, 0, 800), timing_mask=(0xFFFFFFFF, 1, 0), update_small_conformant=(0, 1),
               update_small_max=(512, 64))
# the options that select HOW the likelihood terms are added up: an engine compared bit for bit with another one needs them all
# (scan_presorted: with strict_order 3 the sum follows the device scan array's order — the caller's own when it is taken as it is)
SUM_OPTIONS = ("strict_order", "strict_exact_max", "strict_auto_min", "strict_auto_max_bytes", "strict_chunk", "scan_presorted")


@pytest.fixture(scope="module")
def world():
    """Two small maps (a lattice cube and a cube of displaced points), a pool of poses and scans for each."""
    out = []
    for k, (n, jitter) in enumerate(((41, 0.0), (45, 0.03))):
        sc = make_scene(n=n, n_p=2300, n_s=2600, n_b=64, seed=500 + k, map_jitter=jitter, label_wall=2,
                        lik_clip=(0.5, 4.0, -2.0, 2.0), beam_clip=(0.5, 3.0, -2.0, 2.0))
        out.append(sc)
    return out


class Model:
    """What the engine has been told so far — enough to tell the oracle the same."""

    def __init__(self):
        self.map_id = None
        self.map_xyz = self.map_label = None
        self.dw = (1.0, 1.0, 1.0)
        self.lik = dict(match_dist_min=0.2, match_dist_flat=0.05, match_weight=5.0)
        self.beam = dict(num_points=3, hit_range=0.3, add_penalty_short_only_mode=True, filter_label_max=0xFFFFFFFF)
        self.opt = dict(DEFAULTS)
        self.map_version = 0   # bumped whenever map_xyz is replaced (id() of a fresh array can repeat a freed one's)


INJECT = os.environ.get("MCL3DL_FUZZ_INJECT", "")   # a test-hook option switched on for every sequence (fault injection)


def reset_options(obj):
    for k, v in DEFAULTS.items():
        obj.set_option(k, v)
    if INJECT:
        obj.set_option(INJECT, 1)


def oracle_for(m, kind, cache):
    key = (m.map_id, m.map_version, m.dw, tuple(sorted(m.lik.items())), tuple(sorted(m.beam.items())))
    if cache.get("key") != key:
        o = cache.get("o") or pyoracle.Oracle(kind)
        o.set_map(m.map_xyz, m.map_label, dist_weight=m.dw)
        o.set_likelihood_params(pyoracle.LikelihoodParams(**m.lik))
        o.set_beam_params(pyoracle.BeamParams(**m.beam))
        cache["o"], cache["key"] = o, key
    return cache["o"]


def lik_is_exact(opt, n_s):
    """strict_order 1 / 3: always; the default (2): every scan of at most strict_exact_max points (round 6) — above it the sum may
    still be exact (per-particle rows), which the tolerance accepts as well."""
    mode = opt["strict_order"] if isinstance(opt, dict) else opt
    if mode in (1, 3):
        return True
    return isinstance(opt, dict) and mode == 2 and n_s <= opt["strict_exact_max"] and opt["strict_chunk"] == 0


def check_lik(opt, got_lik, got_q, want_lik, want_q, what, n_s=1 << 30):
    np.testing.assert_array_equal(got_q, want_q, err_msg=what + ": match ratio")
    if lik_is_exact(opt, n_s):
        np.testing.assert_array_equal(got_lik, want_lik, err_msg=what + ": likelihood (float order)")
    else:
        np.testing.assert_allclose(got_lik, want_lik, rtol=RTOL_FP64, err_msg=what + ": likelihood")


def run_sequence(seed, eng, grp, fresh, world, kind, log):
    import torch
    rng = np.random.default_rng(seed)
    m = Model()
    cache = {}
    pending = None     # a progressive batch begun and not yet ended: (arrays, want_fn)
    pinned = []
    reset_options(eng)
    eng.set_likelihood_params(**m.lik)   # (the session's engine carries the previous sequence's parameters)
    eng.set_beam_params(**m.beam)
    grp_map = None

    def pick_inputs(big=None, whole=False):
        # 70 % small shapes (one work-group per particle, the one-launch update, the small-scan kernel), 25 % medium ones with
        # >= 1024 points (the tiled kernel; with lik_tiled_min = 256 the small ones reach it too), 5 % several hundred
        # particles (more than one slice / pf block) — those are checked on a sample of their particles unless the caller
        # needs all of them (`whole`: pf::measure normalises over every particle)
        sc = world[m.map_id]
        u = rng.random()
        size = (2 if u < 0.05 else 1 if u < 0.30 else 0) if big is None else (1 if big else 0)
        if size == 2 and whole:
            size = 1
        n_p = int(rng.integers(300, 700)) if size == 2 else int(rng.integers(16, 128)) if size == 1 else int(rng.integers(1, 90))
        n_s = (int(rng.integers(1024, 2600)) if size == 2 else int(rng.integers(1024, 1600)) if size == 1 else
               int(rng.choice([0, 1, 7, 33, 96, 300, 777])))
        n_b = int(rng.choice([0, 0, 3, 17, 64]))
        p0 = int(rng.integers(0, 700 - n_p + 1))
        poses = np.ascontiguousarray(sc.poses[p0:p0 + n_p])
        scan = np.ascontiguousarray(sc.scan_lik[rng.permutation(2600)[:n_s]])
        beam = np.ascontiguousarray(sc.scan_beam[:n_b])
        lab = np.ascontiguousarray(sc.scan_beam_label[:n_b])
        return poses, scan, beam, lab, sc.origins

    def sample_of(n_p):
        return np.arange(n_p) if n_p <= 128 else np.sort(rng.choice(n_p, 48, replace=False))

    def want_models(poses, scan, beam, lab, origins, order=None):
        o = oracle_for(m, kind, cache)
        s = scan if order is None else np.ascontiguousarray(scan[order])
        if len(s):
            wl, wq = o.likelihood_measure(poses, s)
        else:
            wl, wq = np.ones(len(poses), np.float32), np.zeros(len(poses), np.float32)   # likelihood.cpp:111-114
        wb = o.beam_measure(poses, beam, lab, origins)[0] if len(beam) else np.ones(len(poses), np.float32)
        return wl, wq, wb

    def lik_order(obj, n_s):
        return obj.scan_order(n_s) if (m.opt["strict_order"] == 3 and n_s) else None

    def do_set_map():
        m.map_id = int(rng.integers(0, len(world)))
        sc = world[m.map_id]
        m.dw = [(1.0, 1.0, 1.0), (1.0, 1.0, 5.0), (1.0, 1.0, 2.0)][int(rng.integers(0, 3))]
        m.map_xyz, m.map_label = sc.map_xyz, sc.map_label
        m.map_version += 1
        eng.set_map(sc.map_xyz, sc.map_label, stamp=int(rng.integers(1, 1 << 30)), dist_weight=m.dw)
        log.append("set_map %d dw=%s" % (m.map_id, m.dw))

    def do_map_update():
        sc = world[m.map_id]
        half = sc.meta["n"] * 0.1 / 2.0
        n_new = int(rng.integers(0, 250))
        inside = rng.random() < 0.8
        span = half - 0.3 if inside else half + 1.0
        pts = rng.uniform(-span, span, (n_new, 3)).astype(np.float32)
        pts[:, 2] = rng.uniform(-half + 0.2, -half + 1.2, n_new)          # a band above the floor
        lab = rng.integers(0, 4, n_new).astype(np.uint32)
        n_map, st = eng.map_update(pts, lab, leaf=(0.2, 0.2, 0.2), stamp=int(rng.integers(1, 1 << 30)))
        m.map_xyz, m.map_label = eng.map_download()
        m.map_version += 1
        assert len(m.map_xyz) == n_map
        log.append("map_update %d pts (%s) -> %d, outcome %d" % (n_new, "inside" if inside else "beyond", n_map, st["outcome"]))
        # an updated engine == a fresh engine on the merged map, bit for bit, over the whole pool of poses and points
        # (both on the GPU: no oracle time) — under the options of the moment
        fresh.set_map(m.map_xyz, m.map_label, stamp=int(rng.integers(1, 1 << 30)), dist_weight=m.dw)
        fresh.set_likelihood_params(**m.lik)
        fresh.set_beam_params(**m.beam)
        # (every option of the moment: since round 6 kernel SELECTION decides the summation too — a per-particle kernel adds in the
        # caller's order where the tiled kernel's default is the fp64 tree)
        for k, v in m.opt.items():
            fresh.set_option(k, v)
        a = eng.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        b = fresh.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        np.testing.assert_array_equal(a[1], b[1], err_msg="updated engine vs fresh engine: match ratio")
        np.testing.assert_array_equal(a[2], b[2], err_msg="updated engine vs fresh engine: beam")
        if m.opt["strict_order"] in (1, 3):
            np.testing.assert_array_equal(a[0], b[0], err_msg="updated engine vs fresh engine: likelihood")
        else:
            np.testing.assert_allclose(a[0], b[0], rtol=2e-6, err_msg="updated engine vs fresh engine: likelihood")

    def do_params():
        if rng.random() < 0.5:
            m.lik = dict(match_dist_min=float(rng.choice([0.2, 0.15, 0.3])), match_dist_flat=float(rng.choice([0.05, 0.02])),
                         match_weight=float(rng.choice([5.0, 1.0])))
            eng.set_likelihood_params(**m.lik)
            log.append("set_likelihood_params %s" % m.lik)
        else:
            m.beam = dict(num_points=int(rng.choice([3, 8, 40])), hit_range=float(rng.choice([0.3, 0.2])),
                          add_penalty_short_only_mode=bool(rng.integers(0, 2)),
                          filter_label_max=int(rng.choice([0xFFFFFFFF, 1])))
            eng.set_beam_params(**m.beam)
            log.append("set_beam_params %s" % m.beam)

    def do_option():
        k = list(CHOICES)[int(rng.integers(0, len(CHOICES)))]
        v = CHOICES[k][int(rng.integers(0, len(CHOICES[k])))]
        eng.set_option(k, v)
        m.opt[k] = v
        log.append("set_option %s=%s" % (k, v))

    def do_measure_batch():
        poses, scan, beam, lab, org = pick_inputs()
        log.append("measure_batch %d x %d + %d" % (len(poses), len(scan), len(beam)))
        lik, q, b = eng.measure_batch(poses, scan, beam if len(beam) else None, lab if len(beam) else None,
                                      org if len(beam) else None)
        sel = sample_of(len(poses))
        wl, wq, wb = want_models(poses[sel], scan, beam, lab, org, lik_order(eng, len(scan)))
        check_lik(m.opt, lik[sel], q[sel], wl, wq, "measure_batch", len(scan))
        np.testing.assert_array_equal(b[sel], wb, err_msg="measure_batch: beam")

    def end_pending():
        nonlocal pending
        if pending is None:
            return
        (lik, q, b), (sel, wl, wq, wb), (opt_then, n_s_then) = pending
        eng.measure_batch_end()
        pending = None
        log.append("  _end of the batch begun earlier")
        check_lik(opt_then, lik[sel], q[sel], wl, wq, "progressive batch", n_s_then)
        np.testing.assert_array_equal(b[sel], wb, err_msg="progressive batch: beam")

    def do_progressive():
        nonlocal pending
        end_pending()
        poses, scan, beam, lab, org = pick_inputs()
        if rng.random() < 0.3:      # enough particles for several slices
            sc = world[m.map_id]
            poses = np.ascontiguousarray(sc.poses[:int(rng.integers(300, 700))])
        sl = int(rng.choice([0, 16, 64, 200]))
        log.append("measure_batch_begin %d x %d + %d slice %d" % (len(poses), len(scan), len(beam), sl))
        arrays = eng.measure_batch_begin(poses, scan, beam if len(beam) else None, lab if len(beam) else None,
                                         org if len(beam) else None, slice_particles=sl)
        order = lik_order(eng, len(scan))
        for _ in range(int(rng.integers(0, 3))):
            i = int(rng.integers(0, len(poses)))
            ready = eng.measure_batch_wait(i)
            assert ready > i, "wait(%d) reported %d" % (i, ready)
        # what the batch must deliver is fixed by the state at _begin (everything is enqueued there): later map / parameter
        # calls must not change it
        sel = sample_of(len(poses))
        pending = (arrays, (sel,) + want_models(poses[sel], scan, beam, lab, org, order), (dict(m.opt), len(scan)))
        if rng.random() < 0.5:
            end_pending()     # otherwise: abandoned until a later call ends it

    def do_measure_update():
        poses, scan, beam, lab, org = pick_inputs(whole=True)
        n_p = len(poses)
        w0 = rng.uniform(0.5, 1.5, n_p).astype(np.float32)
        w0 /= w0.sum()
        use_pinned = rng.random() < 0.4
        log.append("measure_update %d x %d + %d %s" % (n_p, len(scan), len(beam), "page-locked" if use_pinned else "pageable"))
        o = oracle_for(m, kind, cache)
        if use_pinned:
            hp, hw = eng.host_array((n_p, 7)), eng.host_array(n_p)
            hl, hq, hb = eng.host_array(n_p), eng.host_array(n_p), eng.host_array(n_p)
            pinned.extend([hp, hw, hl, hq, hb])
            hp[:], hw[:] = poses, w0
            ent, rmin, rmax, rest = eng.measure_update_into(hp, hw, scan if len(scan) else None, beam if len(beam) else None,
                                                            lab if len(beam) else None, org if len(beam) else None, hl, hq, hb,
                                                            extra=np.full(n_p, ND0, np.float32))
            got = dict(weights=hw.copy(), lik=hl.copy(), quality=hq.copy(), beam=hb.copy(), entropy=ent, restored=rest)
        else:
            got = eng.measure_update(poses, w0, scan, beam if len(beam) else None, lab if len(beam) else None,
                                     org if len(beam) else None, extra=np.full(n_p, ND0, np.float32))
        order = lik_order(eng, len(scan))
        s = scan if order is None else np.ascontiguousarray(scan[order])
        want = o.measure_update(poses, w0, s, beam, lab, org)
        check_lik(m.opt, got["lik"], got["quality"], want["lik"], want["quality"], "measure_update", len(scan))
        np.testing.assert_array_equal(got["beam"], want["beam"], err_msg="measure_update: beam")
        assert bool(got["restored"]) == bool(want["restored"])
        # (the default adds the weights in the reference's float order too, up to 1024 particles: host_measure.h: pf_float_order)
        if m.opt["strict_order"] == 1 or (m.opt["strict_order"] == 2 and lik_is_exact(m.opt, len(scan)) and n_p <= 1024):
            np.testing.assert_array_equal(got["weights"], want["weights"], err_msg="measure_update: weights (float order)")
        else:
            np.testing.assert_allclose(got["weights"], want["weights"], rtol=1e-4, atol=1e-12, err_msg="measure_update: weights")
        if not want["restored"]:
            np.testing.assert_allclose(got["entropy"], want["entropy"], rtol=1e-4, atol=1e-5)

    def do_large_update():
        """Round 6: an update of more than 1024 particles — the split pf::measure with its fusions (the tiled kernel's per-tile
        partials and the beam model's counts taken by the first pf kernel, the reduction inside pf_apply), both models in one
        launch where the options of the moment allow it, the beam kernel's counters left zeroed for the next call. The models are
        checked on a sample of the particles, pf::measure on the engine's own factors."""
        sc = world[m.map_id]
        n_p = int(rng.integers(1030, 2300))
        n_s = int(rng.choice([300, 1100, 2048, 2600]))
        n_b = int(rng.choice([0, 3, 17, 64]))
        poses = np.ascontiguousarray(sc.poses[:n_p])
        scan = np.ascontiguousarray(sc.scan_lik[rng.permutation(2600)[:n_s]])
        beam, lab, org = np.ascontiguousarray(sc.scan_beam[:n_b]), np.ascontiguousarray(sc.scan_beam_label[:n_b]), sc.origins
        w0 = rng.uniform(0.5, 1.5, n_p).astype(np.float32)
        extra = rng.uniform(0.2, 0.4, n_p).astype(np.float32)
        log.append("large measure_update %d x %d + %d" % (n_p, n_s, n_b))
        got = eng.measure_update(poses, w0, scan, beam if n_b else None, lab if n_b else None, org if n_b else None, extra=extra)
        order = lik_order(eng, n_s)
        sel = sample_of(n_p)
        wl, wq, wb = want_models(poses[sel], scan, beam, lab, org, order)
        # (from 2048 particles the exact sum of a short scan is the term array's, which strict_auto_max_bytes may rule out: the
        # engine says which sum the launch took — lik_exact — and is held to it)
        exact = bool(eng.get_option("lik_exact")) if n_s else True
        if m.opt["strict_order"] == 3:
            assert exact, "the in-kernel chain is the reference's sum at every size"
        check_lik(3 if exact else 0, got["lik"][sel], got["quality"][sel], wl, wq, "large update", n_s)
        np.testing.assert_array_equal(got["beam"][sel], wb, err_msg="large update: beam")
        wn = w0 * (((np.float32(1.0) * got["beam"]) * got["lik"]) * extra)
        total = wn.sum(dtype=np.float64)
        if total > 0:
            np.testing.assert_allclose(got["weights"], wn / np.float32(total), rtol=2e-5, err_msg="large update: weights")
            assert not got["restored"]
        else:
            np.testing.assert_array_equal(got["weights"], w0)

    def do_scan_prep():
        sc = world[m.map_id]
        n_raw = int(rng.integers(200, 3000))
        raw = np.ascontiguousarray(sc.scan_lik[rng.permutation(2600)[:min(n_raw, 2600)]])
        raw = raw + rng.normal(0, 0.02, raw.shape).astype(np.float32)
        n_full, n_lc, n_bc = eng.scan_begin(raw, None, leaf=(0.1, 0.1, 0.05), clip_lik=(0.5, 4.0, -2.0, 2.0),
                                            clip_beam=(0.5, 3.0, -2.0, 2.0))
        if n_lc == 0:
            log.append("scan_begin %d raw -> nothing left" % n_raw)
            return
        n_s = int(rng.choice([8, 96, 500, 1500]))
        n_b = int(rng.choice([0, 3, 40])) if n_bc else 0
        idx_l = rng.integers(0, n_lc, n_s).astype(np.uint32)
        idx_b = rng.integers(0, n_bc, n_b).astype(np.uint32) if n_b else None
        eng.scan_finish(idx_l, idx_b, sc.origins)
        log.append("scan_begin %d raw -> %d / %d / %d; scan_finish %d + %d" % (n_raw, n_full, n_lc, n_bc, n_s, n_b))
        samp_l, _ = eng.scan_download(3)
        samp_b, lab_b = eng.scan_download(4) if n_b else (np.zeros((0, 3), np.float32), np.zeros(0, np.uint32))
        n_p = int(rng.integers(1, 128))
        poses = np.ascontiguousarray(sc.poses[:n_p])
        dev = torch.device("cuda", 0)
        d_pose = torch.from_numpy(poses).to(dev)
        d_l, d_q, d_b = (torch.zeros(n_p, device=dev) for _ in range(3))
        torch.cuda.synchronize()
        eng.measure_device(d_pose, n_p, d_l, d_q, d_b)
        eng.synchronize()
        order = lik_order(eng, n_s)
        wl, wq, wb = want_models(poses, samp_l, samp_b, lab_b, sc.origins, order)
        check_lik(m.opt, d_l.cpu().numpy(), d_q.cpu().numpy(), wl, wq, "scan_finish + measure_device", n_s)
        np.testing.assert_array_equal(d_b.cpu().numpy(), wb, err_msg="scan_finish + measure_device: beam")

    def do_resample():
        n = int(rng.integers(2, 600))
        w = rng.uniform(0.0, 1.0, n).astype(np.float32)
        w[rng.random(n) < 0.1] = 0.0
        w /= max(float(w.sum()), 1e-30)
        n_out = n if rng.random() < 0.7 else int(rng.integers(1, 900))
        mode = int(rng.integers(0, 2))
        port = cache.get("port") or pyoracle.Oracle("port")
        cache["port"] = port
        pstep = eng.resample_begin(w, n_out)
        ip = float(rng.uniform(0, 1)) * pstep
        src, dup, n_dup = eng.resample_plan(mode, ip)
        wsrc, wdup = port.resample_plan(w, n_out, mode, ip)
        log.append("resample %d -> %d mode %d" % (n, n_out, mode))
        np.testing.assert_array_equal(src, wsrc, err_msg="resample plan: sources")
        np.testing.assert_array_equal(dup, wdup, err_msg="resample plan: duplicates")
        assert n_dup == int(wdup.sum())

    def do_expectation():
        sc = world[m.map_id]
        n = int(rng.integers(1, 500))
        w = rng.uniform(0.1, 1.0, n).astype(np.float32)
        w /= w.sum()
        mean, total, im, ib = eng.expectation(sc.poses[:n], w)
        o = oracle_for(m, kind, cache)
        wm, wim, _ = o.expectation(sc.poses[:n], w)
        log.append("expectation %d" % n)
        np.testing.assert_allclose(mean[:3], wm[:3], rtol=1e-5, atol=1e-5)
        assert im == wim

    def do_group():
        nonlocal grp_map
        sc = world[m.map_id]
        if grp_map != (m.map_id, m.map_version, m.dw):
            grp.set_map(m.map_xyz, m.map_label, stamp=int(rng.integers(1, 1 << 30)), dist_weight=m.dw)
            grp_map = (m.map_id, m.map_version, m.dw)
        grp.set_likelihood_params(**m.lik)
        grp.set_beam_params(**m.beam)
        for k in ("poll_sync", "update_stage", "update_zero_copy", "lik_coop") + SUM_OPTIONS:
            grp.set_option(k, m.opt[k])
        poses, scan, beam, lab, org = pick_inputs(whole=True)
        if len(poses) < 2:
            return
        n_p = len(poses)
        w0 = np.full(n_p, 1.0 / n_p, np.float32)
        log.append("group measure_update %d x %d + %d" % (n_p, len(scan), len(beam)))
        got = grp.measure_update(poses, w0, scan, beam if len(beam) else None, lab if len(beam) else None,
                                 org if len(beam) else None, extra=np.full(n_p, ND0, np.float32))
        o = oracle_for(m, kind, cache)
        order = None
        if m.opt["strict_order"] == 3 and len(scan) and not m.opt["scan_presorted"]:
            order = capi.scan_order_host(scan)   # (every rank of the group orders the same scan the same way)
        s = scan if order is None else np.ascontiguousarray(scan[order])
        want = o.measure_update(poses, w0, s, beam, lab, org)
        check_lik(m.opt, got["lik"], got["quality"], want["lik"], want["quality"], "group measure_update", len(scan))
        np.testing.assert_array_equal(got["beam"], want["beam"], err_msg="group: beam")
        np.testing.assert_allclose(got["weights"], want["weights"], rtol=1e-4, atol=1e-12, err_msg="group: weights")

    ops = [(do_set_map, 2), (do_map_update, 2), (do_params, 3), (do_option, 8), (do_measure_batch, 6), (do_progressive, 5),
           (do_measure_update, 6), (do_large_update, 2), (do_scan_prep, 3), (do_resample, 2), (do_expectation, 1), (do_group, 2)]
    fns = [f for f, _ in ops]
    p = np.array([w for _, w in ops], np.float64)
    p /= p.sum()
    try:
        do_set_map()
        for _ in range(N_CALLS):
            fns[int(rng.choice(len(fns), p=p))]()
        end_pending()
    finally:
        try:
            if pending is not None:
                eng.measure_batch_end()
        finally:
            for a in pinned:
                eng.host_free(a)
            reset_options(eng)


@pytest.fixture(scope="module")
def group2():
    g = capi.Group([0, 0], collective="host")
    yield g
    g.close()


@pytest.fixture(scope="module")
def fresh_engine():
    e = capi.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("chunk", range((N_SEQ + CHUNK - 1) // CHUNK))
def test_random_call_sequences_against_the_reference(engine, group2, fresh_engine, world, oracle_kind, chunk):
    for seed in range(chunk * CHUNK, min(N_SEQ, (chunk + 1) * CHUNK)):
        log = []
        try:
            run_sequence(9000 + seed, engine, group2, fresh_engine, world, oracle_kind, log)
        except Exception:
            print("\nsequence seed %d failed after:\n  " % (9000 + seed) + "\n  ".join(log))
            raise


def test_the_fuzz_finds_the_round_4_staging_hazard_again():
    """'the r04ad hazard reproduces on the pre-fix commit'. The fix (three lines: build lazily-built
    structures BEFORE the staged update allocates its staging memory, api_core.inl:measure_update_staged) and the mode that
    exposed it (poll_sync = 2) arrived in one commit, so the pre-fix state is re-created by a test hook that takes those three
    lines out again — and the same fuzz, run in a child process with the hook on, must FAIL."""
    import subprocess
    import sys
    env = dict(os.environ, MCL3DL_HIP_TEST_HOOKS="1", MCL3DL_FUZZ_INJECT="test_late_structures", MCL3DL_FUZZ_SEQUENCES="120")
    proc = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "random_call_sequences",
                           "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, env=env,
                          cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    tail = proc.stdout[-3000:]
    assert proc.returncode != 0 and "sequence seed" in proc.stdout, "the injected hazard went unnoticed:\n" + tail
