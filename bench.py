#!/usr/bin/env python3
"""bench.py — one measurement update per step on BASELINE.json's headline configuration.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic input: likelihood-field kernel over
N_p particles x N_s scan points (+ beam kernel when the workload has beam points), pf::measure
(weight multiply, {sum w, sum w ln w, ratio min/max} reduction, one all-reduce when N > 1, normalise +
entropy).  Inputs (map structures, ordered scan, poses, prior weights) are resident in HBM before the
timed region starts.  Particles shard across ranks ("weak": every GPU gets the configuration's full
particle count); map and scan are replicated.

Prints ONE JSON line on rank 0 with the driver's contract keys plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", help="C1..C5 of BASELINE.json (default C2: 4096 x 16k, 1M-pt map)")
    ap.add_argument("--particles", type=int, default=0, help="override particles per GPU")
    ap.add_argument("--dist-weight-z", type=float, default=1.0)
    ap.add_argument("--lik-index", type=int, default=2,
                    help="2 = candidate records (default), 1 = candidate runs, 0 = 27-cell scan")
    ap.add_argument("--cand-voxel-ratio", type=float, default=0.5)
    ap.add_argument("--cand-phase", type=float, default=0.5)
    ap.add_argument("--lik-tiled", type=int, default=1)
    ap.add_argument("--beam-points", type=int, default=0, help="override the beam scan size N_b")
    ap.add_argument("--map-jitter", type=float, default=0.0,
                    help="displace every map point uniformly by +-this (m): voxel-filter centroids instead of a lattice")
    ap.add_argument("--lik-small", type=int, default=1)
    ap.add_argument("--overlap-models", type=int, default=1)
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="wall time of untimed updates before the W warm-up steps (GPU clock ramp), 0 = none")
    ap.add_argument("--timing-mask", type=int, default=1,
                    help="kernel groups timed with hipEvents INSIDE the timed region (bit 0 likelihood = the roofline "
                         "kernel, 1 beam, 2 pf); the others are timed in a second pass of the same steps")
    ap.add_argument("--scan-points", type=int, default=0, help="override the number of likelihood scan points")
    ap.add_argument("--lik-group", type=int, default=0, help="0 = the library's choice (16 / 8 / 4 by launch size)")
    ap.add_argument("--strict-order", type=int, default=0,
                    help="1 = reference float summation order (bit-identical likelihoods and weights; slower)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the all-reduce even with one rank (exercises the RCCL path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-particles", type=int, default=0,
                    help="particles in the CPU-baseline sample (0 = as many as ~12 s of one core buys, at most all)")
    return ap.parse_args()


def pmc_traffic(kernel_prefix, workload):
    """HBM-side bytes per launch of the dominant kernel from the newest committed PMC summary for this workload
    (profiles/*_pmc_summary.csv, produced by profiles/run_profiles.sh in separate --pmc passes): FETCH_SIZE and
    WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes actually fetched (MI355X_MICROARCH.md, HBM
    section), hence the factor 2."""
    import csv
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s*_pmc_summary.csv" % workload))):
        vals = {}
        for row in csv.DictReader(open(path)):
            if row["kernel"].startswith(kernel_prefix):
                vals[row["counter"]] = float(row["mean_per_launch"])
        if "FETCH_SIZE" in vals:
            best = (path, vals)
    if not best:
        return None, None
    path, vals = best
    return (2.0 * vals["FETCH_SIZE"] + vals.get("WRITE_SIZE", 0.0)) * 1024.0, os.path.relpath(path, ROOT)


def cpu_baseline(sc, dist_weight, n_particles, beam_points):
    """The reference's own measure() loop on this box's host cores (oracle/_ref when built, the C port otherwise),
    single thread = the reference's execution model (src/mcl_3dl.cpp:1466), on a bounded sample of the same workload."""
    from oracle import pyoracle
    kind = "ref" if pyoracle.available("ref") else "port"
    o = pyoracle.Oracle(kind)
    o.set_map(sc.map_xyz, sc.map_label, dist_weight=dist_weight)
    o.set_likelihood_params(pyoracle.LikelihoodParams())
    o.set_beam_params(pyoracle.BeamParams(num_points=max(beam_points, 1)))
    if n_particles <= 0:
        # ~12 s of single-thread work at the ~4.4e6 evals/s this path runs at on one core
        n_particles = max(64, int(12.0 * 4.4e6 / max(len(sc.scan_lik), 1)))
    n = min(n_particles, len(sc.poses))
    lik, q, sec = o.likelihood_measure(sc.poses[:n], sc.scan_lik, threads=1, return_time=True)
    evals = n * len(sc.scan_lik)
    out = {"value": evals / sec, "unit": "particle\u00b7point evals/s", "cores": 1,
           "kind": "reference" if kind == "ref" else "port",
           "sample": "%d of the workload's particles x %d points, likelihood model, 1 thread, %.1f s"
                     % (n, len(sc.scan_lik), sec),
           "note": "nearest-neighbour index behind pcl::KdTreeFLANN is this repo's stand-in (PCL/FLANN not installed)"}
    threads = o.max_threads()
    if threads > 1:
        _, _, sec_mt = o.likelihood_measure(sc.poses[:n], sc.scan_lik, threads=threads, return_time=True)
        out["all_cores"] = {"value": evals / sec_mt, "cores": threads}
    return out, lik, q


def _tiled_group(n_s, n_p, forced):
    """The tiled kernel's particles-per-work-group as the library picks it (host_measure.h)."""
    if forced:
        return forced
    n_tiles = (n_s + 255) // 256
    for g in (16, 8, 4):
        if n_tiles * ((n_p + g - 1) // g) >= 2048:
            return g
    return 4


def _identity_noise(n):
    """n noise states that leave a particle where it is (zero offsets, identity quaternion)."""
    a = np.zeros((n, 13), np.float32)
    a[:, 6] = 1.0
    return a


def _flush_c_stdio():
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from mcl_3dl_amd import capi
    from mcl_3dl_amd.synthetic import CONFIGS, make_config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
        # RCCL writes a version banner through C stdio when its communicator comes up; push it out NOW so that rank 0's
        # JSON line is the last thing on stdout
        warm = torch.zeros(1, device=dev)
        dist.all_reduce(warm)
        torch.cuda.synchronize(dev)
        _flush_c_stdio()

    cfg = CONFIGS[args.workload]
    n_p = args.particles or cfg["n_p"]
    # weak scaling: every rank holds a full-size particle shard drawn with its own seed; map and scan are replicated
    extra_cfg = dict(n_s=args.scan_points) if args.scan_points else {}
    if args.map_jitter:
        extra_cfg["map_jitter"] = args.map_jitter
    if args.beam_points:
        extra_cfg["n_b"] = args.beam_points  # C3 stress case N_b = 16 384
    sc = make_config(args.workload, n_p=n_p, seed=12345, **extra_cfg)
    if rank > 0:
        shard = make_config(args.workload, n_p=n_p, seed=12345 + rank, **extra_cfg)
        sc.poses = shard.poses
    dist_weight = (1.0, 1.0, args.dist_weight_z)
    n_s, n_b = len(sc.scan_lik), len(sc.scan_beam)

    eng = capi.Engine(local_rank)
    # One explicit (non-null) stream carries everything in order: torch copies, the engine's kernels, the RCCL all-reduce.
    # (torch's default stream has handle 0, which mcl3dl_hip_set_stream reads as "use the context's own stream".)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    eng.set_stream(stream.cuda_stream)
    t0 = time.time()
    eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=dist_weight)
    eng.set_likelihood_params()
    eng.set_option("lik_index", args.lik_index)
    eng.set_option("cand_voxel_ratio", args.cand_voxel_ratio)
    eng.set_option("cand_phase", args.cand_phase)
    eng.set_option("strict_order", args.strict_order)
    eng.set_option("lik_tiled", args.lik_tiled)
    eng.set_option("lik_small", args.lik_small)
    eng.set_option("overlap_models", args.overlap_models)
    eng.set_option("lik_group", args.lik_group)
    eng.set_beam_params(num_points=max(n_b, 1), dda_grid_size=0.2)
    eng.upload_scan(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)

    d_pose = torch.from_numpy(sc.poses).to(dev).contiguous()
    d_w0 = torch.from_numpy(sc.weights).to(dev).contiguous()
    d_w = d_w0.clone()
    d_lik = torch.empty(n_p, dtype=torch.float32, device=dev)
    d_ratio = torch.empty(n_p, dtype=torch.float32, device=dev)
    d_beam = torch.empty(n_p, dtype=torch.float32, device=dev)
    d_stats = torch.zeros(4, dtype=torch.float32, device=dev)
    # one all-reduce(SUM) carries the sums and, in per-rank slots, the max/min candidates
    d_pack = torch.zeros(2 + 2 * world, dtype=torch.float64, device=dev)

    def step():
        d_w.copy_(d_w0)  # resampling leaves uniform weights before every update (pf.h:203,207)
        eng.measure_device(d_pose, n_p, d_lik, d_ratio, d_beam if n_b else None)
        eng.pf_partial_device(d_w, d_lik, d_beam if n_b else None, None, d_ratio, n_p, d_pack, rank, world)
        if use_dist:
            dist.all_reduce(d_pack, op=dist.ReduceOp.SUM)  # the update's single collective (RCCL over xGMI), 16+16*N bytes
        eng.pf_apply_device(d_w, n_p, d_pack, d_stats, world)

    # first call builds + uploads the map structures (outside every timed region)
    step()
    torch.cuda.synchronize(dev)
    setup_s = time.time() - t0

    # exact workload counts for the algorithmic-bytes accounting (counting kernels, not timed)
    ws = eng.workload_stats(d_pose, n_p)
    k_bar = ws["sum_k"] / max(ws["evals"], 1.0)
    # B_lik = 16 (scan point) + 27*4 (cell-range entries) + 16*K (candidate points) per evaluation,
    # + 28 B pose + 8 B result per particle
    bytes_lik_launch = ws["evals"] * (16.0 + 27 * 4.0 + 16.0 * k_bar) + n_p * 36.0
    bytes_beam_launch = 0.0
    if n_b:
        # B_beam per ray = 16 + S*1 + O*8 + T*16
        bytes_beam_launch = ws["rays"] * 16.0 + ws["dda_steps"] + ws["dda_occupied"] * 8.0 + ws["dda_tested"] * 16.0

    # Clock ramp: a GPU coming out of idle needs tens of milliseconds of load before it holds its sustained clocks, far
    # more than W steps of a sub-millisecond update. Run the update for --prewarm-ms of wall time first (set-up, like the
    # map upload above; not part of W or K), then the W warm-up steps the contract asks for.
    prewarm = {"ms": 0.0, "batches": 0}
    if args.prewarm_ms > 0:
        torch.cuda.synchronize(dev)
        t_pre = time.perf_counter()
        for _ in range(8):
            step()
        torch.cuda.synchronize(dev)
        per_step_ms = (time.perf_counter() - t_pre) * 1e3 / 8
        # batches of ~prewarm_ms / 4; at least 4 of them, then until two consecutive batches agree within 1 % (the clock
        # has settled) or 5 x prewarm_ms have gone by. Every rank runs the same number of batches (the decision is
        # all-reduced), so the collectives inside step() stay matched.
        n_batch = torch.tensor([max(1, min(int(args.prewarm_ms / 4 / max(per_step_ms, 1e-3)), 25000))], dtype=torch.int64,
                               device=dev)
        if use_dist:
            dist.all_reduce(n_batch, op=dist.ReduceOp.MAX)
        n_batch = int(n_batch.item())
        prev, stable = None, 0
        t_all = time.perf_counter()
        while True:
            tb = time.perf_counter()
            for _ in range(n_batch):
                step()
            torch.cuda.synchronize(dev)
            cur = (time.perf_counter() - tb) / n_batch
            prewarm["batches"] += 1
            stable = stable + 1 if (prev is not None and abs(cur - prev) <= 0.01 * prev) else 0
            prev = cur
            elapsed_ms = (time.perf_counter() - t_all) * 1e3
            more = torch.tensor([1 if (prewarm["batches"] < 4 or (stable < 2 and elapsed_ms < 5 * args.prewarm_ms)) else 0],
                                dtype=torch.int64, device=dev)
            if use_dist:
                dist.all_reduce(more, op=dist.ReduceOp.MAX)
            if int(more.item()) == 0:
                break
        prewarm["ms"] = (time.perf_counter() - t_pre) * 1e3
    for _ in range(args.warmup):
        step()
    eng.set_option("timing_mask", args.timing_mask)
    eng.set_kernel_timing(True)
    eng.reset_kernel_time()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t1
    lik_ms, lik_n = eng.kernel_time(capi.KERNEL_LIKELIHOOD)
    # Second pass of the same K steps, outside `elapsed`, every kernel group timed and the two models one after the other:
    # the beam / pf durations (each timed group costs two event records per launch, so only the roofline kernel is
    # timed inside the timed region), and the likelihood duration too when it ran concurrently with the beam kernels
    # above (overlapping event intervals say nothing about either kernel).
    overlapped = bool(n_b and n_s and args.overlap_models)
    eng.set_option("timing_mask", 7)
    eng.set_option("overlap_models", 0)
    eng.reset_kernel_time()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    lik2_ms, lik2_n = eng.kernel_time(capi.KERNEL_LIKELIHOOD)
    beam_ms, beam_n = eng.kernel_time(capi.KERNEL_BEAM)
    pf_ms, pf_n = eng.kernel_time(capi.KERNEL_PF)
    eng.set_option("overlap_models", args.overlap_models)
    if overlapped or not (args.timing_mask & 1):
        lik_ms, lik_n = lik2_ms, lik2_n
        kernel_timing_pass = "second pass of the same steps (overlap_models=0, all kernel groups timed)"
    else:
        kernel_timing_pass = "likelihood: hipEvents inside the timed region; beam, pf: second pass of the same steps"
    eng.set_kernel_timing(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        evals_per_step = float(world) * n_p * n_s
        ms_per_step = elapsed / args.steps * 1e3
        value = evals_per_step * args.steps / elapsed
        lik_avg_ms = lik_ms / max(lik_n, 1)
        achieved = bytes_lik_launch / (lik_avg_ms * 1e-3) / 1e9 if lik_n else 0.0
        stats = d_stats.cpu().numpy()
        tiled = bool(args.lik_tiled and n_s >= 1024 and n_p >= 4)
        group = _tiled_group(n_s, n_p, args.lik_group)
        traffic, traffic_src = pmc_traffic("void mcl3dl::likelihood_tiled_kernel<%d, %d>" % (group, args.lik_index)
                                           if tiled else "void mcl3dl::likelihood_kernel<256, %d, false>" % args.lik_index,
                                           args.workload)
        # bytes the shipped index really reads per evaluation, priced against the measured L2 ceiling
        # (MI355X_MICROARCH.md: ~34.5 TB/s aggregate): mode 2 = brick-table entry + one 64-byte voxel record
        l2 = None
        if lik_n and args.lik_index in (0, 2):
            bpe = 68.0 if args.lik_index == 2 else bytes_lik_launch / max(ws["evals"], 1.0)
            l2_gbps = ws["evals"] * bpe / (lik_avg_ms * 1e-3) / 1e9
            l2 = {"bytes_per_eval": bpe, "achieved": l2_gbps, "peak": 34500.0, "unit": "GB/s", "frac": l2_gbps / 34500.0}
        out = {
            "metric": "particle\u00b7point evals/sec; filter-update Hz @ 4096 particles \u00d7 16k-pt scan",  # BASELINE.json
            "value": value,
            "unit": "particle\u00b7point evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d particles/GPU x %d-pt scan, %d-pt cube map, likelihood model%s, dist_weight=(1,1,%g)"
                            % (args.workload, n_p, n_s, len(sc.map_xyz), " + beam (DDA) %d rays/particle" % n_b if n_b else "",
                               args.dist_weight_z),
                "particles_per_gpu": n_p, "scan_points": n_s, "beam_points": n_b, "map_points": int(len(sc.map_xyz)),
                "parallelism": "particles sharded x%d, map+scan replicated, 1 all-reduce/update" % world,
                "update_hz": 1e3 / ms_per_step,
                "accumulate": ("float, reference order (bit-identical results)" if args.strict_order else
                               "fp64 tree (terms bit-identical to the reference's float terms)"),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": (("likelihood_tiled_kernel<%d,%d>" % (group, args.lik_index)) if tiled else
                           ("likelihood_small_kernel<%d>" % args.lik_index) if (n_s <= 32 and n_p >= 256 and args.lik_small) else
                           ("likelihood_kernel<%d,%d>" % (64 if n_s <= 128 else 256, args.lik_index))),
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": bytes_lik_launch,
                "bytes_per_eval": bytes_lik_launch / max(ws["evals"], 1.0),
                "k_bar": k_bar,
                "avg_launch_ms": lik_avg_ms,
                "launches": lik_n,
                # what the shipped index really reads per evaluation (brick-table entry + one 64-byte voxel record;
                # the 27-cell scan reads the canonical bytes above), priced against the measured L2 ceiling
                # (MI355X_MICROARCH.md: ~34.5 TB/s aggregate)
                "l2": l2,
            },
            "prewarm": prewarm,
            "kernel_timing_pass": kernel_timing_pass,
            "kernels_ms_per_step": {"likelihood": lik_avg_ms, "beam": beam_ms / max(beam_n, 1) if n_b else 0.0,
                                    "pf": 2.0 * pf_ms / max(pf_n, 1)},
            "setup_seconds": setup_s,
            "index": dict(eng.index_stats(), lik_index=args.lik_index, voxel_ratio=args.cand_voxel_ratio,
                          footprint_bytes=eng.memory_footprint()),
            "result_check": {"entropy": float(stats[0]), "match_ratio_min": float(stats[1]),
                             "match_ratio_max": float(stats[2]), "restored": bool(stats[3])},
        }
        # self-consistency of the numbers that were timed: normalised weights sum to 1 and reproduce the entropy
        wf = d_w.cpu().numpy().astype(np.float64)
        out["result_check"]["weight_sum"] = float(wf.sum())
        out["result_check"]["entropy_from_weights"] = float(-(wf[wf > 0] * np.log(wf[wf > 0])).sum())
        if n_b:
            beam_avg = beam_ms / max(beam_n, 1)
            out["beam"] = {"rays_per_s": ws["rays"] / (beam_avg * 1e-3), "dda_steps_per_s": ws["dda_steps"] / (beam_avg * 1e-3),
                           "algorithmic_GBps": bytes_beam_launch / (beam_avg * 1e-3) / 1e9, "avg_launch_ms": beam_avg}
        # the reductions that follow the update in the node (expectationBiased + max + covariance) on the
        # device-resident particles; each call ends with a D2H of a dozen scalars
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        for _ in range(10):
            mean7, _tot, _im, _ib = eng.expectation_device(d_pose, d_w, None, n_p)
            eng.covariance_device(d_pose, d_w, n_p, mean7)
        out["post_update_reductions"] = {"ms": (time.perf_counter() - t3) / 10 * 1e3,
                                         "what": "expectationBiased + max + covariance over this rank's particles"}
        # resampling of this rank's particles with the weights the update just produced: host bookkeeping
        # (prefix sums, tie sort, it/it_prev walk) + device lower_bound / gather; noise = identity (timing only)
        w_host = d_w.cpu().numpy()
        st13 = np.zeros((n_p, 13), np.float32)
        st13[:, :7] = sc.poses
        t4 = time.perf_counter()
        for _ in range(5):
            pstep = eng.resample_begin(w_host)
            _src, _dup, n_dup = eng.resample_plan(0, 0.37 * pstep)
            ident = np.zeros((n_dup, 13), np.float32)
            ident[:, 6] = 1.0
            eng.resample_apply(st13, ident)
        out["resample"] = {"ms": (time.perf_counter() - t4) / 5 * 1e3, "duplicates": int(n_dup),
                           "what": "mcl3dl_hip_resample_begin + plan + apply, host buffers, %d particles" % n_p}
        # the same with weights and 13-float states resident on the device (what a multi-GPU host runs per rank after
        # the all-gather, mcl_3dl_amd/distributed.py:sharded_resample)
        d_st_in = torch.from_numpy(st13).to(dev)
        d_st_out = torch.empty_like(d_st_in)
        torch.cuda.synchronize(dev)
        t6 = time.perf_counter()
        for _ in range(5):
            pstep = eng.resample_begin_device(d_w, n_p)
            _src, _dup, n_dup2 = eng.resample_plan(0, 0.37 * pstep, want_plan=False)
            eng.resample_apply_device(d_st_in, ident[:n_dup2], d_st_out)
        out["resample"]["ms_device_resident"] = (time.perf_counter() - t6) / 5 * 1e3
        if world == 1:
            # the drop-in boundary hands over HOST buffers: time the synchronous host entry point too (scan ordering on
            # the host, H2D of scan + poses + weights, kernels, D2H of weights) — never part of `value`
            eng.set_stream(None)
            eng.measure_update(sc.poses, sc.weights, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
            t2 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                eng.measure_update(sc.poses, sc.weights, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
            host_ms = (time.perf_counter() - t2) / reps * 1e3
            out["host_api"] = {"ms_per_update": host_ms, "evals_per_s": n_p * n_s / (host_ms * 1e-3),
                               "note": "mcl3dl_hip_measure_update with host buffers (PCIe + host-side scan ordering included)"}
            eng.set_stream(stream.cuda_stream)
            # the fused device-resident call (measure + pf::measure in one C call) replaying its captured hipGraph, next
            # to the same call enqueuing kernel by kernel: what launch overhead is worth at this size. Not `value`.
            fused = {}
            for use_graph in (0, 1):
                eng.set_option("use_graph", use_graph)
                for _ in range(3):
                    d_w.copy_(d_w0)
                    eng.update_device(d_pose, n_p, d_w, d_stats, d_lik=d_lik, d_ratio=d_ratio, d_beam=d_beam if n_b else None)
                torch.cuda.synchronize(dev)
                t5 = time.perf_counter()
                for _ in range(args.steps):
                    d_w.copy_(d_w0)
                    eng.update_device(d_pose, n_p, d_w, d_stats, d_lik=d_lik, d_ratio=d_ratio, d_beam=d_beam if n_b else None)
                torch.cuda.synchronize(dev)
                fused["graph" if use_graph else "eager"] = (time.perf_counter() - t5) / args.steps * 1e3
            # one whole filter iteration with everything resident on the device: measurement update -> expectation + max
            # (the node publishes the pose from it) -> resampling of the 13-float states; noise = identity
            eng.set_option("use_graph", 0)
            torch.cuda.synchronize(dev)
            t7 = time.perf_counter()
            for _ in range(args.steps):
                d_w.copy_(d_w0)
                eng.update_device(d_pose, n_p, d_w, d_stats, d_lik=d_lik, d_ratio=d_ratio, d_beam=d_beam if n_b else None)
                eng.expectation_device(d_pose, d_w, None, n_p)
                pstep = eng.resample_begin_device(d_w, n_p)
                _s, _d, nd = eng.resample_plan(0, 0.37 * pstep, want_plan=False)
                eng.resample_apply_device(d_st_in, _identity_noise(nd), d_st_out)
            torch.cuda.synchronize(dev)
            out["filter_iteration"] = {"ms": (time.perf_counter() - t7) / args.steps * 1e3,
                                       "what": "update + expectationBiased/max + resample, device-resident, one GPU"}
            out["fused_update"] = {"ms_per_update_graph": fused["graph"], "ms_per_update_eager": fused["eager"],
                                   "graph": eng.graph_stats(),
                                   "what": "mcl3dl_hip_update_device, device-resident, hipGraph replay vs plain launches"}
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_lik, cpu_q = cpu_baseline(sc, dist_weight, args.cpu_particles, n_b)
            out["cpu_baseline"] = cb
            # parity spot check of the very numbers that were timed
            n = len(cpu_lik)
            gl = d_lik[:n].cpu().numpy()
            gq = d_ratio[:n].cpu().numpy()
            out["result_check"]["max_rel_err_vs_cpu"] = float(np.max(np.abs(gl - cpu_lik) / np.maximum(np.abs(cpu_lik), 1e-30)))
            out["result_check"]["match_ratio_equal"] = bool(np.array_equal(gq, cpu_q))
        else:
            out["cpu_baseline"] = None
        line = json.dumps(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0:
        print(line, flush=True)  # the one JSON line, last on stdout


if __name__ == "__main__":
    main()
