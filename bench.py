#!/usr/bin/env python3
"""bench.py — one measurement update per step on BASELINE.json's headline configuration.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic input: likelihood-field kernel over
N_p particles x N_s scan points (+ beam kernel when the workload has beam points), pf::measure
(weight multiply, {sum w, sum w ln w, ratio min/max} reduction, one all-reduce when N > 1, normalise +
entropy).  One GPU steps through the single-GPU entry point mcl3dl_hip_update_device (one C call); with N > 1 (or
--force-dist) the split form runs with the RCCL all-reduce between its halves.  Inputs (map structures, ordered scan, poses, prior weights) are resident in HBM before the
timed region starts (the task contract's definition of `value`); the SURVEY.md §8d form of the same update —
host buffers in, host buffers out: scan upload + pose H2D + kernels + weight D2H — is timed next to it and
reported as `update_8d`, and the node's own call site through the drop-in C++ classes as `route_a`.

Particles shard across ranks; map and scan are replicated.  `--scaling weak` (default): every rank holds the
workload's full particle count.  `--scaling strong`: the workload's particles are split over the ranks.  With
N > 1 both are measured (the one not asked for under `other_scaling`).

Prints ONE JSON line on rank 0 with the driver's contract keys plus `roofline` and `cpu_baseline`.

`roofline` (dominant kernel = the likelihood kernel): every fraction is <= 1 and tied to a resource. Resources whose peak is
a DOCUMENTED figure of /opt/skills/guides/MI355X_MICROARCH.md (`peak_kind` "documented"):
  hbm          HBM-side bytes per launch from the committed PMC pass (FETCH_SIZE x 2 + WRITE_SIZE, profiles/) / live kernel
               time / 8 TB/s.  Small by construction: the index is built so that the working set of a scan tile stays in one
               XCD's L2 — `hbm_target` answers north_star's 40 %-of-HBM target explicitly ("not applicable", with the reason).
  l2           L1->L2 read requests per launch (TCP_TCC_READ_REQ, PMC) x the bytes one request of THIS access pattern moves
               (64: calibrated by profiles/l2_calib.hip, profiles/r03a_l2_calib.json) / live kernel time / 34.5 TB/s
  valu_issue   wave64 VALU instructions per launch (SQ_INSTS_VALU, PMC) / live kernel time against 1024 SIMDs x 2.4 GHz / 2
               cycles per instruction
`bound` names the largest of these three; `achieved / peak / frac` repeat that entry.  Next to them, against ceilings this
repository MEASURED on the part (`peak_kind` "measured", listed under `frac_vs_measured_peaks`, never as `frac`):
  l2_requests        the request rate against what a kernel doing nothing but the quad-cooperative 64-byte record fetch
                     sustains from an L2-resident set (profiles/l2_calib.hip)
  l1_access          L1 (TCP) cache-line accesses per launch / (CUs x kernel cycles) against ~1.15 per cycle and CU (what bound
                     the kernel before the cooperative fetch)
  valu_issue_priced  the instruction mix by class (per-class PMC counters) x the measured cycles per instruction of each class
                     (profiles/r02d_valu_microbench.txt: plain f32 add / mul / mov 2.44, fma / min / compares / conversions
                     4.17, transcendental 8.18); classes the counters do not split are priced between the two rates
                     (frac_low .. frac_high, frac = their mean)
The committed counters belong to ONE launch shape per workload tag (profiles/<session>_<tag>_pmc_summary.csv; C2j = C2 with
--map-jitter): they are used only when their wavefront count equals this launch's — any other shape gets no fractions
(`counters_note` says why).  The canonical algorithmic bytes of SURVEY.md §8d (27-cell structure, 16 + 27*4 + 16*K per
evaluation) are kept as `algorithmic_bytes_per_launch`, NOT divided by the HBM peak: the shipped index never reads them (it
reads 68 B per evaluation, from L2).

Environment (debugging / tests only): MCL3DL_BENCH_SHARE_GPU=1 puts every rank of a torch.distributed.run launch on cuda:0
with the collective over gloo (the multi-rank control flow on a one-GPU box; the line announces the mode, its numbers mean
nothing); MCL3DL_BENCH_TRACE_HANG=<seconds> dumps every thread's Python stack after that long and exits.
"""
import argparse
import contextlib
import gc
import json
import os
import re
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)
L2_PEAK_GBPS = 34500.0   # same guide, "L2 (per XCD)": ~34.5 TB/s aggregate
L2_REQ_BYTES_FALLBACK = 128.0   # bytes one TCP_TCC_READ_REQ moves when no calibration run is committed (profiles/*_l2_calib.json)
VALU_DOC_CYCLES = 2.0    # same guide, "Wave scheduling" / "Per-instruction cycle constants": a wave64 VALU instruction occupies
                         # its SIMD for 2 cycles -> documented issue peak = 1024 SIMDs x 2.4 GHz / 2 wave-instructions per second
HBM_TARGET_FRAC = 0.40   # BASELINE.json north_star: ">= 40 % of HBM-read roofline"
N_SIMD = 256 * 4
CLOCK_HZ = 2.4e9         # sustained shader clock after the pre-warm (GRBM_GUI_ACTIVE / kernel time, profiles/)
# cycles one wave64 VALU instruction occupies a SIMD, by class (profiles/r02*_valu_microbench.txt, two wavefronts per
# SIMD; fallback values = the round-2 measurement): plain VOP2 f32 add / sub / mul, moves and the simple integer ops issue
# at the full rate (~2.6), f32 fma / min / max / compares / selects / conversions / left shifts / three-operand and 64-bit
# integer ops and every f64 op at half of it (~4.4), transcendentals at ~8.4
VALU_COST_FALLBACK = {"full": 2.4, "half": 3.5, "trans": 6.5}
L1_ACCESS_CEILING = 1.15  # cache-line accesses per cycle and CU: the rate the pre-cooperative kernel sat at while flat
                          # against -15 % VALU and 2x loads in flight (profiles/r02b_C2_pmc_summary.csv); no documented peak

def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", help="C1..C5 of BASELINE.json (default C2: 4096 x 16k, 1M-pt map)")
    ap.add_argument("--particles", type=int, default=0, help="override the workload's particle count")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, the driver's contract): every rank holds the workload's full particle count; "
                         "strong: the workload's particles are split over the ranks (BASELINE config 4 is a fixed "
                         "262 144-particle problem sharded over 8 GPUs). With N > 1 the other mode is measured too")
    ap.add_argument("--dist-weight-z", type=float, default=1.0)
    ap.add_argument("--lik-index", type=int, default=2,
                    help="2 = candidate records (default), 1 = candidate runs, 0 = 27-cell scan")
    ap.add_argument("--cand-voxel-ratio", type=float, default=0.0, help="0 = the library picks it per map")
    ap.add_argument("--cand-phase", type=float, default=0.5)
    ap.add_argument("--lik-tiled", type=int, default=1)
    ap.add_argument("--lik-tiled-min", type=int, default=-1,
                    help="scans of at least this many points take the tiled kernel (-1 = the library's default)")
    ap.add_argument("--lik-coop", type=int, default=-1,
                    help="quad-cooperative record fetch (-1 = the library's default, 1 = on, 0 = every lane fetches its "
                         "own record)")
    ap.add_argument("--lik-defer", type=int, default=-1,
                    help="overflow rounds of the tiled kernel deferred and run densely: 0 never, 1 whenever the records allow it, "
                         "2 on crowded maps (library default); -1 = leave the default")
    ap.add_argument("--cand-packed", type=int, default=-1, help="packed w words in the voxel records (library default 1)")
    ap.add_argument("--cand-bound", type=int, default=-1, help="skip bound of the overflow candidates in the voxel records (library default 1)")
    ap.add_argument("--cand-record-parts", type=int, default=-1,
                    help="inline candidates per voxel record: 4 (64 bytes), 8 (128 bytes), 0 = chosen per map (-1 = default)")
    ap.add_argument("--pf-fused", type=int, default=-1, help="pf::measure as one kernel on one GPU (-1 = the library's default)")
    ap.add_argument("--beam-points", type=int, default=0, help="override the beam scan size N_b")
    ap.add_argument("--map-jitter", type=float, default=0.0,
                    help="displace every map point uniformly by +-this (m): voxel-filter centroids instead of a lattice")
    ap.add_argument("--jitter-check", type=float, default=0.045,
                    help="also time the likelihood kernel on the same map with every point displaced by +-this "
                         "(extra key `map_jitter`; 0 = skip; C2/C3 only)")
    ap.add_argument("--lik-small", type=int, default=1)
    ap.add_argument("--overlap-models", type=int, default=1)
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="wall time of untimed updates before the W warm-up steps (GPU clock ramp), 0 = none")
    ap.add_argument("--timing-mask", type=int, default=1,
                    help="kernel groups timed with hipEvents INSIDE the timed region (bit 0 likelihood = the roofline "
                         "kernel, 1 beam, 2 pf); the others are timed in a second pass of the same steps")
    ap.add_argument("--scan-points", type=int, default=0, help="override the number of likelihood scan points")
    ap.add_argument("--lik-group", type=int, default=0, help="0 = the library's choice (16 / 8 / 4 by launch size)")
    ap.add_argument("--strict-order", type=int, default=-1,
                    help="-1 = the library's default (2: the likelihood terms are replayed in the reference's float order "
                         "from 32 768 scan points up), 0 = fp64 sums always, 1 = reference float summation order for "
                         "likelihoods AND weights (bit-identical; slower), 3 = the float recurrence inside the likelihood "
                         "kernel in the engine's scan order (bit-identical to the reference on the scan in that order)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the all-reduce even with one rank (exercises the RCCL path)")
    ap.add_argument("--also-other-scaling", action="store_true",
                    help="time the other scaling mode too even with one rank (exercises the N > 1 reporting path)")
    ap.add_argument("--in-process", action="store_true",
                    help="drive all --gpus N devices from THIS one process through the device-group C ABI "
                         "(mcl3dl_hip_group_*: worker thread per GPU, RCCL all-reduce inside the library) — the route the "
                         "reference's single process would use — instead of one process per GPU; prints the same JSON line")
    ap.add_argument("--sort-poses", choices=("none", "yaw", "xy", "cluster"), default="none",
                    help="A/B: re-order the synthetic particles by pose before the run (what pose-ordered particle groups would buy)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra keys (post-update reductions, resampling, fused update, route A, jitter check)")
    ap.add_argument("--route-a-reps", type=int, default=40,
                    help="timed repetitions of the node's call site through the drop-in C++ classes "
                         "(tests/cpp/adapter_demo.bin), 0 = skip")
    ap.add_argument("--cpu-particles", type=int, default=0,
                    help="particles in the CPU-baseline sample (0 = as many as ~12 s of one core buys, at most all)")
    return ap.parse_args()


PMC_SOURCES = {"beam": ["beam_kernels.h", "device_math.h", "map_structs.h", "Makefile"],
               "lik": ["likelihood_kernels.h", "device_math.h", "map_structs.h", "map_compiler.h", "Makefile"]}


def source_sha(name):
    import hashlib
    return hashlib.sha256(open(os.path.join(ROOT, "mcl_3dl_amd", "csrc", name), "rb").read()).hexdigest()[:16]


def pmc_counters(kernel_prefix, workload, check_sources=True, only=None):
    """Mean per launch of every counter the newest committed PMC summary for this workload holds for the kernel
    (profiles/*_pmc_summary.csv, separate --pmc passes: profiles/run_profiles.sh). Returns (dict, path, note); the counters
    are REFUSED (None, None, why) when the summary does not say which sources it profiled or when one of the kernel's sources
    (or the build flags) has changed since: a kernel edit after the last --pmc session must not keep the old counters.
    check_sources=False / only="r03" (file-name prefix): the historical summaries, for the tests that pin their arithmetic."""
    import csv
    import glob
    vals, src, shas = {}, None, {}
    # exactly this workload tag (r02h_C2_pmc_summary.csv, not r02h_C2j_...: the jittered map has its own counters)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc*summary.csv" % workload))):
        if only and not os.path.basename(path).startswith(only):
            continue
        here, tags = {}, {}
        for row in csv.DictReader(open(path)):
            if row["kernel"] == "__source__":
                tags[row["counter"]] = row["mean_per_launch"]
            elif row["kernel"].startswith(kernel_prefix):
                here[row["counter"]] = float(row["mean_per_launch"])
        if here:
            if tags != shas and src is not None and tags.get("git_head") != shas.get("git_head"):
                vals = {}   # counters of different builds are not mixed: the newest file that has this kernel wins whole
            vals.update(here)
            src, shas = path, tags
    if not vals:
        return None, None, None
    rel = os.path.relpath(src, ROOT)
    family = "beam" if "beam_kernel" in kernel_prefix else "lik"
    if not check_sources:
        return vals, rel, None
    if not shas:
        return None, None, "%s does not record the sources it profiled (written before round 4): not used" % rel
    changed = [n for n in PMC_SOURCES[family] if shas.get(n) != source_sha(n)]
    if changed:
        return None, None, ("%s profiled commit %s; %s changed since (sha %s then): counters not used — run profiles/run_profiles.sh again"
                            % (rel, shas.get("git_head", "?"), ", ".join(changed), ", ".join(str(shas.get(n)) for n in changed)))
    return vals, rel, None


def l2_calibration():
    """What one TCP_TCC_READ_REQ is worth, from the newest committed run of profiles/l2_calib.hip (known byte counts per
    kernel, counters from separate --pmc passes): {"bytes_per_request": coalesced 16 B/lane stream, "requests_per_record64":
    requests the quad-cooperative fetch of one 64-byte record costs, "record64_requests_per_s": the rate at which the
    L2s served that fetch from a set resident in every XCD's L2 (a measured ceiling)}. Falls back to 128 B, uncalibrated."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_l2_calib.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d.get("bytes_per_request", 0) > 0:
                return dict(d, source=os.path.relpath(path, ROOT))
        except (OSError, ValueError):
            continue
    return {"bytes_per_request": L2_REQ_BYTES_FALLBACK, "source": "bench.py L2_REQ_BYTES_FALLBACK (no calibration run committed)"}


def kernel_resources(pmc, kernel_s, cost, cost_src, l2cal=None):
    """Counter-derived utilisation of one kernel: {resource: {achieved, peak, unit, frac, peak_kind}} and the HBM-side bytes
    per launch. pmc = mean per launch of the counters of separate --pmc passes (profiles/), kernel_s = measured seconds per
    launch. `peak_kind` says where the denominator comes from: "documented" (/opt/skills/guides/MI355X_MICROARCH.md) or
    "measured" (a micro-benchmark or an observed ceiling of this repository: profiles/)."""
    res = {}
    traffic = None
    l2cal = l2cal or l2_calibration()
    if pmc and kernel_s > 0:
        if "FETCH_SIZE" in pmc:
            # FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes fetched
            # (MI355X_MICROARCH.md, HBM section), hence the factor 2
            traffic = (2.0 * pmc["FETCH_SIZE"] + pmc.get("WRITE_SIZE", 0.0)) * 1024.0
            gbps = traffic / kernel_s / 1e9
            res["hbm"] = {"achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                          "peak_kind": "documented",
                          "note": "cache-resident by construction: tile-major, XCD-aware launch keeps a scan "
                                  "tile's voxel records in one XCD's L2"}
        if "TCP_TCC_READ_REQ_sum" in pmc:
            # calibrated (profiles/l2_calib.hip): a request moves one 128-byte line in a coalesced stream, and ONE request
            # fetches a whole 64-byte voxel record — these kernels gather records and 4-byte table entries, so their L2 bytes
            # are requests x 64 B at most; the 128 B per request round 2 assumed is kept as `frac_if_full_lines`
            bpr = float(l2cal.get("bytes_per_record_request", l2cal["bytes_per_request"]))
            gbps = pmc["TCP_TCC_READ_REQ_sum"] * bpr / kernel_s / 1e9
            res["l2"] = {"achieved": gbps, "peak": L2_PEAK_GBPS, "unit": "GB/s", "frac": gbps / L2_PEAK_GBPS,
                         "frac_if_full_lines": gbps / L2_PEAK_GBPS * float(l2cal["bytes_per_request"]) / bpr,
                         "peak_kind": "documented",
                         "requests_per_launch": pmc["TCP_TCC_READ_REQ_sum"], "bytes_per_request": bpr,
                         "bytes_per_request_source": l2cal["source"]}
            if l2cal.get("record64_requests_per_s"):
                rate = pmc["TCP_TCC_READ_REQ_sum"] / kernel_s
                res["l2_requests"] = {"achieved": rate / 1e9, "peak": l2cal["record64_requests_per_s"] / 1e9,
                                      "unit": "G requests/s", "frac": rate / l2cal["record64_requests_per_s"],
                                      "peak_kind": "measured",
                                      "note": "peak = request rate of the quad-cooperative 64-byte record fetch alone over an "
                                              "L2-resident set (profiles/l2_calib.hip) — a ceiling of THAT access pattern, not "
                                              "of the part: a kernel that also fetches overflow records (map of centroids) has "
                                              "been measured above it, and the fraction is printed unclamped"}
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in pmc:
            rate = pmc["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / (kernel_s * CLOCK_HZ)
            res["l1_access"] = {"achieved": rate, "peak": L1_ACCESS_CEILING, "unit": "cache-line accesses/cycle/CU",
                                "frac": rate / L1_ACCESS_CEILING, "peak_kind": "measured",
                                "accesses_per_launch": pmc["TCP_TOTAL_CACHE_ACCESSES_sum"],
                                "note": "peak = measured ceiling (see bench.py L1_ACCESS_CEILING), not a datasheet value"}
        if "SQ_INSTS_VALU" in pmc:
            n_all = pmc["SQ_INSTS_VALU"]
            avail = N_SIMD * kernel_s * CLOCK_HZ
            # (i) against the documented issue rate: every wave64 VALU instruction 2 cycles of its SIMD
            doc = n_all * VALU_DOC_CYCLES / avail
            res["valu_issue"] = {"achieved": n_all / kernel_s / 1e9, "peak": N_SIMD * CLOCK_HZ / VALU_DOC_CYCLES / 1e9,
                                 "unit": "G wave64 VALU instructions/s", "frac": doc, "peak_kind": "documented",
                                 "wave_instructions_per_launch": n_all, "cycles_per_instruction": VALU_DOC_CYCLES,
                                 "note": "peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md)"}
            # (ii) the same instructions priced with the measured issue cost of each class (profiles/valu_microbench.hip).
            # Known classes are priced exactly; what the counters do not split (32-bit integer ops, moves, DPP moves,
            # compares, selects) is priced between the full and the half rate -> a low and a high estimate; frac = mean
            have_mix = "SQ_INSTS_VALU_ADD_F32" in pmc
            n_full = pmc.get("SQ_INSTS_VALU_ADD_F32", 0.0) + pmc.get("SQ_INSTS_VALU_MUL_F32", 0.0)
            n_trans = pmc.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            n_half = sum(pmc.get(k, 0.0) for k in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT64",
                                                    "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
                                                    "SQ_INSTS_VALU_FMA_F64"))
            n_mixed = max(n_all - n_full - n_trans - n_half, 0.0)
            fixed = n_full * cost["full"] + n_half * cost["half"] + n_trans * cost["trans"]
            lo, hi = (fixed + n_mixed * cost["full"]) / avail, (fixed + n_mixed * cost["half"]) / avail
            mid = 0.5 * (lo + hi)
            res["valu_issue_priced"] = {"achieved": mid * avail / kernel_s / 1e9, "peak": N_SIMD * CLOCK_HZ / 1e9,
                                        "unit": "G SIMD-cycles/s", "frac": mid, "frac_low": lo, "frac_high": hi,
                                        "peak_kind": "measured",
                                        "full_rate": n_full, "half_rate": n_half, "transcendental": n_trans,
                                        "between_full_and_half_rate": n_mixed, "cycles_per_instruction": cost,
                                        "cycles_per_instruction_source": cost_src, "mix_from_counters": have_mix,
                                        "note": "SIMD cycles the instruction mix needs at the issue cost measured per class "
                                                "(plain f32 add/mul/mov ~2.4, fma/min/max/cvt/compare/f64 ~4.2, sqrt ~8.2 "
                                                "cycles at the nominal clock) over the SIMD cycles of the launch"}
    return res, traffic


VALU_FULL_ROWS = ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_mov_b32", "v_and_b32", "v_add_u32")
VALU_HALF_ROWS = ("v_max_f32", "v_min_f32", "v_fma_f32", "v_cvt_flr_i32_f32", "v_mul_u32_u24", "v_cmp_lt_f32", "v_lshl_or_b32",
                  "v_add_f64")
VALU_TRANS_ROWS = ("v_sqrt_f32", "v_rcp_f32")


def microbench_rows(path, waves_per_simd=8):
    """{row label: cyc@2.4GHz} of one committed run of profiles/valu_microbench.hip at `waves_per_simd` wavefronts per SIMD,
    "x8 independent" rows only. The label is everything ahead of that marker, so `v_mul_f32 (sgpr src)` and `v_mul_f32`
    are different rows (round 2 keyed them by the opcode alone and the former overwrote the latter)."""
    rows = {}
    for line in open(path):
        m = re.match(r"(v_.*?)\s+x8 independent\s+(\d+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s*$", line)
        if m and int(m.group(2)) == waves_per_simd:
            rows[m.group(1).strip()] = float(m.group(6))   # kernel time x 2.4 GHz / (instructions x wavefronts per SIMD)
    return rows


def valu_costs():
    """Cycles one wave64 VALU instruction occupies a SIMD, by class, from the newest committed run of
    profiles/valu_microbench.hip (the version that places exactly W wavefronts on every SIMD: r02d onwards): column
    cyc@2.4GHz (kernel time x the nominal clock, the same clock `valu_issue_priced` prices the likelihood kernel's time
    with) at eight wavefronts per SIMD, the occupancy the likelihood kernels run at: full = mean of the plain
    v_mul / add / sub_f32, v_mov, v_and, v_add_u32 rows, half = mean of the max / min / fma / cvt / 24-bit multiply /
    compare / lshl_or / f64-add rows, trans = v_sqrt_f32 and v_rcp_f32."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_microbench.txt")), reverse=True):
        if os.path.basename(path) < "r02d":
            continue  # earlier runs measured the dispatcher's uneven spread of 256-thread groups, not the pipe
        rows = microbench_rows(path)
        full = [rows[k] for k in VALU_FULL_ROWS if k in rows]
        half = [rows[k] for k in VALU_HALF_ROWS if k in rows]
        trans = [rows[k] for k in VALU_TRANS_ROWS if k in rows]
        if full and half and trans:
            return ({"full": sum(full) / len(full), "half": sum(half) / len(half), "trans": sum(trans) / len(trans)},
                    os.path.relpath(path, ROOT))
    return dict(VALU_COST_FALLBACK), "bench.py VALU_COST_FALLBACK (no micro-benchmark run committed)"


def cpu_baseline(sc, dist_weight, n_particles, beam_points, scan_order=None):
    """The reference's own measure() loop on this box's host cores (oracle/_ref when built, the C port otherwise),
    single thread = the reference's execution model (src/mcl_3dl.cpp:1466), on a bounded sample of the same workload.
    scan_order (strict_order = 3): the reference is given the scan in the engine's order (mcl3dl_hip_scan_order), the
    order the timed kernel summed it in."""
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
    scan = sc.scan_lik if scan_order is None else np.ascontiguousarray(sc.scan_lik[scan_order])
    lik, q, sec = o.likelihood_measure(sc.poses[:n], scan, threads=1, return_time=True)
    evals = n * len(sc.scan_lik)
    out = {"value": evals / sec, "unit": "particle·point evals/s", "cores": 1,
           "kind": "reference" if kind == "ref" else "port",
           "sample": "%d of the workload's particles x %d points, likelihood model, 1 thread, %.1f s"
                     % (n, len(sc.scan_lik), sec),
           "note": "nearest-neighbour index behind pcl::KdTreeFLANN is this repo's stand-in (PCL/FLANN not installed)"}
    threads = o.max_threads()
    if threads > 1:
        _, _, sec_mt = o.likelihood_measure(sc.poses[:n], scan, threads=threads, return_time=True)
        out["all_cores"] = {"value": evals / sec_mt, "cores": threads}
    return out, lik, q


def _tiled_group(n_s, n_p, forced):
    """The tiled kernel's particles-per-work-group as the library picks it (host_measure.h:plan_lik)."""
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


@contextlib.contextmanager
def no_gc(collect=False):
    """Timed host loops run with Python's cyclic garbage collector off (a generation-2 pass over a process that has torch
    loaded takes tens of milliseconds — two orders of magnitude more than one update). It does NOT collect on entry unless
    asked to: a collection right before a timed region leaves the GPU idle for ~35 ms, the clocks fall back, and the first
    tens of launches of the region run ~20 % slow (profiles/r03o_C2_gc_gap_trace.txt: 231 us per launch through the
    pre-warm, 34 ms of nothing, then 281 us decaying to 255 us over the 20 timed steps). main() collects once, before the
    pre-warm, and freezes what is left."""
    was = gc.isenabled()
    if collect:
        gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def keep_gpu_warm(call, seconds=0.15, sync=None):
    """Run `call` back to back for `seconds` of wall time (clock ramp before a side measurement that follows host-only work)."""
    t = time.perf_counter()
    while time.perf_counter() - t < seconds:
        for _ in range(4):
            call()
        if sync is not None:
            sync()


def _flush_c_stdio():
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def route_a(sc, dist_weight, n_b, reps):
    """The node's call site (src/mcl_3dl.cpp:377-426) through the drop-in C++ classes over the C ABI: per-particle
    measure() virtuals answered from one batched launch per model, weights normalised by the reference's pf.h on the
    CPU. tests/cpp/adapter_demo.bin is built where the reference headers exist and travels as a file."""
    exe = os.path.join(ROOT, "tests", "cpp", "adapter_demo.bin")
    if not os.path.exists(exe) or reps <= 0:
        return None
    with tempfile.TemporaryDirectory() as tmp:
        scene, result = os.path.join(tmp, "scene.bin"), os.path.join(tmp, "result.bin")
        with open(scene, "wb") as f:
            f.write(struct.pack("<8Q", len(sc.map_xyz), len(sc.poses), len(sc.scan_lik), len(sc.scan_beam),
                                len(sc.origins), max(n_b, 1), 1, 0xFFFFFFFF))
            f.write(struct.pack("<5f", dist_weight[0], dist_weight[1], dist_weight[2], 1.0, 1.0))
            for a, dt in ((sc.map_xyz, np.float32), (sc.map_label, np.uint32), (sc.poses, np.float32),
                          (sc.odom_err, np.float32), (sc.weights, np.float32), (sc.scan_lik, np.float32),
                          (sc.scan_beam, np.float32), (sc.scan_beam_label, np.uint32), (sc.origins, np.float32)):
                f.write(np.ascontiguousarray(a, dtype=dt).tobytes())
        try:
            proc = subprocess.run([exe, scene, result, str(reps)], capture_output=True, text=True, timeout=600)
        except subprocess.TimeoutExpired:
            return {"error": "adapter_demo timed out"}
        m = re.search(r"route_a_ms_per_update (\S+)", proc.stdout)
        if proc.returncode != 0 or not m:
            return {"error": (proc.stdout + proc.stderr)[-300:]}
        ms = float(m.group(1))
        bd = re.search(r"route_a_breakdown_us pose_gather_upload (\S+) cloud_pack (\S+) measure_batch (\S+) "
                       r"batched_calls_per_update (\S+) waiting (\S+)", proc.stdout)
        breakdown = None
        if bd:
            inside = float(bd.group(1)) + float(bd.group(2)) + float(bd.group(3)) + float(bd.group(5))
            breakdown = {"pose_gather_upload_us": float(bd.group(1)), "cloud_pack_us": float(bd.group(2)),
                         "measure_batch_begin_us": float(bd.group(3)), "waiting_for_slices_us": float(bd.group(5)),
                         "batched_calls_per_update": float(bd.group(4)),
                         "reference_pf_loop_us": ms * 1e3 - inside,
                         "note": "measure_batch_begin = staging the scans + enqueueing the kernels of BOTH models in particle "
                                 "slices; waiting = blocked until the slice a particle belongs to has arrived (the GPU works "
                                 "on the later slices while the loop runs); reference_pf_loop = the rest: the node's measure "
                                 "lambda (src/mcl_3dl.cpp:399-426, a std::map per particle), 2 N virtual calls, weight "
                                 "product, normalisation and entropy on the CPU — the reference's own code"}
        return {"ms_per_update": ms, "breakdown": breakdown, "evals_per_s": len(sc.poses) * len(sc.scan_lik) / (ms * 1e-3), "reps": reps,
                "what": "pf_->measure(measure_func) through the drop-in LidarMeasurementModel{Likelihood,Beam} classes "
                        "(per-particle virtuals, host pose/cloud packing, ONE batch for both models — launched by the "
                        "first measure() of the update and delivered in particle slices while the loop runs —, weights "
                        "normalised by the reference's pf.h on the CPU)"}


def in_process_group_run(workload, n_cfg, extra_cfg, dist_weight, devices, steps, warmup, collective=None):
    """One process, len(devices) GPUs: the update through mcl3dl_hip_group_measure_update (host arrays in, host arrays out;
    particles sharded inside the library, scan ordered once and pushed to every device, one all-reduce per update). Weak
    scaling like the main line: n_cfg particles per GPU. Returns a dict for the bench line."""
    from mcl_3dl_amd import capi
    from mcl_3dl_amd.synthetic import make_config
    n_dev = len(devices)
    parts = [make_config(workload, n_p=n_cfg, seed=12345 + r, **extra_cfg) for r in range(n_dev)]
    sc = parts[0]
    poses = np.ascontiguousarray(np.concatenate([p.poses for p in parts], 0))
    n_total = len(poses)
    w0 = np.full(n_total, 1.0 / n_total, np.float32)
    n_b = len(sc.scan_beam)
    g = capi.Group(devices, collective=collective)
    try:
        if n_dev == 1:
            g.set_option("direct_single", 0)   # one GPU: still the sharded path with the (one-rank) RCCL all-reduce
        g.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=dist_weight)
        g.set_likelihood_params()
        g.set_beam_params(num_points=max(n_b, 1), dda_grid_size=0.2)
        args_u = (poses, w0, sc.scan_lik, sc.scan_beam if n_b else None, sc.scan_beam_label if n_b else None, sc.origins)
        for _ in range(max(warmup, 3)):
            g.measure_update(*args_u)
        keep_gpu_warm(lambda: g.measure_update(*args_u), 0.15)   # clock ramp on every GPU of the group (set-up, not timed)
        with no_gc():
            t0 = time.perf_counter()
            for _ in range(steps):
                g.measure_update(*args_u)
            el = time.perf_counter() - t0
        st = g.collective_stats()
        return {"n_gpus": n_dev, "particles_total": n_total, "ms_per_update": el / steps * 1e3,
                "value": float(n_total) * len(sc.scan_lik) * steps / el, "unit": "particle·point evals/s",
                "collective": "rccl" if st["rccl"] else "host", "collectives": st,
                "what": "mcl3dl_hip_group_measure_update from ONE process over %d GPU(s): host arrays in / out (PCIe included, "
                        "like update_8d), particles sharded inside the library, one all-reduce per update" % n_dev}
    finally:
        g.close()


def in_process_group_child(args, n_dev, timeout_s=120):
    """The in-process group over n_dev GPUs in a process of its OWN (`bench.py --in-process`), with a time limit: the N-rank
    RCCL bring-up inside one process has never run on more than one GPU before the driver's scaling run, and a hang there
    must cost this extra key, not the line. The child sees every GPU and none of the launcher's rendezvous variables."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE",
                        "MASTER_ADDR", "MASTER_PORT", "GROUP_WORLD_SIZE", "ROLE_NAME")
           and not k.startswith("TORCHELASTIC_")}
    cmd = [sys.executable, os.path.abspath(__file__), "--in-process", "--gpus", str(n_dev), "--workload", args.workload,
           "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.particles:
        cmd += ["--particles", str(args.particles)]
    if args.scan_points:
        cmd += ["--scan-points", str(args.scan_points)]
    if args.beam_points:
        cmd += ["--beam-points", str(args.beam_points)]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": "bench.py --in-process --gpus %d did not finish within %d s" % (n_dev, timeout_s)}
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": "bench.py --in-process --gpus %d: rc %d: %s" % (n_dev, p.returncode, (p.stderr or p.stdout)[-300:])}
    return json.loads(lines[-1])["in_process_group"]


def cloud_path_extras(eng, sc, n_s, n_b, with_cpu):
    """SURVEY.md 8f-2 / 8f-4 timings (run last: they replace the context's scan): scan preparation of an accumulated cloud
    (VoxelGrid with the node's default leaf, both clips, sample gather, device-side ordering) and the matched / unmatched
    split of the down-sampled cloud for one pose, each next to the reference's code path on one host core."""
    rng = np.random.default_rng(4)
    reps = max(60000 // max(len(sc.scan_lik), 1), 1) + 1
    raw = np.concatenate([sc.scan_lik + rng.normal(0, 0.02, sc.scan_lik.shape).astype(np.float32) for _ in range(reps)], 0)
    raw = np.ascontiguousarray(raw[:max(60000, 4 * n_s)], dtype=np.float32)
    leaf = (0.1, 0.1, 0.05)   # downsample_x / _y / _z defaults, src/parameters.cpp:88-90
    clip_lik, clip_beam = (0.5, 10.0, -2.0, 2.0), (0.5, 4.0, -2.0, 2.0)
    origins = np.array([[0.0, 0.0, 0.5]], np.float32)
    n_full, n_lik, n_beam = eng.scan_begin(raw, None, leaf=leaf, clip_lik=clip_lik, clip_beam=clip_beam)
    ns, nb = min(n_s, max(n_lik, 1)), (min(n_b, max(n_beam, 1)) if n_b else 0)
    def once():
        f, l, b = eng.scan_begin(raw, None, leaf=leaf, clip_lik=clip_lik, clip_beam=clip_beam)
        il = rng.integers(0, max(l, 1), ns).astype(np.uint32)
        ib = rng.integers(0, max(b, 1), nb).astype(np.uint32) if nb else None
        eng.scan_finish(il, ib, origins=origins)
    for _ in range(3):
        once()
    # per-call times with Python's cyclic collector off: a generation-2 pass of a torch process takes ~35 ms and once landed
    # inside this loop (round-2 lines read 4-6 ms instead of 0.3: scripts/r03_scanprep_bisect.py)
    calls = []
    with no_gc():
        for _ in range(10):
            t0 = time.perf_counter()
            once()
            calls.append((time.perf_counter() - t0) * 1e3)
    prep_ms = float(np.median(calls))
    res = {"scan_preparation": {"ms": prep_ms, "ms_mean": float(np.mean(calls)), "ms_max": float(np.max(calls)), "raw_points": int(len(raw)), "after_voxel_grid": int(n_full),
                                "after_clip": [int(n_lik), int(n_beam)], "sampled": [int(ns), int(nb)], "leaf": list(leaf),
                                "what": "mcl3dl_hip_scan_begin + _scan_finish: H2D of the accumulated cloud, VoxelGrid, both clip "
                                        "filters, gather of the drawn samples, device-side scan ordering (host wall time, index "
                                        "draw included)"}}
    pose = np.asarray(sc.true_pose, np.float32)
    eng.scan_begin(raw, None, leaf=leaf, clip_lik=clip_lik, clip_beam=clip_beam)
    m, u = eng.match_split(pose)
    out_m, out_u = np.zeros((max(n_full, 1), 3), np.float32), np.zeros((max(n_full, 1), 3), np.float32)
    calls = []
    with no_gc():
        for _ in range(10):
            t0 = time.perf_counter()
            n_m, n_u = eng.match_split_into(pose, out_m, out_u)
            calls.append((time.perf_counter() - t0) * 1e3)
    assert n_m == len(m) and n_u == len(u) and np.array_equal(out_m[:n_m], m) and np.array_equal(out_u[:n_u], u)
    res["match_split"] = {"ms": float(np.median(calls)), "ms_max": float(np.max(calls)), "points": int(n_full),
                          "matched": int(n_m), "unmatched": int(n_u),
                          "what": "mcl3dl_hip_match_split of the down-sampled cloud left on the device (src/mcl_3dl.cpp:761-805) "
                                  "into the caller's (pageable) arrays: classification, both compactions by one kernel that "
                                  "writes into page-locked memory, one polled completion"}
    # the two linear-time map structures (cell-sorted exact-NN grid, DDA occupancy + voxel index): device builders
    # (default) next to the sequential host form they replaced
    q1 = np.asarray(sc.true_pose[:3], np.float32).reshape(1, 3)
    gb = {}
    for mode, tag in ((1, "host"), (0, "device")):
        eng.set_option("grid_build_host", mode)   # marks both structures dirty
        eng.radius_search(q1, 0.3)
        gb["lik_grid_%s_wall_ms" % tag] = eng.get_option("lik_grid_build_wall_ms")
        if n_b:
            eng.beam_status(q1, q1 + np.float32(1.0))
            gb["dda_grid_%s_wall_ms" % tag] = eng.get_option("dda_grid_build_wall_ms")
    gb["lik_grid_device_ms"] = eng.get_option("lik_grid_build_ms")
    if n_b:
        gb["dda_grid_device_ms"] = eng.get_option("dda_grid_build_ms")
    gb["map_points"] = int(len(sc.map_xyz))
    gb["what"] = ("build of the cell-sorted map (radius search, matched / unmatched, lik_index 0) and of the DDA grid: "
                  "wall = host time of the whole build incl. map upload; device_ms = hipEvent time of the device builder")
    res["grid_build"] = gb
    if with_cpu:
        from oracle import pyoracle
        if pyoracle.available("ref"):
            orc = pyoracle.Oracle("ref")
            orc.set_likelihood_params(pyoracle.LikelihoodParams(num_points=ns))
            orc.set_beam_params(pyoracle.BeamParams(num_points=max(nb, 1)))
            t0 = time.perf_counter()
            full, full_label = orc.voxel_grid(raw, None, leaf)
            orc.filter_uniform(0, full, full_label, 1, max(ns, 1))
            if nb:
                orc.filter_uniform(1, full, full_label, 2, max(nb, 1))
            res["scan_preparation"]["cpu_reference_ms"] = (time.perf_counter() - t0) * 1e3
            res["scan_preparation"]["cpu_note"] = ("restated pcl::VoxelGrid + the reference's filter() and PointCloudUniformSampler, "
                                                   "1 thread (oracle/_ref)")
    return res


def main():
    args = parse()
    if os.environ.get("MCL3DL_BENCH_TRACE_HANG"):
        # debugging aid: every thread's Python stack on stderr after that many seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MCL3DL_BENCH_TRACE_HANG"]), exit=True)
    import torch
    import torch.distributed as dist
    from mcl_3dl_amd import capi
    from mcl_3dl_amd.distributed import shard_bounds
    from mcl_3dl_amd.synthetic import CONFIGS, make_config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.in_process and world == 1:
        # ---- one process, N GPUs (no torch.distributed): the line's headline is the in-process group's update
        if not torch.cuda.is_available() or torch.cuda.device_count() < args.gpus:
            raise SystemExit("--in-process --gpus %d needs that many visible GPUs" % args.gpus)
        cfg = CONFIGS[args.workload]
        n_cfg = args.particles or cfg["n_p"]
        extra_cfg = dict(n_s=args.scan_points) if args.scan_points else {}
        if args.beam_points:
            extra_cfg["n_b"] = args.beam_points
        r = in_process_group_run(args.workload, n_cfg, extra_cfg, (1.0, 1.0, args.dist_weight_z), list(range(args.gpus)),
                                 args.steps, args.warmup)
        _flush_c_stdio()   # RCCL announces itself through C stdio: out before the line, which must be the last one
        print(json.dumps({
            "metric": "particle·point evals/sec; filter-update Hz @ 4096 particles × 16k-pt scan", "value": r["value"],
            "unit": r["unit"], "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["ms_per_update"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d particles/GPU, in-process device group" % (args.workload, n_cfg),
                       "parallelism": "ONE process, particles sharded x%d inside libmcl3dl_hip (worker thread per GPU), "
                                      "map+scan replicated, 1 %s all-reduce/update" % (args.gpus, r["collective"]),
                       "particles_total": r["particles_total"]},
            "value_definition": "host arrays in, host arrays out (PCIe included): the in-process route has no device-resident "
                                "form — the reference's process keeps its particles on the host",
            "roofline": None, "cpu_baseline": None, "in_process_group": r}), flush=True)
        return
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # MCL3DL_BENCH_SHARE_GPU=1 (tests only): every rank on cuda:0 with the collective over gloo — RCCL refuses two ranks on one
    # device — so that the multi-rank control flow of this file (shards, both scaling modes, barriers, MAX over ranks, the
    # in-process child) runs on a one-GPU box before the driver's N-GPU run. The line says so; its numbers mean nothing.
    share_gpu = os.environ.get("MCL3DL_BENCH_SHARE_GPU") == "1" and world > 1
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        # RCCL writes a version banner through C stdio when its communicator comes up; push it out NOW so that rank 0's
        # JSON line is the last thing on stdout
        warm = torch.zeros(1, device=dev)
        dist.all_reduce(warm)
        torch.cuda.synchronize(dev)
        _flush_c_stdio()
        # a CPU-side group: the other ranks wait THERE (no kernel spinning on their GPUs) while rank 0 drives every GPU
        # from its one process (in_process_group, below)
        cpu_pg = dist.new_group(backend="gloo") if world > 1 else None

    cfg = CONFIGS[args.workload]
    n_cfg = args.particles or cfg["n_p"]
    extra_cfg = dict(n_s=args.scan_points) if args.scan_points else {}
    if args.map_jitter:
        extra_cfg["map_jitter"] = args.map_jitter
    if args.beam_points:
        extra_cfg["n_b"] = args.beam_points  # C3 stress case N_b = 16 384
    sc = make_config(args.workload, n_p=n_cfg, seed=12345, **extra_cfg)
    if args.sort_poses != "none":
        # A/B only (round 6): what the tiled kernel would gain if the particles of a work-group were neighbours
        # in pose space (16 consecutive particles share a work-group). The CALLER's array is re-ordered here, outside any timed
        # region: an upper bound of what an in-engine ordering pass could buy, before paying for that pass.
        q = sc.poses[:, 3:7].astype(np.float64)
        yaw = np.arctan2(2.0 * (q[:, 3] * q[:, 2] + q[:, 0] * q[:, 1]), 1.0 - 2.0 * (q[:, 1] ** 2 + q[:, 2] ** 2))
        x, y = sc.poses[:, 0].astype(np.float64), sc.poses[:, 1].astype(np.float64)
        if args.sort_poses == "yaw":
            order = np.argsort(yaw, kind="stable")
        elif args.sort_poses == "xy":
            order = np.lexsort((y, np.floor(x / 0.1)))
        else:  # "cluster": yaw slabs of 256 particles, inside a slab x strips of 16, inside a strip by y
            order = np.argsort(yaw, kind="stable")
            out = []
            for lo in range(0, len(order), 256):
                slab = order[lo:lo + 256]
                slab = slab[np.argsort(x[slab], kind="stable")]
                for l2 in range(0, len(slab), 16):
                    strip = slab[l2:l2 + 16]
                    out.append(strip[np.argsort(y[strip], kind="stable")])
            order = np.concatenate(out)
        sc.poses = np.ascontiguousarray(sc.poses[order])
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
    if args.cand_record_parts >= 0:
        eng.set_option("cand_record_parts", args.cand_record_parts)
    if args.lik_defer >= 0:
        eng.set_option("lik_defer", args.lik_defer)
    if args.cand_packed >= 0:
        eng.set_option("cand_packed", args.cand_packed)
    if args.cand_bound >= 0:
        eng.set_option("cand_bound", args.cand_bound)
    if args.strict_order >= 0:
        eng.set_option("strict_order", args.strict_order)
    strict_mode = int(eng.get_option("strict_order"))
    # mirror of host_measure.h:lik_mode — which kernel family the launch takes and how its terms are added
    def _lik_mode(np_, ns_):
        tmin = int(eng.get_option("lik_tiled_min"))
        by_size = bool(int(eng.get_option("lik_tiled")) and np_ >= 4 and (ns_ >= tmin or (np_ >= 256 and 4 * ns_ >= 3 * tmin)))
        if strict_mode == 3:
            return dict(tiled=True, rows=False, replay=False)
        exact = strict_mode == 1 or (strict_mode == 2 and (ns_ <= int(eng.get_option("strict_exact_max")) or
                                                           ns_ >= int(eng.get_option("strict_auto_min"))))
        rows_fit = ns_ <= 12288
        if not exact:
            return dict(tiled=by_size, rows=(not by_size) and strict_mode == 2 and rows_fit, replay=False)
        if rows_fit and (not by_size or np_ < 2048):
            return dict(tiled=False, rows=True, replay=False)
        return dict(tiled=True, rows=False, replay=True)
    eng.set_option("lik_tiled", args.lik_tiled)
    eng.set_option("lik_small", args.lik_small)
    eng.set_option("overlap_models", args.overlap_models)
    eng.set_option("lik_group", args.lik_group)
    if args.lik_tiled_min >= 0:
        eng.set_option("lik_tiled_min", args.lik_tiled_min)
    tiled_min = int(eng.get_option("lik_tiled_min"))
    if args.lik_coop >= 0:
        eng.set_option("lik_coop", args.lik_coop)
    lik_coop = int(eng.get_option("lik_coop"))
    if args.pf_fused >= 0:
        eng.set_option("pf_fused", args.pf_fused)
    eng.set_beam_params(num_points=max(n_b, 1), dda_grid_size=0.2)
    eng.upload_scan(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)

    class Shard:
        """This rank's particles for one scaling mode, resident on the device."""

        def __init__(self, mode):
            self.mode = mode
            if mode == "weak":
                # every rank holds a full-size particle set drawn with its own seed
                poses = sc.poses if rank == 0 else make_config(args.workload, n_p=n_cfg, seed=12345 + rank, **extra_cfg).poses
                self.n_total = n_cfg * world
            else:
                lo, hi = shard_bounds(n_cfg, world, rank)
                poses = sc.poses[lo:hi]
                self.n_total = n_cfg
            self.n = len(poses)
            self.host_poses = poses
            self.d_pose = torch.from_numpy(np.ascontiguousarray(poses)).to(dev).contiguous()
            self.d_w0 = torch.full((self.n,), 1.0 / self.n_total, dtype=torch.float32, device=dev)
            self.d_w = self.d_w0.clone()
            self.d_lik = torch.empty(self.n, dtype=torch.float32, device=dev)
            self.d_ratio = torch.empty(self.n, dtype=torch.float32, device=dev)
            self.d_beam = torch.empty(self.n, dtype=torch.float32, device=dev)
            self.d_stats = torch.zeros(4, dtype=torch.float32, device=dev)
            # one all-reduce(SUM) carries the sums and, in per-rank slots, the max/min candidates
            self.d_pack = torch.zeros(2 + 2 * world, dtype=torch.float64, device=dev)

    def step(sh):
        sh.d_w.copy_(sh.d_w0)  # resampling leaves uniform weights before every update (pf.h:203,207)
        if not use_dist:
            # one GPU: the single-GPU entry point — measure + pf::measure in ONE C call (pf::measure as one kernel up to
            # 4096 particles); the split form below exists for the collective between its halves
            eng.update_device(sh.d_pose, sh.n, sh.d_w, sh.d_stats, d_lik=sh.d_lik, d_ratio=sh.d_ratio,
                              d_beam=sh.d_beam if n_b else None)
            return
        eng.measure_device(sh.d_pose, sh.n, sh.d_lik, sh.d_ratio, sh.d_beam if n_b else None)
        eng.pf_partial_device(sh.d_w, sh.d_lik, sh.d_beam if n_b else None, None, sh.d_ratio, sh.n, sh.d_pack, rank, world)
        if use_dist:
            dist.all_reduce(sh.d_pack, op=dist.ReduceOp.SUM)  # the update's single collective (RCCL over xGMI), 16+16*N bytes
        eng.pf_apply_device(sh.d_w, sh.n, sh.d_pack, sh.d_stats, world)

    def timed(sh, steps):
        """K steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        with no_gc():
            t1 = time.perf_counter()
            for _ in range(steps):
                step(sh)
            torch.cuda.synchronize(dev)
            if use_dist:
                dist.barrier()
            el = time.perf_counter() - t1
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    main_sh = Shard(args.scaling)
    n_p = main_sh.n
    lik_mode = _lik_mode(n_p, n_s)
    strict_lik = lik_mode["replay"]

    def rewarm(seconds=0.1):
        """Side measurements follow host-only stretches (D2H of results, the CPU baseline): bring the clocks back first.
        Rank-local work only — the single-GPU entry point, NO collective: rank 0 runs its extras alone while the other ranks
        wait at the final barrier (with step() here, the all-reduce inside it hung a two-rank run:
        tests/test_gpu_bench_contract.py::test_two_ranks_through_the_launcher_on_one_gpu)."""
        def local_step():
            main_sh.d_w.copy_(main_sh.d_w0)
            eng.update_device(main_sh.d_pose, main_sh.n, main_sh.d_w, main_sh.d_stats, d_lik=main_sh.d_lik,
                              d_ratio=main_sh.d_ratio, d_beam=main_sh.d_beam if n_b else None)
        keep_gpu_warm(local_step, seconds, lambda: torch.cuda.synchronize(dev))

    # first call builds + uploads the map structures (outside every timed region)
    step(main_sh)
    torch.cuda.synchronize(dev)
    setup_s = time.time() - t0

    # exact workload counts for the algorithmic-bytes accounting (counting kernels, not timed)
    ws = eng.workload_stats(main_sh.d_pose, n_p)
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
    # one collection of the set-up's garbage NOW (before the clock ramp), survivors frozen out of later passes, collector off
    # until the line is printed: nothing below may put a ~35 ms host pause between the pre-warm and the timed steps
    gc.collect()
    gc.freeze()
    gc.disable()
    prewarm = {"ms": 0.0, "batches": 0}
    if args.prewarm_ms > 0:
        torch.cuda.synchronize(dev)
        t_pre = time.perf_counter()
        for _ in range(8):
            step(main_sh)
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
                step(main_sh)
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
        step(main_sh)
    # (bit 3 = the whole update as one launch, which stands in for the likelihood kernel up to update_small_max particles)
    eng.set_option("timing_mask", args.timing_mask | (8 if args.timing_mask & 1 else 0))
    eng.set_kernel_timing(True)
    eng.reset_kernel_time()
    elapsed = timed(main_sh, args.steps)
    lik_ms, lik_n = eng.kernel_time(capi.KERNEL_LIKELIHOOD)
    one_launch = False
    if lik_n == 0:
        lik_ms, lik_n = eng.kernel_time(capi.KERNEL_UPDATE)
        one_launch = lik_n > 0
    # Second pass of the same K steps, outside `elapsed`, every kernel group timed and the two models one after the other:
    # the beam / pf durations (each timed group costs two event records per launch, so only the roofline kernel is
    # timed inside the timed region), and the likelihood duration too when it ran concurrently with the beam kernels
    # above (overlapping event intervals say nothing about either kernel).
    overlapped = bool(n_b and n_s and args.overlap_models)
    eng.set_option("timing_mask", 31)
    eng.set_option("overlap_models", 0)
    eng.reset_kernel_time()
    coll_ms = 0.0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(args.steps):
        sh = main_sh
        sh.d_w.copy_(sh.d_w0)
        if not use_dist:
            eng.update_device(sh.d_pose, sh.n, sh.d_w, sh.d_stats, d_lik=sh.d_lik, d_ratio=sh.d_ratio,
                              d_beam=sh.d_beam if n_b else None)
            continue
        eng.measure_device(sh.d_pose, sh.n, sh.d_lik, sh.d_ratio, sh.d_beam if n_b else None)
        eng.pf_partial_device(sh.d_w, sh.d_lik, sh.d_beam if n_b else None, None, sh.d_ratio, sh.n, sh.d_pack, rank, world)
        if use_dist:
            ev0.record(stream)
            dist.all_reduce(sh.d_pack, op=dist.ReduceOp.SUM)
            ev1.record(stream)
        eng.pf_apply_device(sh.d_w, sh.n, sh.d_pack, sh.d_stats, world)
        if use_dist:
            torch.cuda.synchronize(dev)
            coll_ms += ev0.elapsed_time(ev1)
    torch.cuda.synchronize(dev)
    lik2_ms, lik2_n = eng.kernel_time(capi.KERNEL_LIKELIHOOD)
    if lik2_n == 0:
        lik2_ms, lik2_n = eng.kernel_time(capi.KERNEL_UPDATE)
    beam_ms, beam_n = eng.kernel_time(capi.KERNEL_BEAM)
    pf_ms, pf_n = eng.kernel_time(capi.KERNEL_PF)
    upd_ms, upd_n = eng.kernel_time(capi.KERNEL_UPDATE)
    eng.set_option("overlap_models", args.overlap_models)
    if overlapped or not (args.timing_mask & 1):
        lik_ms, lik_n = lik2_ms, lik2_n
        kernel_timing_pass = "second pass of the same steps (overlap_models=0, all kernel groups timed)"
    else:
        kernel_timing_pass = "likelihood: hipEvents inside the timed region; beam, pf: second pass of the same steps"
    if one_launch:
        kernel_timing_pass += "; `likelihood` = the one-launch update (likelihood + beam + pf::measure in one kernel)"
    eng.set_kernel_timing(False)

    # the other scaling mode, same K steps, same bracketing (only when there is more than one rank to tell them apart)
    other = None
    if world > 1 or args.also_other_scaling:
        other_sh = Shard("strong" if args.scaling == "weak" else "weak")
        for _ in range(max(args.warmup, 3)):
            step(other_sh)
        other_el = timed(other_sh, args.steps)
        other = {"scaling": other_sh.mode, "particles_total": other_sh.n_total, "particles_per_gpu": other_sh.n,
                 "ms_per_step": other_el / args.steps * 1e3,
                 "value": float(other_sh.n_total) * n_s * args.steps / other_el}

    if rank == 0:
        n_total = main_sh.n_total
        evals_per_step = float(n_total) * n_s
        ms_per_step = elapsed / args.steps * 1e3
        value = evals_per_step * args.steps / elapsed
        lik_avg_ms = lik_ms / max(lik_n, 1)
        stats = main_sh.d_stats.cpu().numpy()
        tiled = lik_mode["tiled"]
        group = _tiled_group(n_s, n_p, args.lik_group)
        if strict_mode == 3:
            group = min(group, 16)
        small = (not tiled) and n_s <= 32 and n_p >= 256 and args.lik_small
        if tiled:
            coop = bool(lik_coop and args.lik_index == 2)
            defer = bool(coop and int(eng.get_option("lik_defer_active")))
            kernel_name = "likelihood_tiled_kernel<%d, %d, %d, %s, %s, %s>" % (group, args.lik_index, 4 if group == 32 else 8,
                                                                               "true" if coop else "false",
                                                                               "true" if defer else "false",
                                                                               "true" if strict_mode == 3 else "false")
        elif small:
            kernel_name = "likelihood_small_kernel<"
        else:
            wide = args.lik_index == 2 and n_s > 512 and n_p <= 512
            kernel_name = "likelihood_kernel<%d, %d, false>" % (64 if n_s <= 128 else 1024 if wide else 256, args.lik_index)
            if one_launch:
                # the whole update as ONE launch (update_kernels.h): that kernel is what was timed, and what the counters are of
                kernel_name = "update_small_kernel<%d, %d, true>" % (64 if n_s <= 128 else 1024 if wide else 256,
                                                                     args.lik_index)
        # counters of the same map kind: profiles/*_C2j_* = C2 with displaced map points (--map-jitter)
        # (... and *_C2c_* / *_C5c_* = the same workload with the float sums inside the kernel, strict_order = 3)
        pmc_tag = args.workload + ("j" if args.map_jitter else "") + ("c" if strict_mode == 3 else "")
        pmc, pmc_src, pmc_refused = pmc_counters("void mcl3dl::" + kernel_name, pmc_tag)
        # the committed counters are per launch of ONE shape: use them only for a launch of that many wavefronts
        if tiled:
            expected_waves = 4 * ((n_s + 255) // 256) * ((n_p + group - 1) // group)
        elif small:
            expected_waves = None
        else:
            expected_waves = n_p * (1 if n_s <= 128 else 16 if "<1024" in kernel_name else 4)
        counters_note = pmc_refused
        if pmc and (expected_waves is None or abs(pmc.get("SQ_WAVES", 0.0) - expected_waves) > 0.01 * expected_waves):
            counters_note = ("committed counters (%s) are for a launch of %.0f wavefronts, this one has %s: not used"
                             % (pmc_src, pmc.get("SQ_WAVES", 0.0), expected_waves))
            pmc, pmc_src = None, None
        cost, cost_src = valu_costs()
        kernel_s = lik_avg_ms * 1e-3
        res, traffic = kernel_resources(pmc, kernel_s if lik_n else 0.0, cost, cost_src)
        # `bound` / `frac` are taken over the resources whose peak is a DOCUMENTED figure of the guide; fractions against
        # peaks this repository measured itself are listed next to them (`frac_vs_measured_peaks`), never as `frac`
        documented = {k: r for k, r in res.items() if r.get("peak_kind") == "documented"}
        if documented:
            bound = max(documented, key=lambda k: documented[k]["frac"])
            top = documented[bound]
        else:
            bound, top = "hbm", {"achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None}
        hbm_frac = res["hbm"]["frac"] if "hbm" in res else None
        roofline = {
            "bound": bound,
            "kernel": kernel_name,
            "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
            "frac_definition": ("largest fraction among the resources with a documented peak (hbm 8 TB/s, l2 34.5 TB/s, "
                                "valu_issue = wave64 VALU instructions/s against 1024 SIMDs x 2.4 GHz / 2 cycles per "
                                "instruction); the same launch against peaks measured here: frac_vs_measured_peaks"),
            "frac_vs_measured_peaks": {k: r["frac"] for k, r in res.items() if r.get("peak_kind") == "measured"},
            # north_star: ">= 40 % of HBM-read roofline" — answered, not dropped
            "hbm_target": {"frac": hbm_frac, "target": HBM_TARGET_FRAC,
                           "status": "not applicable: the index is built so that a scan tile's working set is L2-resident "
                                     "(TCC hit %s); HBM traffic is %s of the algorithmic bytes of SURVEY.md 8d, i.e. nothing is "
                                     "re-read from HBM — the kernel is bound by VALU issue and L2 request rate, not by HBM"
                                     % ("%.1f %%" % (100.0 * pmc["TCC_HIT_sum"] / max(pmc["TCC_HIT_sum"] + pmc.get("TCC_MISS_sum", 0.0), 1.0))
                                        if pmc and "TCC_HIT_sum" in pmc else "n/a",
                                        "%.2g" % (traffic / bytes_lik_launch) if traffic else "n/a")},
            "traffic": traffic,
            "counters_source": pmc_src,
            "counters_note": counters_note,
            "counters_box": "counters = mean per launch of the committed --pmc passes (another MI355X box than this run's: "
                            "instruction and request counts are properties of the launch, identical from box to box; the "
                            "kernel time they are divided by is this run's)",
            "avg_launch_ms": lik_avg_ms, "launches": lik_n,
            "resources": res,
            # the canonical structure of SURVEY.md §8d, for the record; NOT priced against the HBM peak (the shipped index
            # reads 68 B per evaluation — brick-table entry + one 64-byte voxel record — and reads them from L2)
            "algorithmic_bytes_per_launch": bytes_lik_launch,
            "algorithmic_bytes_per_eval": bytes_lik_launch / max(ws["evals"], 1.0),
            "algorithmic_rate_GBps": bytes_lik_launch / kernel_s / 1e9 if lik_n else None,
            # SURVEY.md 8d's own fraction, spelled out so that nobody has to recompute it: algorithmic bytes of the canonical
            # 27-cell structure / kernel time / 8 TB/s. It EXCEEDS 1 (20 at C2) because those bytes are never read: the
            # kernel fetches one pre-pruned 64-byte record per evaluation (index_bytes_per_eval) out of L2, so the HBM
            # roofline of 8d does not bind this design — `frac` above is the resource that does
            "frac_algorithmic_hbm": (bytes_lik_launch / kernel_s / 1e9 / HBM_PEAK_GBPS) if lik_n else None,
            "frac_algorithmic_hbm_note": "SURVEY 8d: (16 + 27*4 + 16*K) B per evaluation / kernel time / 8 TB/s; > 1 means the "
                                         "canonical bytes are not read at all (pre-pruned voxel records from L2), not that "
                                         "the part was exceeded",
            "k_bar": k_bar,
            "index_bytes_per_eval": 68.0 if args.lik_index == 2 else None,
            # L1 (TCP) cache-line accesses per cycle and CU: the resource that bound the kernel before the cooperative
            # fetch — kept as an observation, the counter has no documented peak
            "l1_accesses_per_cycle_per_cu": (pmc["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / (kernel_s * CLOCK_HZ)
                                             if pmc and "TCP_TOTAL_CACHE_ACCESSES_sum" in pmc and lik_n else None),
        }
        out = {
            "metric": "particle·point evals/sec; filter-update Hz @ 4096 particles × 16k-pt scan",  # BASELINE.json
            "value": value,
            "unit": "particle·point evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d particles%s x %d-pt scan, %d-pt cube map, likelihood model%s, dist_weight=(1,1,%g)"
                            % (args.workload, n_cfg, "/GPU" if args.scaling == "weak" else " in total", n_s, len(sc.map_xyz),
                               " + beam (DDA) %d rays/particle" % n_b if n_b else "", args.dist_weight_z),
                "particles_total": n_total, "particles_per_gpu": n_p, "scan_points": n_s, "beam_points": n_b,
                "map_points": int(len(sc.map_xyz)),
                "parallelism": (("TEST MODE MCL3DL_BENCH_SHARE_GPU: %d ranks on ONE GPU, collective over gloo — control flow only, "
                                 "the numbers mean nothing" % world) if share_gpu else
                                "particles sharded x%d, map+scan replicated, 1 all-reduce/update" % world if use_dist else
                                "one GPU, mcl3dl_hip_update_device (measure + pf::measure in one call, no collective)"),
                "update_hz": 1e3 / ms_per_step,
                "accumulate": ("likelihood terms added as floats INSIDE the likelihood kernel, in the engine's scan order "
                               "(strict_order = 3: bit-identical to the reference on the scan in that order), weights in an fp64 tree"
                               if strict_mode == 3 else
                               "likelihood terms and weights added as floats in the reference's order (bit-identical results)"
                               if strict_mode == 1 else
                               "likelihood terms replayed as floats in the reference's order (default for scans of <= %d and >= %d "
                               "points), weights in an fp64 tree" % (int(eng.get_option("strict_exact_max")),
                                                                     int(eng.get_option("strict_auto_min"))) if strict_lik else
                               "likelihood terms added as floats in the caller's order inside the per-particle kernel (LDS rows: "
                               "bit-identical to the reference), weights in the reference's float order up to 1024 particles"
                               if lik_mode["rows"] else
                               "fp64 tree (terms bit-identical to the reference's float terms)"),
                "lik_coop": lik_coop,
            },
            "value_definition": "value = value_device_resident at every N: inputs (map structures, ordered scan, poses, prior "
                                "weights) in HBM before the timed region — the bench contract's wording ('inputs already resident in "
                                "HBM when the timed region starts ... the PCIe-inclusive rate is never `value`'), and the one "
                                "region that is the same code path at N = 1 and N > 1, so that the driver's scaling efficiency "
                                "compares like with like. The METRIC's own region (SURVEY.md section 8d: scan upload + pose / weight "
                                "H2D + kernels + weight D2H from / to host buffers) is `value_8d` / `ms_per_step_8d` right below "
                                "(and `headline`), timed in the same run from C; it is the lower of the two and the one DESIGN.md "
                                "quotes as the update rate. (VERDICT round 5 asked for `value` := the 8d region; the bench contract "
                                "forbids a PCIe-inclusive `value`, so both stand side by side at the top level.)",
            "value_device_resident": value,
            "value_8d": None,
            "ms_per_step_8d": None,
            "roofline": roofline,
            "prewarm": prewarm,
            "kernel_timing_pass": kernel_timing_pass,
            "kernels_ms_per_step": {"likelihood": lik_avg_ms, "beam": beam_ms / max(beam_n, 1) if n_b else 0.0,
                                    "pf": pf_ms / max(args.steps, 1),
                                    # the whole update as ONE launch (<= update_small_max particles): then the three above are 0
                                    "update_one_launch": upd_ms / max(args.steps, 1),
                                    "collective": coll_ms / args.steps if use_dist else 0.0},
            "setup_seconds": setup_s,
            "index": dict(eng.index_stats(), lik_index=args.lik_index, voxel_ratio=args.cand_voxel_ratio,
                          footprint_bytes=eng.memory_footprint()),
            "result_check": {"entropy": float(stats[0]), "match_ratio_min": float(stats[1]),
                             "match_ratio_max": float(stats[2]), "restored": bool(stats[3])},
        }
        if other:
            out["other_scaling"] = other
        # self-consistency of the numbers that were timed: normalised weights sum to 1 and reproduce the entropy
        # (one rank's share of the sum when particles are sharded)
        wf = main_sh.d_w.cpu().numpy().astype(np.float64)
        out["result_check"]["weight_sum_this_rank"] = float(wf.sum())
        if world == 1:
            out["result_check"]["entropy_from_weights"] = float(-(wf[wf > 0] * np.log(wf[wf > 0])).sum())
        if n_b:
            beam_avg = beam_ms / max(beam_n, 1)
            out["beam"] = {"rays_per_s": ws["rays"] / (beam_avg * 1e-3), "dda_steps_per_s": ws["dda_steps"] / (beam_avg * 1e-3),
                           "algorithmic_GBps": bytes_beam_launch / (beam_avg * 1e-3) / 1e9, "avg_launch_ms": beam_avg,
                           "avg_launch_ms_includes": "penalty memset + beam_origin_kernel + beam_kernel + beam_finalize_kernel "
                                                     "(one timed group, run alone: overlap_models = 0)"}
            # the beam kernel's own counter-derived fractions (same pricing as `roofline`), against the kernel's share of the
            # timed group: its rocprofv3 share of beam_kernel in the group is > 95 % at these sizes
            bpmc, bsrc, bnote = pmc_counters("void mcl3dl::beam_kernel<false", pmc_tag)
            if bnote:
                out["beam"]["counters_note"] = bnote
            beam_waves = 4 * ((n_p * n_b + 255) // 256)
            if bpmc and abs(bpmc.get("SQ_WAVES", 0.0) - beam_waves) > 0.01 * beam_waves:
                bpmc = None   # counters of another launch shape
            if bpmc:
                bres, btraffic = kernel_resources(bpmc, beam_avg * 1e-3, cost, cost_src)
                for r in bres.values():
                    r.pop("note", None)
                bdoc = {k: r for k, r in bres.items() if r.get("peak_kind") == "documented"}
                if bdoc:
                    bb = max(bdoc, key=lambda k: bdoc[k]["frac"])
                    out["beam"]["roofline"] = {"bound": bb, "frac": bdoc[bb]["frac"], "traffic": btraffic,
                                               "frac_vs_measured_peaks": {k: r["frac"] for k, r in bres.items()
                                                                          if r.get("peak_kind") == "measured"},
                                               "counters_source": bsrc, "resources": bres,
                                               "algorithmic_bytes_per_launch": bytes_beam_launch}
        d_pose, d_w, d_w0, d_lik, d_ratio, d_beam, d_stats = (main_sh.d_pose, main_sh.d_w, main_sh.d_w0, main_sh.d_lik,
                                                              main_sh.d_ratio, main_sh.d_beam, main_sh.d_stats)
        if world == 1:
            # SURVEY.md section 8d's definition of one update: host buffers in, host buffers out (scan ordering + upload,
            # pose and prior-weight H2D, kernels, reduction, weight D2H) through the synchronous host entry point
            eng.set_stream(None)
            # timed from C (tools/benchloop.c: the reference's caller is C++; ctypes spends ~15 us per call converting the 19
            # arguments) on pageable arrays — every array staged through the library's page-locked block — and on arrays the
            # caller allocated with mcl3dl_hip_host_alloc (read and written in place); the Python-call figure next to them
            h_pose = np.ascontiguousarray(sc.poses, np.float32)
            h_w0 = np.ascontiguousarray(sc.weights, np.float32)
            h_lik = np.ascontiguousarray(sc.scan_lik, np.float32)
            h_beam = np.ascontiguousarray(sc.scan_beam, np.float32) if n_b else None
            h_lab = np.ascontiguousarray(sc.scan_beam_label, np.uint32) if n_b else None
            h_org = np.ascontiguousarray(sc.origins, np.float32)
            o_lik, o_ratio, o_beam = (np.zeros(n_p, np.float32) for _ in range(3))
            with no_gc():
                host_ms, per = eng.time_measure_update(h_pose, h_w0, h_w0.copy(), h_lik, h_beam, h_lab, h_org, o_lik, o_ratio,
                                                       o_beam, args.steps, warm_ms=150.0)
            pin = dict(pose=eng.host_array((n_p, 7)), w=eng.host_array(n_p), lik=eng.host_array((n_s, 3)),
                       beam=eng.host_array((n_b, 3)) if n_b else None, lab=eng.host_array(n_b, np.uint32) if n_b else None,
                       org=eng.host_array((len(h_org), 3)), o_lik=eng.host_array(n_p), o_ratio=eng.host_array(n_p),
                       o_beam=eng.host_array(n_p))
            pin["pose"][:] = h_pose
            pin["lik"][:] = h_lik
            pin["org"][:] = h_org
            if n_b:
                pin["beam"][:] = h_beam
                pin["lab"][:] = h_lab
            with no_gc():
                pinned_ms, _per = eng.time_measure_update(pin["pose"], h_w0, pin["w"], pin["lik"], pin["beam"], pin["lab"],
                                                          pin["org"], pin["o_lik"], pin["o_ratio"], pin["o_beam"], args.steps,
                                                          warm_ms=100.0)
            same_results = bool(np.array_equal(pin["o_lik"], o_lik) and np.array_equal(pin["o_ratio"], o_ratio))
            keep_gpu_warm(lambda: eng.measure_update(sc.poses, sc.weights, sc.scan_lik, sc.scan_beam, sc.scan_beam_label,
                                                     sc.origins), 0.1)
            with no_gc():
                t2 = time.perf_counter()
                for _ in range(args.steps):
                    eng.measure_update(sc.poses, sc.weights, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
                python_ms = (time.perf_counter() - t2) / args.steps * 1e3
            for a in pin.values():
                if a is not None:
                    eng.host_free(a)
            out["value_8d"] = n_p * n_s / (host_ms * 1e-3)
            out["ms_per_step_8d"] = host_ms
            fp = out["index"]["footprint_bytes"]
            index_bytes = int(fp["cand_table"] + fp["cand_start"] + fp["cand_points"])
            map_bytes = 16 * int(len(sc.map_xyz))
            out["headline"] = {
                "value_8d": out["value_8d"], "ms_per_update_8d": host_ms, "update_hz_8d": 1e3 / host_ms,
                "value_device_resident": value, "ms_per_step_device_resident": ms_per_step,
                "region_8d": "SURVEY.md section 8d: host buffers in (scan, poses, prior weights), host buffers out (weights, "
                             "likelihoods), PCIe and scan ordering included — mcl3dl_hip_measure_update timed from C",
                "likelihood_sum": out["config"]["accumulate"],
                "index_bytes": index_bytes, "map_bytes": map_bytes, "index_bytes_per_map_byte": index_bytes / max(map_bytes, 1),
            }
            roofline["frac_of_update"] = {"device_resident": lik_avg_ms / ms_per_step if lik_n else None,
                                          "update_8d": lik_avg_ms / host_ms if lik_n else None,
                                          "what": "dominant kernel's average launch time / time of one whole update"}
            out["update_8d"] = {"ms_per_update": host_ms, "value": n_p * n_s / (host_ms * 1e-3),
                                "unit": "particle·point evals/s", "update_hz": 1e3 / host_ms, "steps": args.steps,
                                "median_ms": float(np.median(per)), "min_ms": float(per.min()),
                                "overhead_over_device_resident_ms": host_ms - ms_per_step,
                                "ms_per_update_page_locked_arrays": pinned_ms,
                                "value_page_locked_arrays": n_p * n_s / (pinned_ms * 1e-3),
                                "page_locked_results_equal": same_results,
                                "ms_per_update_called_from_python": python_ms,
                                "what": "mcl3dl_hip_measure_update on host buffers, timed from C (tools/benchloop.c): scan upload + "
                                        "ordering + pose/weight H2D + kernels + weight/likelihood D2H, PCIe included (SURVEY.md "
                                        "section 8d's timed region); pageable caller arrays (ms_per_update) and arrays from "
                                        "mcl3dl_hip_host_alloc (ms_per_update_page_locked_arrays)"}
            if not args.no_extras and args.workload in ("C1", "C2", "C3") and strict_mode in (0, 2):
                # The same region for a caller that HOLDS its scan in the engine's order (mcl3dl_hip_scan_order_host; the drop-in
                # classes with MCL3DL_HIP_ENGINE_ORDER=1): no ordering launches (scan_presorted) and the reference's float
                # recurrence inside the likelihood kernel (strict_order = 3) — likelihoods bit-identical to the reference on the
                # caller's own array — next to the caller-order replay (strict_order = 1). Checked against the CPU reference below.
                held = np.ascontiguousarray(h_lik[capi.scan_order_host(h_lik)])
                eo = {}
                e_lik, e_ratio, e_beam = (np.zeros(n_p, np.float32) for _ in range(3))
                try:
                    for tag, scan_arr, so, pre in (("caller_order_replay", h_lik, 1, 0), ("engine_order_in_kernel", h_lik, 3, 0),
                                                   ("held_in_engine_order_presorted_fp64", held, 0, 1),
                                                   ("held_in_engine_order_presorted_in_kernel", held, 3, 1)):
                        eng.set_option("strict_order", so)
                        eng.set_option("scan_presorted", pre)
                        with no_gc():
                            ms_e, _ = eng.time_measure_update(h_pose, h_w0, h_w0.copy(), scan_arr, h_beam, h_lab, h_org, e_lik, e_ratio,
                                                              e_beam, args.steps, warm_ms=100.0)
                        eo[tag + "_ms"] = ms_e
                finally:
                    eng.set_option("strict_order", strict_mode)
                    eng.set_option("scan_presorted", 0)
                eo["default_ms"] = host_ms
                eo["held_scan_likelihoods"] = e_lik.copy()   # (popped again below: compared with the CPU reference on `held`)
                eo["held_scan"] = held
                eo["what"] = ("mcl3dl_hip_measure_update on host buffers timed from C, as update_8d: the float sums replayed in the "
                              "caller's order behind the kernel (strict_order 1), run inside the kernel in the engine's order (3), "
                              "and for a caller that holds its scan in that order (option scan_presorted: no ordering launches) — "
                              "the last one is bit-identical to the reference on the caller's own array")
                out["engine_order"] = eo
            eng.set_stream(stream.cuda_stream)
        if not args.no_extras:
            # the reductions that follow the update in the node (expectationBiased + max + covariance) on the
            # device-resident particles; each call ends with a D2H of a dozen scalars
            rewarm()
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
            st13[:, :7] = main_sh.host_poses
            t4 = time.perf_counter()
            for _ in range(5):
                pstep = eng.resample_begin(w_host)
                _src, _dup, n_dup = eng.resample_plan(0, 0.37 * pstep)
                ident = _identity_noise(n_dup)
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
        if world == 1 and not args.no_extras:
            # the fused device-resident call (measure + pf::measure in one C call), host wall per call back to back. Not `value`.
            # (its hipGraph-replay form measured slower for two rounds and was removed in round 5)
            fused = {}
            rewarm()
            for use_graph in (0,):
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
            rewarm()
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
            out["fused_update"] = {"ms_per_update_eager": fused["eager"],
                                   "what": "mcl3dl_hip_update_device, device-resident, plain launches"}
        if world == 1 and not args.no_cpu_baseline:
            order = None
            if strict_mode == 3:
                eng.upload_scan(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
                order = eng.scan_order(len(sc.scan_lik))
                out["result_check"]["sum_order"] = ("engine (mcl3dl_hip_scan_order): the CPU reference below was given the scan "
                                                    "in that order, the order the timed kernel summed it in")
            cb, cpu_lik, cpu_q = cpu_baseline(sc, dist_weight, args.cpu_particles, n_b, scan_order=order)
            out["cpu_baseline"] = cb
            # parity spot check of the very numbers that were timed
            n = len(cpu_lik)
            gl = d_lik[:n].cpu().numpy()
            gq = d_ratio[:n].cpu().numpy()
            out["result_check"]["max_rel_err_vs_cpu"] = float(np.max(np.abs(gl - cpu_lik) / np.maximum(np.abs(cpu_lik), 1e-30)))
            out["result_check"]["match_ratio_equal"] = bool(np.array_equal(gq, cpu_q))
            out["result_check"]["tolerance"] = ("default mode: every float term bit-identical to the reference's, summed in fp64; "
                                                "the reference sums sequentially in float, so max_rel_err_vs_cpu is the "
                                                "reference's own rounding (random walk: ~5e-6 at 16 384 points, ~1e-5 at "
                                                "65 536); --strict-order 1 reproduces the reference's float bit for bit")
            out["result_check"]["cpu_sample_particles"] = n
            eo = out.get("engine_order")
            if eo and "held_scan" in eo:
                from oracle import pyoracle as _po
                o2 = _po.Oracle("ref" if _po.available("ref") else "port")
                o2.set_map(sc.map_xyz, sc.map_label, dist_weight=dist_weight)
                o2.set_likelihood_params(_po.LikelihoodParams())
                m = min(n, 64)
                ref_lik, _q = o2.likelihood_measure(sc.poses[:m], eo["held_scan"])
                eo["bit_identical_to_cpu_reference_on_the_held_scan"] = bool(np.array_equal(eo["held_scan_likelihoods"][:m], ref_lik))
                eo["checked_particles"] = m
        else:
            out["cpu_baseline"] = None
        if "engine_order" in out:
            out["engine_order"].pop("held_scan", None)
            out["engine_order"].pop("held_scan_likelihoods", None)
        if world == 1 and not args.no_extras:
            ra = route_a(sc, dist_weight, n_b, args.route_a_reps if args.workload in ("C1", "C2", "C3") else 0)
            if ra:
                out["route_a"] = ra
        if world == 1 and not args.no_extras and args.workload in ("C1", "C2", "C3") and args.lik_index == 2:
            # a mapcloud_update of ~1 % of the map (a new surface 0.12 m in front of the walls around the
            # robot) — only the touched bricks of the candidate-voxel index are re-compiled; next to it the whole-map build
            full_ms = eng.index_stats()["build_ms"]
            tp = sc.true_pose[:3]
            range_to_pose = np.linalg.norm(sc.map_xyz - tp, axis=1)
            near = sc.map_xyz[np.argsort(range_to_pose)[:max(len(sc.map_xyz) // 100, 16)]]
            inward = (tp - near) / np.maximum(np.linalg.norm(tp - near, axis=1, keepdims=True), 1e-6)
            upd = (near + 0.12 * inward).astype(np.float32)
            t8 = time.perf_counter()
            n_map, ust = eng.map_update(upd, None, leaf=(0.1, 0.1, 0.1), stamp=77)
            first_ms = (time.perf_counter() - t8) * 1e3
            # the node replaces its update cloud every few seconds (src/mcl_3dl.cpp:141-153): the steady state is an update
            # that REPLACES the previous one — the same surface moved by a centimetre each time (the first call also pays for
            # the scratch blocks every later one recycles)
            walls, outcomes, after, steady, split_after, split_steady = [], [], [], [], [], []
            split_cloud = np.ascontiguousarray(sc.scan_lik)
            m_out, u_out = (eng.host_array((len(split_cloud), 3)) for _ in range(2))

            def small_split():
                # matched / unmatched of the scan (src/mcl_3dl.cpp:761-805): the consumer of the cell grid, which a map update
                # now merges into instead of rebuilding (host_grid_builders.h)
                t9 = time.perf_counter()
                eng.match_split_into(sc.true_pose, m_out, u_out, xyz=split_cloud)
                return (time.perf_counter() - t9) * 1e3

            small_split()

            def small_measure():
                # 64 particles against the whole scan, both models: whatever the update left to rebuild is paid here
                t9 = time.perf_counter()
                eng.measure_batch(sc.poses[:64], sc.scan_lik, sc.scan_beam if n_b else None,
                                  sc.scan_beam_label if n_b else None, sc.origins)
                return (time.perf_counter() - t9) * 1e3

            small_measure()
            for rep in range(6):
                upd_k = (near + (0.13 + 0.01 * rep) * inward).astype(np.float32)
                t8 = time.perf_counter()
                n_map, ust = eng.map_update(upd_k, None, leaf=(0.1, 0.1, 0.1), stamp=177 + rep)
                walls.append((time.perf_counter() - t8) * 1e3)
                outcomes.append(int(ust["outcome"]))
                after.append(small_measure())
                steady.append(small_measure())
                split_after.append(small_split())
                split_steady.append(small_split())
            wall_ms = float(np.median(walls))
            out["map_update"] = dict(ust, update_points=int(n_map - len(sc.map_xyz)), map_points=int(len(sc.map_xyz)),
                                     wall_ms=wall_ms, wall_ms_first=first_ms, wall_ms_each=walls, outcomes=outcomes,
                                     overflow_compactions=int(eng.get_option("cand_ovf_compactions")),
                                     first_measure_after_update_ms=float(np.median(after)),
                                     same_measure_steady_ms=float(np.median(steady)),
                                     first_match_split_after_update_ms=float(np.median(split_after)),
                                     same_match_split_steady_ms=float(np.median(split_steady)),
                                     cell_grid_merges=int(eng.get_option("lik_grid_merges")),
                                     cell_grid_rebuilds=int(eng.get_option("lik_grid_rebuilds")),
                                     dda_overlay_updates=int(eng.get_option("dda_overlay_updates")),
                                     full_build_ms=full_ms,
                                     what="mcl3dl_hip_map_update: VoxelGrid of the update + incremental index update "
                                          "(device_ms = the index part), median wall time of six updates that each replace the "
                                          "previous one; wall_ms_first = the first update (allocates the scratch blocks the "
                                          "others recycle); includes the host copy of the map; outcome 0 = incremental; "
                                          "first_measure_after_update_ms = a 64-particle measure_batch of both models right "
                                          "behind each update (it pays for whatever the update left to rebuild: nothing, when "
                                          "the DDA grid took the update as an overlay) next to the same call once more")
            eng.map_update(None, None, stamp=78)  # withdraw it again
            eng.host_free(m_out)
            eng.host_free(u_out)
        if world == 1 and not args.no_extras and args.jitter_check > 0 and args.workload in ("C2", "C3") and not args.map_jitter:
            # standing robustness figure: the same workload on a map whose points are voxel-filter centroids, not a lattice
            scj = make_config(args.workload, n_p=n_cfg, seed=12345, map_jitter=args.jitter_check, **extra_cfg)
            eng.set_map(scj.map_xyz, scj.map_label, stamp=2, dist_weight=dist_weight)
            eng.upload_scan(scj.scan_lik, scj.scan_beam, scj.scan_beam_label, scj.origins)
            dj = torch.from_numpy(scj.poses).to(dev).contiguous()
            t_warm = time.perf_counter()   # the GPU has been idle through the CPU baseline: ramp the clock up again
            while time.perf_counter() - t_warm < 0.3:
                for _ in range(20):
                    eng.measure_device(dj, n_p, d_lik, d_ratio, None)
                eng.synchronize()
            eng.set_option("timing_mask", 1)
            eng.set_kernel_timing(True)
            eng.reset_kernel_time()
            for _ in range(args.steps):
                eng.measure_device(dj, n_p, d_lik, d_ratio, None)
            jm, jn = eng.kernel_time(capi.KERNEL_LIKELIHOOD)
            eng.set_kernel_timing(False)
            out["map_jitter"] = {"jitter_m": args.jitter_check, "likelihood_ms": jm / max(jn, 1),
                                 "vs_lattice": (jm / max(jn, 1)) / lik_avg_ms if lik_avg_ms else None,
                                 "index": eng.index_stats()}
        if world == 1 and not args.no_extras and args.workload in ("C2", "C3") and not args.map_jitter and args.dist_weight_z == 1.0:
            # standing figure at the metric the reference SHIPS (dist_weight_z = 5, src/parameters.cpp:108-110; the demo's 2.0,
            # config/test_localization.yaml:5): the rescaled z axis stretches the index (3.7 x the records at z x 5). Parity at
            # these weights and sizes: tests/test_gpu_dist_weight_fullsize.py
            shipped = {}
            for wz in (5.0, 2.0):
                eng.set_map(sc.map_xyz, sc.map_label, stamp=40 + int(wz), dist_weight=(1.0, 1.0, wz))
                eng.upload_scan(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
                dw_pose = torch.from_numpy(sc.poses).to(dev).contiguous()
                t_warm = time.perf_counter()
                while time.perf_counter() - t_warm < 0.2:
                    for _ in range(20):
                        eng.measure_device(dw_pose, n_p, d_lik, d_ratio, None)
                    eng.synchronize()
                eng.set_option("timing_mask", 1)
                eng.set_kernel_timing(True)
                eng.reset_kernel_time()
                for _ in range(args.steps):
                    eng.measure_device(dw_pose, n_p, d_lik, d_ratio, None)
                wm, wn = eng.kernel_time(capi.KERNEL_LIKELIHOOD)
                eng.set_kernel_timing(False)
                st = eng.index_stats()
                shipped["z%g" % wz] = {"dist_weight": [1.0, 1.0, wz], "likelihood_ms": wm / max(wn, 1),
                                       "vs_unit_weight": (wm / max(wn, 1)) / lik_avg_ms if lik_avg_ms else None,
                                       "evals_per_s": n_p * n_s / (wm / max(wn, 1) * 1e-3) if wm else None,
                                       "records_bytes": eng.memory_footprint()["cand_start"], "build_ms": st["build_ms"],
                                       "voxels_with_overflow": st["voxels_with_overflow"]}
            out["dist_weight_shipped"] = shipped
            eng.set_map(sc.map_xyz, sc.map_label, stamp=49, dist_weight=dist_weight)
            eng.upload_scan(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        if world == 1 and not args.no_extras:
            out.update(cloud_path_extras(eng, sc, n_s, n_b, with_cpu=not args.no_cpu_baseline))
        # the in-process route (the one the reference's single process would use) on the same GPUs, next to the
        # one-process-per-GPU figures above: a group of 1 with the RCCL call in the loop on one GPU, of all N on a node
        if not args.no_extras or world > 1:
            try:
                if world == 1:
                    out["in_process_group"] = in_process_group_run(args.workload, n_cfg, extra_cfg, dist_weight,
                                                                   [0], args.steps, args.warmup)
                else:
                    out["in_process_group"] = in_process_group_child(args, world)
            except Exception as e:  # never lose the line over the side measurement
                out["in_process_group"] = {"error": str(e)[-300:]}
        line = json.dumps(out)
    if use_dist and world > 1:
        dist.barrier(group=cpu_pg)   # ranks 1.. waited on the CPU while rank 0 ran the in-process group
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0:
        print(line, flush=True)  # the one JSON line, last on stdout


if __name__ == "__main__":
    main()
