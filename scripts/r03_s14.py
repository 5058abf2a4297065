"""Candidate pruning by sub-box domination (cand_refine) on the jittered C2 map: overflow fraction, index build time,
likelihood kernel time — and that the results do not move (bit-identical to cand_refine = 1)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_config
eng = capi.Engine(0)
dev = torch.device("cuda", 0)
for jitter in (0.045, 0.0):
    sc = make_config("C2", map_jitter=jitter) if jitter else make_config("C2")
    d_pose = torch.from_numpy(sc.poses).to(dev); n_p = len(sc.poses)
    d_l, d_r = (torch.empty(n_p, device=dev) for _ in range(2))
    ref = None
    for parts in (0, 4):
        for refine in (1, 2, 3):
            eng.set_option("cand_record_parts", parts)
            eng.set_option("cand_refine", refine)
            eng.set_map(sc.map_xyz, sc.map_label, stamp=int(1000 * jitter) + 10 * parts + refine, dist_weight=(1, 1, 1)); eng.set_likelihood_params()
            eng.upload_scan(sc.scan_lik, None, None, sc.origins)
            for _ in range(30):
                eng.measure_device(d_pose, n_p, d_l, d_r, None)
            eng.synchronize()
            eng.set_option("timing_mask", 1); eng.set_kernel_timing(True); eng.reset_kernel_time()
            for _ in range(20):
                eng.measure_device(d_pose, n_p, d_l, d_r, None)
            ms, n = eng.kernel_time(capi.KERNEL_LIKELIHOOD)
            eng.set_kernel_timing(False)
            st = eng.index_stats()
            lik = d_l.cpu().numpy().copy(); ratio = d_r.cpu().numpy().copy()
            if ref is None:
                ref = (lik, ratio)
            same = np.array_equal(ref[0], lik) and np.array_equal(ref[1], ratio)
            print("jitter %.3f parts %d refine %d: lik %.4f ms  build %.1f ms  ratio %.2f parts %s  cand/voxel %.2f  overflow %.3f  over8 %.4f  identical %s"
                  % (jitter, parts, refine, ms / max(n, 1), st["build_ms"], st["voxel_ratio"], st["record_parts"],
                     st["candidates"] / max(st["voxels_with_candidates"], 1), st["voxels_with_overflow"] / max(st["voxels_with_candidates"], 1),
                     st.get("voxels_over8", 0) / max(st["voxels_with_candidates"], 1), same), flush=True)
eng.set_option("cand_refine", 1); eng.set_option("cand_record_parts", 0)
