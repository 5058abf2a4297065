import sys, numpy as np, torch
sys.path.insert(0, '.')
from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_config
for wl, kw in (("C3", {}), ("C5", dict(n_p=8192))):
    sc = make_config(wl, **kw)
    eng = capi.Engine(0)
    eng.set_map(sc.map_xyz, sc.map_label); eng.set_likelihood_params(); eng.set_beam_params(num_points=len(sc.scan_beam))
    eng.upload_scan(sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
    d_pose = torch.from_numpy(sc.poses).cuda()
    ws = eng.workload_stats(d_pose, len(sc.poses))
    r = ws["rays"]
    print(wl, "rays %.3g steps/ray %.2f occupied/ray %.3f tested/ray %.3f" % (r, ws["dda_steps"]/r, ws["dda_occupied"]/r, ws["dda_tested"]/r))
    eng.close()
