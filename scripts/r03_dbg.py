import sys, numpy as np, torch
sys.path.insert(0, ".")
from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_scene
sc = make_scene(n=91, n_p=4400, n_s=2000, n_b=256, seed=77)
engine = capi.Engine(0)
n_p, n_s, n_b = int(sys.argv[1]), 96, int(sys.argv[2])
engine.set_option("overlap_min_rays", float(sys.argv[3]))
engine.set_map(sc.map_xyz, sc.map_label, stamp=8104, dist_weight=(1.0, 1.0, 1.0))
engine.set_likelihood_params()
engine.set_beam_params(num_points=n_b)
engine.upload_scan(sc.scan_lik[:n_s], sc.scan_beam[:n_b] if n_b else None, sc.scan_beam_label[:n_b] if n_b else None, sc.origins)
dev = torch.device("cuda", 0)
d_pose = torch.from_numpy(np.ascontiguousarray(sc.poses[:n_p])).to(dev)
w0 = torch.full((n_p,), 1.0 / n_p, device=dev)
for tag, one, graph in (("graph-split", 0, 1),):
    engine.set_option("update_small", one)
    engine.set_option("use_graph", graph)
    d_w = w0.clone()
    d_lik, d_ratio, d_beam = (torch.empty(n_p, device=dev) for _ in range(3))
    d_stats = torch.zeros(4, device=dev)
    for it in range(8):
        d_w.copy_(w0)
        torch.cuda.synchronize()
        print(tag, it, "enqueue", flush=True)
        engine.update_device(d_pose, n_p, d_w, d_stats, d_lik=d_lik, d_ratio=d_ratio, d_beam=d_beam)
        print(tag, it, "sync", flush=True)
        engine.synchronize()
    print(tag, "ok", engine.graph_stats(), flush=True)
