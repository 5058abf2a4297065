import numpy as np, torch
from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_scene
sc = make_scene(n=91, n_p=128, n_s=600, n_b=0, seed=5)
e = capi.Engine(0)
e.set_map(sc.map_xyz, sc.map_label, stamp=1)
e.upload_scan(sc.scan_lik, None, None, sc.origins)
dev = torch.device("cuda:0")
d_pose = torch.from_numpy(sc.poses).to(dev); d_w = torch.from_numpy(sc.weights).to(dev); d_stats = torch.zeros(4, device=dev)
torch.cuda.synchronize()
for i in range(4):
    e.update_device(d_pose, 128, d_w, d_stats); e.synchronize(); print(i, e.graph_stats())
