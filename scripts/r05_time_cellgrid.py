"""Where the first match_split after a map update spends its time (cell grid: merge of the update into the base map's grid)."""
import time
import numpy as np
from mcl_3dl_amd import capi
from mcl_3dl_amd.synthetic import make_config

sc = make_config("C2", seed=12345)
eng = capi.Engine(0)
eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 1.0))
eng.set_likelihood_params()
tp = sc.true_pose[:3]
near = sc.map_xyz[np.argsort(np.linalg.norm(sc.map_xyz - tp, axis=1))[:len(sc.map_xyz) // 100]]
inward = (tp - near) / np.maximum(np.linalg.norm(tp - near, axis=1, keepdims=True), 1e-6)
cloud = np.ascontiguousarray(sc.scan_lik)
m_out, u_out = (eng.host_array((len(cloud), 3)) for _ in range(2))


def split():
    t = time.perf_counter()
    eng.match_split_into(sc.true_pose, m_out, u_out, xyz=cloud)
    return (time.perf_counter() - t) * 1e3


print("first split (builds the base grid): %.3f ms, device %.3f ms, wall of the build %.3f ms" % (split(), eng.get_option("lik_grid_build_ms"), eng.get_option("lik_grid_build_wall_ms")))
print("steady split: %.3f %.3f" % (split(), split()))
for rep in range(6):
    upd = (near + (0.12 + 0.01 * rep) * inward).astype(np.float32)
    t = time.perf_counter()
    eng.map_update(upd, None, leaf=(0.1, 0.1, 0.1), stamp=10 + rep)
    t_upd = (time.perf_counter() - t) * 1e3
    a = split()
    print("update %.3f ms; split after %.3f ms (grid: device %.3f, wall %.3f; merges %d rebuilds %d); steady %.3f" % (
        t_upd, a, eng.get_option("lik_grid_build_ms"), eng.get_option("lik_grid_build_wall_ms"),
        eng.get_option("lik_grid_merges"), eng.get_option("lik_grid_rebuilds"), split()))
