"""One-off (GPU box): parity at extreme likelihood radii / grid sizes against the oracle, strict order bit for bit."""
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, ".")
from mcl_3dl_amd import capi  # noqa: E402
from mcl_3dl_amd.synthetic import make_scene  # noqa: E402
from oracle import pyoracle  # noqa: E402

kind = "ref" if pyoracle.available("ref") else "port"
sc = make_scene(n=61, n_p=40, n_s=400, n_b=16, seed=3)
eng = capi.Engine(0)
eng.set_map(sc.map_xyz, sc.map_label, stamp=1, dist_weight=(1.0, 1.0, 5.0))
eng.set_option("strict_order", 1)
bad = 0
for r, flat in ((2.0, 0.3), (1.0, 0.0), (0.01, 0.001), (5.0, 4.0)):
    for dda in (0.05, 1.5):
        eng.set_likelihood_params(r, flat, 1.0)
        eng.set_beam_params(dda_grid_size=dda, num_points=16, hit_range=0.5)
        o = pyoracle.Oracle(kind, chunk_length=20.0, max_search_radius=max(r, 0.4))
        o.set_map(sc.map_xyz, sc.map_label, dist_weight=(1.0, 1.0, 5.0))
        o.set_likelihood_params(pyoracle.LikelihoodParams(match_dist_min=r, match_dist_flat=flat, match_weight=1.0))
        o.set_beam_params(pyoracle.BeamParams(dda_grid_size=dda, num_points=16, hit_range=0.5))
        lik, ratio, beam = eng.measure_batch(sc.poses, sc.scan_lik, sc.scan_beam, sc.scan_beam_label, sc.origins)
        wl, wq = o.likelihood_measure(sc.poses, sc.scan_lik)
        wb, _ = o.beam_measure(sc.poses, sc.scan_beam, sc.scan_beam_label, sc.origins)
        ok = np.array_equal(lik, wl) and np.array_equal(ratio, wq) and np.array_equal(beam, wb)
        st = eng.index_stats()
        print("r %.2f flat %.3f dda %.2f: %s  (candidates %d, bricks %d, build %.1f ms, max lik %.3f)" %
              (r, flat, dda, "OK" if ok else "MISMATCH", st["candidates"], st["bricks"], st["build_ms"], float(lik.max())))
        bad += 0 if ok else 1
print("mismatches:", bad)
