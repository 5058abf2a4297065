"""bench.py's counter plumbing, on the CPU: which committed PMC summary feeds the roofline block of which workload, and that
the fractions it derives from the committed counters and the committed kernel times are fractions (<= 1)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_pmc_summaries_are_selected_by_exact_workload_tag():
    """A profile of C2 on a jittered map (tag C2j) once leaked into the lattice C2 line (l2 frac 1.07): the tag must match
    exactly, and a jittered run takes the C2j counters."""
    for tag in ("C2", "C3", "C5", "C2j"):
        vals, src = bench.pmc_counters("void mcl3dl::likelihood_tiled_kernel<16, 2, 8, true>", tag)
        assert vals and src, tag
        assert re.search(r"r\d+[a-z]?_%s_pmc" % tag, os.path.basename(src)), (tag, src)
    assert bench.pmc_counters("void mcl3dl::likelihood_tiled_kernel", "C9") == (None, None)


def test_committed_counters_and_committed_times_give_fractions():
    cost, _src = bench.valu_costs()
    for tag, line in (("C2", "r02i_bench_C2_default.json"), ("C3", "r02i_bench_C3_full.json"), ("C5", "r02i_bench_C5_shard.json")):
        d = json.load(open(os.path.join(ROOT, "profiles", line)))
        kernel = "void mcl3dl::" + d["roofline"]["kernel"]
        pmc, _ = bench.pmc_counters(kernel, tag)
        res, traffic = bench.kernel_resources(pmc, d["roofline"]["avg_launch_ms"] * 1e-3, cost, "test")
        assert {"hbm", "l2", "l1_access", "valu_issue"} <= set(res)
        for name, r in res.items():
            assert 0.0 < r["frac"] <= 1.0, (tag, name, r)
        assert res["valu_issue"]["frac"] == max(r["frac"] for r in res.values())   # the binding resource, every workload
        assert traffic and traffic < 0.05 * d["roofline"]["algorithmic_bytes_per_launch"]   # cache-resident by construction


def test_bench_lines_of_the_profiled_shapes_keep_their_fractions_below_one():
    """(Lines of OTHER shapes written before bench.py compared the counters' wavefront count with the launch's carry
    fractions priced with the wrong counters; from r02j on such a line has no `resources`.)"""
    for name in ("r02i_bench_C2_default.json", "r02i_bench_C2_noextras.json", "r02i_bench_C3_full.json",
                 "r02i_bench_C5_shard.json", "r02h_bench_C2_full.json", "r02h_bench_C3_full.json"):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert d["roofline"]["resources"], name
        for res, r in d["roofline"]["resources"].items():
            assert r["frac"] <= 1.0, (name, res)
        if "roofline" in d.get("beam", {}):
            for res, r in d["beam"]["roofline"]["resources"].items():
                assert r["frac"] <= 1.0, (name, "beam", res)
