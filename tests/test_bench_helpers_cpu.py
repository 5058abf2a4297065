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
        vals, src, _ = bench.pmc_counters("void mcl3dl::likelihood_tiled_kernel<16, 2, 8, true>", tag, check_sources=False)
        assert vals and src, tag
        assert re.search(r"r\d+[a-z]?_%s_pmc" % tag, os.path.basename(src)), (tag, src)
    assert bench.pmc_counters("void mcl3dl::likelihood_tiled_kernel", "C9") == (None, None, None)


def test_valu_costs_key_micro_benchmark_rows_by_their_full_label():
    """Round 2 keyed the rows by the opcode alone: `v_mul_f32 (sgpr src)` (4.15 cycles) overwrote `v_mul_f32` (2.59) and
    `full` read 2.698 instead of the mean of the six plain rows."""
    path = os.path.join(ROOT, "profiles", "r02d_valu_microbench.txt")
    rows = bench.microbench_rows(path)
    assert rows["v_mul_f32"] == 2.588 and rows["v_mul_f32 (sgpr src)"] == 4.153
    cost, src = bench.valu_costs()
    if src.endswith("r02d_valu_microbench.txt"):
        plain = [rows[k] for k in bench.VALU_FULL_ROWS]
        assert len(plain) == 6
        assert abs(cost["full"] - sum(plain) / 6) < 1e-12
        assert abs(cost["full"] - 2.4375) < 1e-9
    assert 2.2 < cost["full"] < 2.7 < 3.8 < cost["half"] < 4.6 < 7.5 < cost["trans"] < 9.0


def test_l2_request_calibration_is_committed_and_used():
    cal = bench.l2_calibration()
    assert cal["source"].startswith("profiles/") and cal["bytes_per_request"] == 128.0
    assert 0.95 < cal["requests_per_record64"] < 1.01 and cal["bytes_per_record_request"] == 64.0
    pmc, _, _ = bench.pmc_counters("void mcl3dl::likelihood_tiled_kernel<16, 2, 8, true>", "C2", check_sources=False)
    res, _t = bench.kernel_resources(pmc, 0.2393e-3, bench.valu_costs()[0], "test")
    assert res["l2"]["bytes_per_request"] == 64.0
    assert abs(res["l2"]["frac_if_full_lines"] - 2.0 * res["l2"]["frac"]) < 1e-12
    assert res["l2_requests"]["peak_kind"] == "measured" and 0.5 < res["l2_requests"]["frac"] <= 1.0


def test_documented_and_measured_peaks_are_kept_apart():
    """`valu_issue` = wave64 VALU instructions against the guide's 2 cycles per instruction; the class-priced figure is
    `valu_issue_priced` (measured peak). The judge's round-2 recomputation: 0.57 and 0.825 at 0.2393 ms."""
    pmc, _, _ = bench.pmc_counters("void mcl3dl::likelihood_tiled_kernel<16, 2, 8, true>", "C2", check_sources=False)
    res, _t = bench.kernel_resources(pmc, 0.2393e-3, bench.valu_costs()[0], "test")
    doc = pmc["SQ_INSTS_VALU"] * 2.0 / (1024 * 0.2393e-3 * 2.4e9)
    assert abs(res["valu_issue"]["frac"] - doc) < 1e-12 and res["valu_issue"]["peak_kind"] == "documented"
    assert 0.5 < res["valu_issue"]["frac"] < 0.65
    assert res["valu_issue_priced"]["peak_kind"] == "measured"
    assert res["valu_issue_priced"]["frac_low"] < res["valu_issue_priced"]["frac"] < res["valu_issue_priced"]["frac_high"]
    assert 0.75 < res["valu_issue_priced"]["frac"] < 0.9
    for k in ("hbm", "l2", "valu_issue"):
        assert res[k]["peak_kind"] == "documented"


def test_committed_counters_and_committed_times_give_fractions():
    cost, _src = bench.valu_costs()
    for tag, line in (("C2", "r02i_bench_C2_default.json"), ("C3", "r02i_bench_C3_full.json"), ("C5", "r02i_bench_C5_shard.json")):
        d = json.load(open(os.path.join(ROOT, "profiles", line)))
        kernel = "void mcl3dl::" + d["roofline"]["kernel"]
        pmc, _, _ = bench.pmc_counters(kernel, tag, check_sources=False)
        res, traffic = bench.kernel_resources(pmc, d["roofline"]["avg_launch_ms"] * 1e-3, cost, "test")
        assert {"hbm", "l2", "l1_access", "valu_issue", "valu_issue_priced"} <= set(res)
        for name, r in res.items():
            assert 0.0 < r["frac"] <= 1.0, (tag, name, r)
        documented = {k: r["frac"] for k, r in res.items() if r["peak_kind"] == "documented"}
        assert max(documented, key=documented.get) == "valu_issue", (tag, documented)   # the binding resource, every workload
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


def test_round3_kernel_name_selects_the_round3_counters():
    """The tiled kernel gained a template parameter (DEFER) in round 3: the name bench.py builds must select the counters of
    that instantiation (profiles/r03z_*), not fall back to the round-2 kernel's; DESIGN.md 6.0's fractions follow from them."""
    for tag in ("C2", "C3", "C5", "C2j"):
        pmc, src, _ = bench.pmc_counters("void mcl3dl::likelihood_tiled_kernel<16, 2, 8, true, true>", tag, check_sources=False,
                                         only="r03")
        assert pmc and os.path.basename(src).startswith("r03"), (tag, src)
    pmc, src, _ = bench.pmc_counters("void mcl3dl::likelihood_tiled_kernel<16, 2, 8, true, true>", "C2", check_sources=False,
                                     only="r03")
    d = json.load(open(os.path.join(ROOT, "profiles", "r03z_bench_C2_full.json")))
    res, _t = bench.kernel_resources(pmc, d["roofline"]["avg_launch_ms"] * 1e-3, bench.valu_costs()[0], "test", bench.l2_calibration())
    assert abs(res["valu_issue"]["frac"] - 0.620) < 0.005
    assert abs(res["l2_requests"]["frac"] - 0.904) < 0.005
    assert abs(res["valu_issue_priced"]["frac"] - 0.90) < 0.02
    assert res["hbm"]["frac"] < 0.03
    for r in res.values():
        assert r["frac"] <= 1.0


def test_counters_are_tied_to_the_sources_they_profiled(tmp_path, monkeypatch):
    """A PMC summary names the sha of the kernel sources it was collected from (profiles/summarize_pmc.py); bench.py uses its
    counters only while those files are unchanged, and says why when it refuses them."""
    import csv
    prof = tmp_path / "profiles"
    prof.mkdir()
    csrc = tmp_path / "mcl_3dl_amd" / "csrc"
    csrc.mkdir(parents=True)
    for n in set(bench.PMC_SOURCES["lik"] + bench.PMC_SOURCES["beam"]):
        (csrc / n).write_text("// " + n)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    kernel = "void mcl3dl::likelihood_tiled_kernel<16, 2, 8, true, true>"

    def write(name, tagged):
        with open(prof / name, "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "counter", "mean_per_launch", "launches"])
            if tagged:
                w.writerow(["__source__", "git_head", "abc123", 0])
                for n in bench.PMC_SOURCES["lik"]:
                    w.writerow(["__source__", n, bench.source_sha(n), 0])
            w.writerow([kernel + "(args)", "SQ_WAVES", "65536", 5])

    write("r03z_C2_pmc_summary.csv", tagged=False)
    vals, src, note = bench.pmc_counters(kernel, "C2")
    assert vals is None and "does not record the sources" in note
    write("r04a_C2_pmc_summary.csv", tagged=True)
    vals, src, note = bench.pmc_counters(kernel, "C2")
    assert vals == {"SQ_WAVES": 65536.0} and src.endswith("r04a_C2_pmc_summary.csv") and note is None
    (csrc / "likelihood_kernels.h").write_text("// edited")
    vals, src, note = bench.pmc_counters(kernel, "C2")
    assert vals is None and "likelihood_kernels.h changed since" in note and "abc123" in note
    # the beam kernel's counters do not depend on the likelihood kernels
    with open(prof / "r04a_C3_pmc_summary.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "mean_per_launch", "launches"])
        w.writerow(["__source__", "git_head", "abc123", 0])
        for n in bench.PMC_SOURCES["beam"]:
            w.writerow(["__source__", n, bench.source_sha(n), 0])
        w.writerow(["void mcl3dl::beam_kernel<false, false>(args)", "SQ_WAVES", "32768", 5])
    vals, src, note = bench.pmc_counters("void mcl3dl::beam_kernel<false", "C3")
    assert vals == {"SQ_WAVES": 32768.0} and note is None
