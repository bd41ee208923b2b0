"""bench.py's bookkeeping that needs no GPU: the algorithmic byte counts of SURVEY.md §8(d), the lookup of the
committed rocprofv3 PMC summary behind roofline.traffic, and the host-thread census behind cpu_baseline_all_cores."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_are_the_surveys_figures():
    # C2 833 392 B, C3 873 292 B, C5 3 378 292 B per alignment
    assert bench.algorithmic_bytes(640, 480, 4, 300) == 833392
    assert bench.algorithmic_bytes(640, 480, 4, 1000) == 873292
    assert bench.algorithmic_bytes(1280, 960, 4, 2000) == 3378292


def test_traffic_lookup_reads_the_committed_pmc_summary(monkeypatch):
    path = os.path.join(ROOT, bench.PMC_SUMMARY)
    assert os.path.exists(path), "profiles/ must carry the PMC summary bench.py quotes"
    with open(path) as f:
        d = json.load(f)
    # the summary's numbers are only quoted for the build they were taken with: the binary's hash, or (hipcc's objects
    # are not reproducible byte for byte) the hash of the sources and flags of an up-to-date library
    monkeypatch.setattr(bench, "library_sha", lambda: "0" * 16)
    monkeypatch.setattr(bench, "library_source_sha", lambda: "1" * 16)
    assert bench.pmc_traffic("sparse_align_reg_kernel", 1024 * 833392) is None and bench.pmc_summary() == {}
    monkeypatch.setattr(bench, "library_source_sha", lambda: d["profile_source_sha"])
    assert bench.pmc_traffic("sparse_align_reg_kernel", 1024 * 833392) is not None
    monkeypatch.setattr(bench, "library_source_sha", lambda: None)              # library stale against its sources
    assert bench.pmc_summary() == {}
    monkeypatch.setattr(bench, "library_sha", lambda: d["profile_binary_sha"])
    rows = {(t["case"], int(t["algorithmic_bytes_per_launch"])): t for t in d["hbm_traffic_per_launch"]}
    main = rows[("solo", 1024 * 833392)]
    t = bench.pmc_traffic("sparse_align_reg_kernel", 1024 * 833392)
    assert t == main["fetch_bytes_gfx950_corrected"] + main["write_bytes"]
    assert main["fetch_bytes_gfx950_corrected"] == 2.0 * main["fetch_bytes_raw"]          # the guide's gfx950 correction
    assert 0.3 < t / (1024 * 833392) < 1.5                                                 # no wasted re-reads on the main kernel
    assert bench.pmc_traffic("sparse_align_reg_kernel", 12345) is None                     # another launch size: no number
    assert bench.pmc_traffic("pyrdown", 2048 * 504000) is not None


def test_usable_cpus_is_positive_and_bounded_by_the_machine():
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_fp64_block_reads_the_committed_counter_pass(monkeypatch):
    """The second roofline of the bench line (FP64 vector fraction) comes from the committed
    SQ_INSTS_VALU_*_F64 pass and only for the BASELINE workload."""
    import types
    import bench
    with open(os.path.join(ROOT, bench.PMC_SUMMARY)) as f:
        monkeypatch.setattr(bench, "library_sha", lambda sha=json.load(f)["profile_binary_sha"]: sha)
    a = types.SimpleNamespace(pairs=1024, patches=300, width=640, height=480, levels=4, iters=10)
    b = bench.fp64_block(a, 0.2)
    assert b["flops_per_launch"] is not None and 1e9 < b["flops_per_launch"] < 1e10
    assert abs(b["achieved"] - b["flops_per_launch"] / 0.2e-3 / 1e12) < 1e-9 and 0 < b["frac"] < 1
    a.patches = 1000
    assert bench.fp64_block(a, 0.2)["achieved"] is None
