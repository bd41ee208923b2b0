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


def _full_size_result():
    """Round 4's own 20 KB line (profiles/r04_bench.json.log — the one the driver could not parse) as the emitter's input,
    plus the fields later rounds added."""
    with open(os.path.join(ROOT, "profiles", "r04_bench.json.log")) as f:
        full = json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])
    assert len(json.dumps(full)) > 16000 and len(full["secondary"]) >= 12
    for i, e in enumerate(full["secondary"]):               # later rounds key their entries
        e.setdefault("key", f"entry_{i}")
    full.update(per_rank_ms_per_step=[0.1786] * 8, per_rank_value=[5.73e6] * 8, ranks_seen=8, barrier_backend="nccl",
                library_sha="0123456789abcdef")
    return full


def test_last_line_is_compact_and_complete(tmp_path, capsys):
    """The driver parses the LAST stdout line: it must be one JSON object under 4 KB that carries the contract's keys and the
    `roofline` / `cpu_baseline` objects with numbers only; the secondary entries precede it, one short line each, and the
    full objects land in bench_secondary.json."""
    import bench_line
    full = _full_size_result()
    bench_line.emit(full, str(tmp_path))
    lines = capsys.readouterr().out.splitlines()
    assert all(len(l) < bench_line.SECONDARY_LINE_LIMIT for l in lines[:-1]) and len(lines) == len(full["secondary"]) + 1
    last = lines[-1]
    assert len(last) < bench_line.COMPACT_LIMIT == 4096
    out = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["value"] == float(f"{full['value']:.6g}") and out["config"]["workload"]
    r = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_avg", "hbm_frac_measured", "frac_overlapped"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    c = out["cpu_baseline"]
    assert c["cores"] == 1 and c["kind"] == "port" and c["value"] > 0 and c["sample"] and c["host_cpu"]
    assert out["cpu_baseline_all_cores"]["cores"] >= 1 and out["pose_delta_vs_cpu"]["n_tracked_equal"] is True
    assert len(out["per_rank_ms_per_step"]) == len(out["per_rank_value"]) == out["ranks_seen"] == 8
    # no prose anywhere in the line: no *_note keys, no string longer than the workload's 160 characters

    def walk(o, key=""):
        if isinstance(o, dict):
            for k, v in o.items():
                assert k != "note" and not k.endswith("_note"), k
                walk(v, k)
        elif isinstance(o, list):
            for v in o:
                walk(v, key)
        elif isinstance(o, str):
            assert len(o) <= 160, (key, len(o))
    walk(out)
    # the secondary entries: one keyed line each, and the file holds the full objects
    keys = [json.loads(l)["secondary"] for l in lines[:-1]]
    assert len(set(keys)) == len(keys)
    with open(tmp_path / bench_line.SECONDARY_FILE) as f:
        d = json.load(f)
    assert len(d["secondary"]) == len(full["secondary"]) and "roofline" in d["headline"]
    assert out["secondary"]["entries"] == len(full["secondary"]) and out["secondary"]["file"] == bench_line.SECONDARY_FILE


def test_compact_line_sheds_extras_but_never_the_required_objects():
    import bench_line
    full = _full_size_result()
    for i in range(200):                                    # far more secondary entries than any run produces
        full["secondary"].append({"key": f"extra_entry_number_{i}", "value": 1.0 + i, "unit": "x/s"})
    line = bench_line.compact_line(full)
    out = json.loads(line)
    assert len(line) < 4096 and "roofline" in out and "cpu_baseline" in out and out["secondary"]["entries"] == len(full["secondary"])
    assert "values" not in out["secondary"]
    # a result whose required part alone cannot fit is a bug, reported as such
    full["config"]["workload"] = "x" * 100
    full["roofline"] = {k: "y" * 150 for k in bench_line.ROOFLINE_KEYS}
    full["cpu_baseline"] = {k: "z" * 120 for k in bench_line.CPU_KEYS}
    full["metric"] = "m" * 1500
    try:
        bench_line.compact_line(full)
    except AssertionError as e:
        assert "limit 4096" in str(e)
    else:
        raise AssertionError("an oversized line must not be printed")


def test_rank_checks_aggregate_to_one_verdict():
    """N > 1: every rank's own parity row [max_rad, max_m, n_tracked_equal, iterations_equal, pairs_checked] is gathered and
    folded by bench.aggregate_rank_checks: the job passes only when EVERY rank checked pairs and stayed inside north_star's
    1e-4 rad / 1e-4 m with equal n_tracked."""
    import bench
    good = [[2e-15, 1e-15, 1.0, 1.0, 32.0], [7e-16, 3e-15, 1.0, 1.0, 32.0]]
    pd, ok = bench.aggregate_rank_checks(good)
    assert ok and pd["ranks_checked"] == 2 and pd["pairs_checked"] == 64 and pd["max_rad"] == 2e-15 and pd["max_m"] == 3e-15
    assert pd["n_tracked_equal"] and pd["iterations_equal"] and len(pd["per_rank_max_rad"]) == 2
    for bad in ([[2e-15, 1e-15, 1.0, 1.0, 32.0], [3e-4, 1e-15, 1.0, 1.0, 32.0]],           # rank 1 over the tolerance
                [[2e-15, 1e-15, 1.0, 1.0, 32.0], [1e-15, 1e-15, 0.0, 1.0, 32.0]],           # rank 1: another n_tracked
                [[2e-15, 1e-15, 1.0, 1.0, 32.0], [float("inf"), 0.0, 1.0, 1.0, 32.0]],      # rank 1: NaN poses
                [[2e-15, 1e-15, 1.0, 1.0, 32.0], [0.0, 0.0, 1.0, 1.0, 0.0]]):               # rank 1 checked nothing
        assert not bench.aggregate_rank_checks(bad)[1]
    # the compact line keeps the verdict and the count of ranks behind it
    import bench_line
    line = bench_line.compact({"metric": "m", "value": 1.0, "pose_delta_vs_cpu": pd})
    assert line["pose_delta_vs_cpu"]["ranks_checked"] == 2 and line["pose_delta_vs_cpu"]["pairs_checked"] == 64


def test_committed_profiles_describe_the_sources_in_the_tree():
    """profiles/r06_bench_pmc.json (counter traffic, FP64 mix, kernel durations: what bench.py prints as roofline.traffic / fp64)
    names the binary it was measured with and the sources + flags that binary was built from. The round's last commit must not
    leave the two apart: a source edited after the last profile run would silently null those fields in the driver's line."""
    from dsdtm_amd.csrc import build as hip_build
    with open(os.path.join(ROOT, "profiles", "r06_bench_pmc.json")) as f:
        d = json.load(f)
    assert d["profile_source_sha"] == hip_build.source_sha(), "library sources changed after the committed profile run: re-run tools/profile.sh + merge_profiles.py"
