"""Folds the rocprofv3 output of tools/profile.sh (per case: kernel-trace stats, kernel trace, one --pmc pass per
counter group) into the summaries kept under profiles/:
    <out>/r06_kernel_stats.csv     per case and kernel: calls, average / min / max duration
    <out>/r06_bench_pmc.json       "profile_binary_sha": sha256 (16 hex digits) of the libdsdtm_amd.so the passes ran with —
                                   bench.py reports these numbers only while it loads that same binary;
                                   per case and kernel: counters per dispatch + "hbm_traffic_per_launch" entries (what
                                   bench.py's roofline.traffic reads): FETCH_SIZE and WRITE_SIZE are KiB per dispatch;
                                   FETCH_SIZE x 2 is the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md
                                   (128-B requests tallied at 64 B);
                                   "overlap": case "main" (several launch streams): begin/end of every dispatch of the
                                   alignment kernel -> launches in flight, a launch's own duration, union span per launch
Usage: python tools/summarize_profile.py gpurun_out/<dir>"""
import collections, csv, glob, hashlib, json, os, sys

src = sys.argv[1]
ALG = {   # case -> (kernel substring, algorithmic bytes per launch, dispatches per launch)
    "solo": ("sparse_align_reg_kernel", 1024 * 833392, 1),
    "n1000": ("sparse_align", 1024 * 873292, 1),
    "n2000": ("sparse_align", 256 * 3378292, 1),
    "kernels": ("pyrdown_kernel", 2048 * 504000, 3),
    # Optimizer::PoseOptimization, 4096 frames x 200 features per launch (tools/pose_opt_bench.py 4096 200 = bench.py's entry):
    # per feature bearing 24 + map point 24 + level 4 + flag 1 in, residual norm 8 out; per frame the pose in and out + summary
    "poseopt": ("pose_opt_", 4096 * (200 * 61 + 2 * 96 + 64), 1),
}
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(REPO, "dsdtm_amd", "csrc", "libdsdtm_amd.so"), "rb") as f:
    sha = hashlib.sha256(f.read()).hexdigest()[:16]
sys.path.insert(0, REPO)
from dsdtm_amd.csrc import build as hip_build
stats_rows, pmc = [], {"round": 6, "profile_binary_sha": sha, "profile_source_sha": hip_build.source_sha(), "command": "tools/profile.sh (see the file for every command line)", "cases": {},
                       "hbm_traffic_per_launch": [], "fp64_per_launch": [], "overlap": {}}
for case in sorted(os.listdir(src)):
    d = os.path.join(src, case)
    if not os.path.isdir(d):
        continue
    for f in glob.glob(f"{d}/trace/**/*_kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "dsdtm" not in r["Name"]:
                continue                              # torch's data-generation kernels are not ours to report
            if case not in ("track", "trackone", "poseopt", "kernels", "secondary") and "sparse_align" not in r["Name"]:
                continue                              # bench cases: the kernel of record only (pyramids there are set-up)
            stats_rows.append(dict(case=case, kernel=r["Name"], calls=r["Calls"], avg_ns=r["AverageNs"], min_ns=r["MinNs"],
                                   max_ns=r["MaxNs"], total_ns=r["TotalDurationNs"], percent=r["Percentage"]))
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    launch = {}
    for f in glob.glob(f"{d}/pmc_*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dsdtm" not in k:
                continue
            ctr[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            launch[k] = dict(grid=r["Grid_Size"], workgroup=r["Workgroup_Size"], lds_bytes=r["LDS_Block_Size"],
                             scratch=r["Scratch_Size"], vgpr_count_field=r["VGPR_Count"])
    if ctr:
        pmc["cases"][case] = {k: dict(launch=launch[k], counters={c: dict(dispatches=len(v), mean_per_dispatch=sum(v) / len(v))
                                                                      for c, v in sorted(cs.items())}) for k, cs in ctr.items()}
    for k, cs in ctr.items():
        if all(c in cs for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")):
            m = {c: sum(cs[c]) / len(cs[c]) for c in cs if c.startswith("SQ_INSTS_VALU_")}
            # wave-level instruction counts; a wave instruction is 64 lane operations, an FMA two flops (inactive lanes
            # — 300 patches on 320 lanes, the solver wave's uniform arithmetic — are counted: an upper bound of ~7 %)
            flops = 64.0 * (m["SQ_INSTS_VALU_ADD_F64"] + m["SQ_INSTS_VALU_MUL_F64"] + 2.0 * m["SQ_INSTS_VALU_FMA_F64"] + m["SQ_INSTS_VALU_TRANS_F64"])
            if flops > 0:
                pmc["fp64_per_launch"].append(dict(case=case, kernel=k, wave_instructions=m, fp64_flops_per_launch=flops))
    if case == "secondary":
        # kernel durations of the driver line's secondary entries, told apart by grid size (the stats file averages all sizes)
        dur = collections.defaultdict(list)
        for f in glob.glob(f"{d}/trace/**/*_kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "dsdtm" in r["Kernel_Name"]:
                    g = int(r.get("Grid_Size") or (int(r.get("Grid_Size_X", 0)) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))))
                    dur[(r["Kernel_Name"], g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        pmc["secondary_kernel_durations"] = [dict(kernel=k, grid_work_items=g, calls=len(v), avg_ns=sum(v) / len(v), min_ns=min(v))
                                             for (k, g), v in sorted(dur.items())]
        # Entries of the driver line that are made of several kernels / of kernels that also run at other sizes: dispatches are
        # told apart by their grid size (work-items), traffic = sum over the kernels of ONE call of the entry.
        by_grid = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"{d}/pmc_*/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "dsdtm" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    by_grid[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        def one(sub, grid):
            for (k, g), cs in by_grid.items():
                if sub in k and g == grid and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                    return k, sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024.0, sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024.0
            return None
        def entry(name, parts, alg):
            got = [one(sub, grid) for sub, grid in parts]
            if any(g is None for g in got):
                print("secondary: no dispatches for", name, parts)
                return
            fetch, write = sum(g[1] for g in got), sum(g[2] for g in got)
            pmc["hbm_traffic_per_launch"].append(dict(
                case=case, kernel=name, dispatches_per_launch=len(parts), algorithmic_bytes_per_launch=alg, fetch_bytes_raw=fetch,
                fetch_bytes_gfx950_corrected=2.0 * fetch, write_bytes=write, traffic_over_algorithmic=(2.0 * fetch + write) / alg))
        M, Mm, nfm, nfr = 262144, 64 * 800, 64, 256
        pyr5 = 640 * 480 + 320 * 240 + 160 * 120 + 80 * 60 + 40 * 30
        # (round 4: Align2D runs 16 features per 256-thread group, the warp prelude 16 candidates per 128-thread group in batches)
        entry("align2d_rows_kernel", [("align2d_rows_kernel", M // 16 * 256)], M * 197 + 1280 * 960)
        # (round 5: FindMatchDirect is ONE kernel, 16 candidates per 256-thread group, the warped patches stay in LDS)
        entry("match_kernel", [("match_kernel", (Mm + 15) // 16 * 256)],
              Mm * (4 + 4 + 8 + 4 + 24 + 24 + 16 + 16 + 4 + 1) + 2 * nfm * 640 * 480)
        # Tracking's own constructor arguments (5 levels, cap 8, 190 patches): sparse_align_reg_kernel<3, 3> = 3 slots of 3 + 1 waves,
        # one persistent workgroup per CU (round 6: its `traffic` was null — no entry here)
        entry("sparse_align_reg_kernel<3, 3, false>", [("sparse_align_reg_kernel<3, 3, false>", 256 * 3 * 4 * 64)], 1024 * (2 * pyr5 + 190 * 57 + 292))
        strip_tasks = (640 // 4) * ((480 + 3) // 4)
        sel_tasks = (640 // 4) * 480
        entry("fast_score_strip_kernel+fast_select_rows_kernel+detect_decode_kernel",
              [("fast_score_strip_kernel", (strip_tasks + 255) // 256 * 256 * 5 * nfr), ("fast_select_rows_kernel", (strip_tasks + 255) // 256 * 256 * 5 * nfr),
               ("detect_decode_kernel", (nfr * 26 * 20 + 255) // 256 * 256)], nfr * 3 * pyr5)
        for (k, g), cs in sorted(by_grid.items()):
            print("secondary dispatch", k[:60], g, {c: len(v) for c, v in cs.items()})
    if case in ALG:
        sub, alg, nd = ALG[case]
        for k, cs in ctr.items():
            if sub in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024.0 * nd
                write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024.0 * nd
                pmc["hbm_traffic_per_launch"].append(dict(
                    case=case, kernel=k, dispatches_per_launch=nd, algorithmic_bytes_per_launch=alg, fetch_bytes_raw=fetch,
                    fetch_bytes_gfx950_corrected=2.0 * fetch, write_bytes=write,
                    traffic_over_algorithmic=(2.0 * fetch + write) / alg))
with open(os.path.join(src, "r06_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=["case", "kernel", "calls", "avg_ns", "min_ns", "max_ns", "total_ns", "percent"])
    w.writeheader()
    w.writerows(stats_rows)
# overlap of consecutive launches on several streams (case main)
ov = {}
for f in glob.glob(f"{src}/main/trace/**/*_kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "sparse_align_reg_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[20:]                                  # the bench's warm-up launches
    if len(rows) < 10:
        continue
    st = [int(r["Start_Timestamp"]) for r in rows]
    en = [int(r["End_Timestamp"]) for r in rows]
    events = sorted([(t, 1) for t in st] + [(t, -1) for t in en])
    busy, depth, last, weighted = 0, 0, events[0][0], 0
    for t, dlt in events:
        if depth > 0:
            busy += t - last
            weighted += (t - last) * depth
        depth += dlt
        last = t
    ov = dict(launches=len(rows), streams=len({r.get("Stream_Id", r.get("Queue_Id")) for r in rows}),
              own_duration_avg_ns=sum(e - s for s, e in zip(st, en)) / len(rows),
              union_span_ns=busy, union_span_per_launch_ns=busy / len(rows), launches_in_flight_avg=weighted / busy,
              first_start=st[0], last_end=max(en), wall_per_launch_ns=(max(en) - st[0]) / len(rows))
pmc["overlap"] = ov
json.dump(pmc, open(os.path.join(src, "r06_bench_pmc.json"), "w"), indent=1)
print(json.dumps(ov, indent=1))
print(json.dumps(pmc["hbm_traffic_per_launch"], indent=1))
for r in stats_rows:
    print(r["case"], r["kernel"][:70], r["calls"], r["avg_ns"])
