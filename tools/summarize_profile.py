"""Folds the rocprofv3 output of tools/profile.sh / tools/pose_opt_profile.sh (kernel-trace stats + one
--pmc pass per counter group) into the per-dispatch summary kept under profiles/.
Usage: python tools/summarize_profile.py gpurun_out/<dir> <kernel name substring>[,<substring>...] profiles/<out>.json
       "<version note>" "<command note>" [<substring>=<algorithmic bytes per launch> ...]
For every kernel given with its algorithmic bytes an entry of "hbm_traffic_per_launch" is written (what bench.py's
roofline.traffic reads): FETCH_SIZE and WRITE_SIZE are KiB per dispatch; FETCH_SIZE x 2 is the gfx950 correction
of /opt/skills/guides/MI355X_MICROARCH.md (128-B requests tallied at 64 B)."""
import collections, csv, glob, json, sys

src, kerns, out, version, command = sys.argv[1:6]
alg = dict(a.split("=") for a in sys.argv[6:])
res = {"round": 2, "version": version, "command": command, "kernels": {}, "hbm_traffic_per_launch": []}
for kern in kerns.split(","):
    k = {}
    for f in glob.glob(f"{src}/trace/**/*_kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Name"]:
                k["kernel"] = r["Name"]
                k["kernel_trace"] = dict(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), min_ns=float(r["MinNs"]), max_ns=float(r["MaxNs"]))
    ctr = collections.defaultdict(list)
    for f in glob.glob(f"{src}/pmc_*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
                k.setdefault("kernel", r["Kernel_Name"])
                k["launch"] = dict(grid=r["Grid_Size"], workgroup=r["Workgroup_Size"], lds_bytes=r["LDS_Block_Size"],
                                   scratch=r["Scratch_Size"], vgpr_count_field=r["VGPR_Count"])
    k["counters"] = {c: dict(dispatches=len(v), mean_per_dispatch=sum(v) / len(v)) for c, v in sorted(ctr.items())}
    res["kernels"][kern] = k
    if kern in alg and "FETCH_SIZE" in k["counters"] and "WRITE_SIZE" in k["counters"]:
        fetch = k["counters"]["FETCH_SIZE"]["mean_per_dispatch"] * 1024.0
        write = k["counters"]["WRITE_SIZE"]["mean_per_dispatch"] * 1024.0
        res["hbm_traffic_per_launch"].append(dict(
            kernel=k["kernel"], algorithmic_bytes_per_launch=int(alg[kern]), fetch_bytes_raw=fetch,
            fetch_bytes_gfx950_corrected=2.0 * fetch, write_bytes=write,
            traffic_over_algorithmic=(2.0 * fetch + write) / float(alg[kern])))
json.dump(res, open(out, "w"), indent=1)
for kern, k in res["kernels"].items():
    print(kern, json.dumps({x: k.get(x) for x in ("kernel_trace", "launch")}))
    print({c: round(v["mean_per_dispatch"], 1) for c, v in k["counters"].items()})
print(json.dumps(res["hbm_traffic_per_launch"], indent=1))
