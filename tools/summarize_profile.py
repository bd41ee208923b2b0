"""Folds the rocprofv3 output of tools/profile.sh / tools/pose_opt_profile.sh (kernel-trace stats + one
--pmc pass per counter group) into the per-dispatch summary kept under profiles/.
Usage: python tools/summarize_profile.py gpurun_out/<dir> <kernel name substring> profiles/<out>.json "<version note>" "<command note>" """
import collections, csv, glob, json, sys

src, kern, out, version, command = sys.argv[1:6]
res = {"round": 1, "version": version, "command": command}
for f in glob.glob(f"{src}/trace/runc/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if kern in r["Name"]:
            res["kernel"] = r["Name"]
            res["kernel_trace"] = dict(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]), min_ns=float(r["MinNs"]), max_ns=float(r["MaxNs"]))
ctr = collections.defaultdict(list)
for f in glob.glob(f"{src}/pmc_*/runc/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
            res["launch"] = dict(grid=r["Grid_Size"], workgroup=r["Workgroup_Size"], lds_bytes=r["LDS_Block_Size"],
                                 scratch=r["Scratch_Size"], vgpr_count_field=r["VGPR_Count"])
res["counters"] = {k: dict(dispatches=len(v), mean_per_dispatch=sum(v) / len(v)) for k, v in sorted(ctr.items())}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("kernel_trace", "launch")}, indent=1))
print({k: round(v["mean_per_dispatch"], 1) for k, v in res["counters"].items()})
