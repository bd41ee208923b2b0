"""Merges the summaries of several tools/profile.sh calls (each call covers some cases) into profiles/:
    python tools/merge_profiles.py gpurun_out/<dir1> gpurun_out/<dir2> ...   -> profiles/r06_kernel_stats.csv, profiles/r06_bench_pmc.json
All parts must have been taken with the same binary (profile_binary_sha); the merge refuses anything else."""
import csv, json, os, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows, pmc = [], None
for d in sys.argv[1:]:
    with open(os.path.join(d, "r06_kernel_stats.csv")) as f:
        rows += list(csv.DictReader(f))
    with open(os.path.join(d, "r06_bench_pmc.json")) as f:
        p = json.load(f)
    if pmc is None:
        pmc = p
        continue
    assert p["profile_binary_sha"] == pmc["profile_binary_sha"], "parts were taken with different binaries"
    pmc["cases"].update(p["cases"])
    pmc["hbm_traffic_per_launch"] += p["hbm_traffic_per_launch"]
    pmc["fp64_per_launch"] += p["fp64_per_launch"]
    if p.get("overlap"):
        pmc["overlap"] = p["overlap"]
    if p.get("secondary_kernel_durations"):
        pmc["secondary_kernel_durations"] = p["secondary_kernel_durations"]
with open(os.path.join(REPO, "profiles", "r06_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
with open(os.path.join(REPO, "profiles", "r06_bench_pmc.json"), "w") as f:
    json.dump(pmc, f, indent=1)
print(len(rows), "kernel rows;", len(pmc["hbm_traffic_per_launch"]), "traffic entries; cases:", sorted(pmc["cases"]))
