#!/bin/bash
# Counters of the <= 320-feature kernel with two and with three pair slots per compute unit, same box, one launch
# stream: kernel trace + one --pmc pass per group (never together with other trace domains).
# Usage on the GPU box: tools/r3_counters.sh <outdir-under-gpurun_out>
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/${1:-r3ctr}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="$REPO/bench.py --steps 60 --warmup 10 --streams 1 --no-cpu --no-secondary"
for slots in 2 3; do
  export DSDTM_REG_SLOTS=$slots
  D="$OUT/s$slots"; mkdir -p "$D"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- python3 $CMD > "$D/trace.log" 2>&1 || echo "trace $slots failed"
  i=0
  for grp in "SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
             "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d "$D/pmc_$i" -- python3 $CMD > "$D/pmc_$i.log" 2>&1 || echo "pmc $slots/$i failed"
  done
  echo "counted $slots"
done
unset DSDTM_REG_SLOTS
cd "$REPO"
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for slots in ("s2", "s3"):
    d = os.path.join(out, slots)
    r = {"kernel_ns": None, "counters": {}}
    for f in glob.glob(f"{d}/trace/**/*_kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "sparse_align" in row["Name"]:
                r["kernel"] = row["Name"]; r["calls"] = int(row["Calls"]); r["kernel_ns"] = float(row["AverageNs"])
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{d}/pmc_*/**/*_counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "sparse_align" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                r["launch"] = dict(grid=row["Grid_Size"], workgroup=row["Workgroup_Size"], lds=row["LDS_Block_Size"], scratch=row["Scratch_Size"], vgpr=row["VGPR_Count"])
    r["counters"] = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
    res[slots] = r
json.dump(res, open(os.path.join(out, "r3_counters.json"), "w"), indent=1)
for slots, r in res.items():
    c = r["counters"]
    print(slots, r.get("kernel", "?")[:60], "avg ns", r["kernel_ns"], r.get("launch"))
    for k, v in c.items():
        print(f"   {k:28s} {v:16.0f}")
    if "SQ_INSTS_VALU" in c and r["kernel_ns"]:
        slots_total = 256 * 4 * (r["kernel_ns"] * 1e-9 * 2.4e9) / 4
        print(f"   VALU issue fraction (2.4 GHz, 4 cycles per wave instruction): {c['SQ_INSTS_VALU'] / slots_total:.3f}")
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        print(f"   SQ_WAIT_ANY / SQ_WAVE_CYCLES: {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f}")
PY
find "$OUT" -name "*_agent_info.csv" -delete
for slots in 2 3; do rm -rf "$OUT/s$slots/trace" "$OUT/s$slots"/pmc_*/; done
echo ctrdone
