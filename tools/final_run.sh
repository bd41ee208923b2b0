#!/bin/bash
# Everything behind profiles/r06_* that describes the round's final binary besides the rocprofv3 passes (tools/profile.sh):
# the driver-style and the default bench lines, the soak tools. Usage on the GPU box: tools/final_run.sh <outdir-under-gpurun_out>
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"
O=gpurun_out/${1:-final}; mkdir -p $O
python - > $O/sha.txt <<'PY'
import hashlib
print("libdsdtm_amd.so sha256[:16] =", hashlib.sha256(open("dsdtm_amd/csrc/libdsdtm_amd.so", "rb").read()).hexdigest()[:16])
PY
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench.json.log 2> $O/bench.err; echo "bench (driver-style) rc=$?"
cp bench_secondary.json $O/r06_bench_secondary.json
timeout -k 10 600 python bench.py --no-secondary > $O/r06_bench_500.json.log 2>> $O/bench.err; echo "bench (500 steps) rc=$?"
( cat $O/sha.txt
  echo "== soak_streamed";  timeout -k 10 400 python tools/soak_streamed.py 2>&1 | tail -4
  echo "== soak_duo";       timeout -k 10 300 python tools/soak_duo.py 2>&1 | tail -4
  echo "== soak_team";      timeout -k 10 300 python tools/soak_team.py 2>&1 | tail -4
  echo "== soak (150 random configurations)"; timeout -k 10 600 python tools/soak.py 150 2>&1 | tail -4
  echo "== soak_fmd";       timeout -k 10 300 python tools/soak_fmd.py 2>&1 | tail -4
  echo "== soak_pose_opt";  timeout -k 10 300 python tools/soak_pose_opt.py 700 2>&1 | tail -5 ) 2>&1 | grep -v amdgpu.ids > $O/r06_soak.txt
( timeout -k 10 200 python tools/stamps.py ) 2>/dev/null | grep -v amdgpu > $O/r06_stamps.txt
echo done
