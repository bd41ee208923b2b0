#!/bin/bash
# bench.py with 1..8 launch streams (same binary, same box): where do consecutive launches start to overlap?
OUT=gpurun_out/${1:-sweep}; mkdir -p $OUT
for s in 1 2 3 4 5 6 8; do
  python bench.py --no-secondary --no-cpu --streams $s > $OUT/s$s.json 2> $OUT/s$s.err
  python - <<PY
import json
d=json.loads(open("$OUT/s$s.json").read().strip().splitlines()[-1])
print("streams", $s, "value %.0f" % d["value"], "ms/step %.4f" % d["ms_per_step"], "frac %.4f" % d["roofline"]["frac"])
PY
done
for q in 2 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-secondary --no-cpu --streams 2 > $OUT/q${q}_s2.json 2> $OUT/q${q}_s2.err
  python - <<PY
import json
d=json.loads(open("$OUT/q${q}_s2.json").read().strip().splitlines()[-1])
print("GPU_MAX_HW_QUEUES", $q, "streams 2", "value %.0f" % d["value"], "ms/step %.4f" % d["ms_per_step"])
PY
done
