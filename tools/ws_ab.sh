#!/bin/bash
# workspace-kernel variants on the batched large-pair shapes (bench secondary entries), same box
for v in "$@"; do
  cp variants/lib_$v.so dsdtm_amd/csrc/libdsdtm_amd.so
  python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for s in d['secondary'][:2]: print('$v', s['workload'][:48], '%.0f /s' % s['value'], '%.4f ms' % s['roofline']['kernel_ms_avg'], s['pose_delta_vs_cpu']['iterations_equal'], s['pose_delta_vs_cpu']['max_rad'])
"
done
