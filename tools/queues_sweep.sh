#!/bin/bash
# launch streams x HIP hardware queues (GPU_MAX_HW_QUEUES, default 4): how many launches can really overlap?
for q in 4 8 16; do for s in 8 12 16; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-secondary --no-cpu --streams $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GPU_MAX_HW_QUEUES $q streams $s: %.0f /s  %.4f ms/step' % (d['value'], d['ms_per_step']))"
done; done
