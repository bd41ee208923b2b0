#!/bin/bash
# pyrDown variants, same box (DSDTM_PYRDOWN_OVERLAP=1: the general overlapping-load kernel, which carries the switches)
for v in "$@"; do
  cp variants/lib_$v.so dsdtm_amd/csrc/libdsdtm_amd.so
  python -m pytest tests/test_align2d_gpu.py -m gpu -x -q -k pyrdown 2>&1 | tail -1
  for rep in 1 2; do
    DSDTM_PYRDOWN_OVERLAP=1 python tools/kernels.py 2>/dev/null | grep pyrDown | sed "s/^/$v overlap: /" | cut -c1-140
  done
done
