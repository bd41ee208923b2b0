for v in pd8 pd16 pd4 pd8; do cp variants/lib_$v.so dsdtm_amd/csrc/libdsdtm_amd.so; echo $v; python tools/kernels.py 2>/dev/null | grep pyrDown; done
