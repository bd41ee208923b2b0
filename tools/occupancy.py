import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsdtm_amd import capi
ctx = capi.Context(0, diag=True)   # the diagnostic library (dsdtm_debug_* / switches)
print(ctx.lib.dsdtm_version(), "occupancy WG/CU:", [ctx.lib.dsdtm_debug_occupancy(ctx.handle, v) for v in (0,1,2)])
