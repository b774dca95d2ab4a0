import sys, os, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from util import SMALL, make_pair, small_stream
from taichislam_amd import _lib
K, frames = small_stream(3)
g, o = make_pair(SMALL, K)
for a in sys.argv[1:]:
    k, v = a.split("="); g.set_option(k, int(v))
for R, T, d in frames:
    g.recast_depth_to_map(R, T, d, None)
g.sync()
L = _lib.lib(); L.tsl_tsdf_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
out = np.zeros((16384, 16), np.int64)
L.tsl_tsdf_debug_counters(g.h, out.ctypes.data_as(ctypes.c_void_p), 0)
flat = out.reshape(-1)
print("ctr: nunion %d nheavy %d c0 %d c1 %d c2 %d c3 %d n %d" % tuple(flat[:7]))
tot = int(flat[2] + flat[3] + flat[4] + flat[5])
for it in range(tot):
    d = flat[64 + it * 8: 64 + it * 8 + 8]
    if d[1] >= 0:
        print("  item %d: u %d q %d k %d np %d brick %d nseg %d ticket %d hs %d" % ((it,) + tuple(d)), " slab counts of u:", flat[32768 + d[0] * 8: 32768 + d[0] * 8 + 3].tolist())
