"""Developer probe: per-phase cycle counters of a TSL_TIMING build (see tsl_integrate.hip)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd import _lib
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
for a in sys.argv[1:]:
    k, v = a.split("="); m.set_option(k, int(v))
frames = list(syn.sphere_room_stream(30))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
L = _lib.lib(); L.tsl_tsdf_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
out = np.zeros((16384, 16), np.int64)
for i in range(12): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
L.tsl_tsdf_debug_counters(m.h, out.ctypes.data_as(ctypes.c_void_p), 1)
m.recast_depth_to_map(frames[12][0], frames[12][1], dev[12], None)
L.tsl_tsdf_debug_counters(m.h, out.ctypes.data_as(ctypes.c_void_p), 1)
used = out[:, 8] != 0
o = out[used]
t0 = o[:, 8].min()
work = o[:, 15] != 0
print("waves", used.sum(), "working", work.sum(), "kernel span (ticks of wall_clock64, 100 MHz = 10 ns):", o[:, 6].max() - t0)
names = {8: "start", 0: "lds zero+sync", 1: "load_ray+occupy", 2: "walk", 3: "sync", 4: "resolve", 5: "reserve+sync", 6: "copy+hist flush"}
order = [8, 0, 1, 2, 3, 4, 5, 6]
for grp, sel in (("working", work), ("idle", ~work)):
    g = o[sel]
    print(grp, "waves: start offset mean %.0f max %.0f ; end offset mean %.0f max %.0f" % ((g[:, 8] - t0).mean(), (g[:, 8] - t0).max(), (g[:, 6] - t0).mean(), (g[:, 6] - t0).max()))
    for a, b in zip(order[:-1], order[1:]):
        d = g[:, b] - g[:, a]
        print(f"   {names[b]:18s} mean {d.mean():8.1f}  p50 {np.median(d):8.1f}  max {d.max():8.1f}")
