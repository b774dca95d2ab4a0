"""Developer probe: per-wave phase timestamps of k_integrate_bricks (TSL_TIMING build)."""
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
frames = list(syn.sphere_room_stream(14))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
L = _lib.lib(); L.tsl_tsdf_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
out = np.zeros((16384, 16), np.int64)
for i in range(12): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
L.tsl_tsdf_debug_counters(m.h, out.ctypes.data_as(ctypes.c_void_p), 1)
m.recast_depth_to_map(frames[12][0], frames[12][1], dev[12], None)
L.tsl_tsdf_debug_counters(m.h, out.ctypes.data_as(ctypes.c_void_p), 1)
o = out[out[:, 4] != 0]
t0 = o[:, 0].min()
print("waves with work", len(o), "span (10 ns ticks)", o[:, 4].max() - t0, "(last part of each wave only)")
for nm, a, b in (("zero+sort", 0, 1), ("compute", 1, 2), ("barrier", 2, 3), ("flush", 3, 4)):
    d = o[:, b] - o[:, a]
    print(f"  {nm:10s} mean {d.mean():8.1f} p50 {np.median(d):8.1f} p90 {np.percentile(d, 90):8.1f} max {d.max():8.1f}")
print("  start offset mean %.0f max %.0f; end offset mean %.0f" % ((o[:, 0] - t0).mean(), (o[:, 0] - t0).max(), (o[:, 4] - t0).mean()))
ns = o[:, 10]
for lo, hi in ((0, 64), (64, 256), (256, 600), (600, 1025)):
    sel = (ns >= lo) & (ns < hi)
    if sel.any():
        print(f"  parts with {lo:4d}<=nseg<{hi:4d}: {sel.sum():5d} waves, compute mean {(o[sel, 2] - o[sel, 1]).mean():8.1f}, flush mean {(o[sel, 4] - o[sel, 3]).mean():7.1f}, whole frac {o[sel, 11].mean():.2f}")
print("timeline (us from first wave start): bucket, waves, start p10/p50/p90/max, end p10/p50/p90/max")
for lo, hi in ((0, 64), (64, 256), (256, 600), (600, 1025)):
    for wh in (1, 0):
        sel = (ns >= lo) & (ns < hi) & (o[:, 11] == wh)
        if not sel.any(): continue
        st = (o[sel, 0] - t0) / 100.0; en = (o[sel, 4] - t0) / 100.0
        f = lambda a: "%.1f/%.1f/%.1f/%.1f" % (np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max())
        print(f"  nseg [{lo},{hi}) whole={wh}: {sel.sum():5d} waves  start {f(st)}  end {f(en)}  | sort {np.mean(o[sel,1]-o[sel,0])/100:.1f} compute {np.mean(o[sel,2]-o[sel,1])/100:.1f} barrier {np.mean(o[sel,3]-o[sel,2])/100:.1f} flush {np.mean(o[sel,4]-o[sel,3])/100:.1f}")
