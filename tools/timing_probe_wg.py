"""Developer probe: per-workgroup part timeline of k_integrate_bricks3 (TSL_TIMING build: TSL_LIB=.../libtaichislam_hip_timing.so)."""
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
rec = out.reshape(-1)[131072:131072 + 512 * 64].reshape(512, 8, 8)
used = rec[:, :, 0] != 0
t0 = rec[:, :, 0][used].min()
nparts = used.sum(1)
print("workgroups with work", (nparts > 0).sum(), "parts", used.sum(), "parts per wg histogram", np.bincount(nparts))
end = np.where(used, rec[:, :, 4], 0).max(1)
print("wg end time us: p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile((end[nparts > 0] - t0) / 100.0, [50, 90, 99, 100])))
order = np.argsort(-end)
print("slowest workgroups: wg | per part: nseg np start walk_end(dur) flush_end(dur)")
for w in order[:14]:
    s = f"  wg {w:3d} |"
    for t in range(8):
        if not used[w, t]: break
        r = rec[w, t]
        s += f" [{r[1]:5d} np{r[2]:2d} @{(r[0]-t0)/100:5.1f} walk {(r[3]-r[0])/100:5.1f} flush {(r[4]-r[3])/100:5.1f}]"
    print(s)
# aggregate by class
ns = rec[:, :, 1][used]; npp = rec[:, :, 2][used]; walk = (rec[:, :, 3] - rec[:, :, 0])[used] / 100.0; fl = (rec[:, :, 4] - rec[:, :, 3])[used] / 100.0; st = (rec[:, :, 0][used] - t0) / 100.0
for lo, hi in ((0, 64), (64, 256), (256, 600), (600, 1025), (1025, 2049), (2049, 70000)):
    for split in (0, 1):
        sel = (ns >= lo) & (ns < hi) & ((npp > 1) == bool(split))
        if sel.any():
            print(f"  nseg [{lo},{hi}) split={split}: {sel.sum():4d} parts  start p50 {np.median(st[sel]):5.1f}  walk mean {walk[sel].mean():5.1f} p90 {np.percentile(walk[sel],90):5.1f}  flush mean {fl[sel].mean():5.1f} p90 {np.percentile(fl[sel],90):5.1f} max {fl[sel].max():5.1f}")
