"""Developer probe: per-relaxation timing of k_esdf_round (TSL_TIMING build: TSL_LIB=.../libtaichislam_hip_timing.so)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd import _lib
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
md = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
frames = list(syn.sphere_room_stream(22))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
L = _lib.lib(); L.tsl_tsdf_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
out = np.zeros((16384, 16), np.int64)
for i in range(20):
    m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None); m.update_esdf(max_dist=md)
L.tsl_tsdf_debug_counters(m.h, out.ctypes.data_as(ctypes.c_void_p), 1)
m.recast_depth_to_map(frames[20][0], frames[20][1], dev[20], None); m.update_esdf(max_dist=md)
L.tsl_tsdf_debug_counters(m.h, out.ctypes.data_as(ctypes.c_void_p), 1)
flat = out.reshape(-1)
n = int(flat[0] & 0xffffffff)
rec = flat[8:8 + 4 * n].reshape(n, 4)
print("relaxations", n, "stats", m.esdf_stats())
print("stage us: mean %.1f p90 %.1f max %.1f" % (rec[:, 0].mean() / 100, np.percentile(rec[:, 0], 90) / 100, rec[:, 0].max() / 100))
print("relax us: mean %.1f p50 %.1f p90 %.1f max %.1f" % (rec[:, 1].mean() / 100, np.median(rec[:, 1]) / 100, np.percentile(rec[:, 1], 90) / 100, rec[:, 1].max() / 100))
print("passes: mean %.1f p90 %.0f max %d" % (rec[:, 2].mean(), np.percentile(rec[:, 2], 90), rec[:, 2].max()))
print("us per pass: %.2f" % (rec[:, 1].sum() / rec[:, 2].sum() / 100))
for r in range(int(rec[:, 3].max()) + 1):
    s = rec[:, 3] == r
    if s.any(): print(f"  round {r}: {s.sum():4d} bricks, relax mean {rec[s,1].mean()/100:7.1f} us max {rec[s,1].max()/100:7.1f}, passes mean {rec[s,2].mean():5.1f} max {rec[s,2].max()}")
