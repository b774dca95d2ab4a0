"""Developer probe: the timeline of k_seq_group's work items in a TSL_SEQ_TRACE build (TSL_EXTRA_FLAGS=-DTSL_SEQ_TRACE python -m taichislam_amd.build, or TSL_LIB=...):
one isolated batch of eight frames; per item start / end on the device's 100 MHz clock, segments, tuples, CU."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd import _lib
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH); m.set_option("semantics", 1); m.set_option("ramp", 0)
frames = list(syn.sphere_room_stream(40))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
L = _lib.lib(); L.tsl_tsdf_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
out = np.zeros(16384 * 16, np.int64)
for i in range(32): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
m.sync()
L.tsl_tsdf_debug_counters(m.h, out.ctypes.data_as(ctypes.c_void_p), 1)
for i in range(32, 40): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)      # ONE batch of eight, alone on the device
m.sync()
L.tsl_tsdf_debug_counters(m.h, out.ctypes.data_as(ctypes.c_void_p), 1)
r = out[1024:1024 + 4 * 8 * 1024].reshape(8, 1024, 4)
on = r[:, :, 1] > 0
t0 = r[:, :, 0][on].min()
st, en = (r[:, :, 0][on] - t0) / 100.0, (r[:, :, 1][on] - t0) / 100.0          # us
seg, tup = r[:, :, 2][on], r[:, :, 3][on] & 0xffffffff
hw = (r[:, :, 3][on] >> 32) & 0xffffffff
dur = en - st
print(f"items {on.sum()} ({on.sum(axis=1).tolist()} per frame), span {en.max():.0f} us, sum of item times {dur.sum() / 1000:.1f} ms -> mean concurrency {dur.sum() / en.max():.0f} items")
print(f"item time us: median {np.median(dur):.0f} p90 {np.percentile(dur, 90):.0f} max {dur.max():.0f};  segments median {np.median(seg):.0f} max {seg.max()};  tuples median {np.median(tup):.0f} max {tup.max()}")
for lo in range(0, int(en.max()) + 1, 50):
    a = ((st < lo + 50) & (en > lo)).sum()
    print(f"  {lo:5d}..{lo + 50:5d} us: {a:4d} items alive, {((st >= lo) & (st < lo + 50)).sum():4d} started")
k = np.argsort(-dur)[:12]
print("longest items: " + ", ".join(f"{dur[i]:.0f} us ({seg[i]} seg, {tup[i]} tup, start {st[i]:.0f})" for i in k))
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 16) & 0xf) << 7)      # cu_id | se_id | (xcc?) -- only used to count distinct places
print(f"distinct (cu, se, ..) ids seen: {len(np.unique(cu))}")
# per-tuple and per-segment cost (least squares): t = a + b * seg + c * tup
A = np.stack([np.ones_like(dur), seg.astype(float), tup.astype(float)], 1)
coef = np.linalg.lstsq(A, dur, rcond=None)[0]
print(f"fit: item time = {coef[0]:.1f} us + {1000 * coef[1]:.1f} ns per segment + {1000 * coef[2]:.2f} ns per tuple")
