"""Developer probe: per-stage cycles of k_seq_group in a TSL_SEQ_TIMING build (TSL_EXTRA_FLAGS=-DTSL_SEQ_TIMING python -m taichislam_amd.build).
Round 5: the stage table in DESIGN.md section 4 was taken with round 4's kernel; on the persistent, stash-free k_seq_group the TSL_SEQ_TIMING build has NOT been
revalidated (it ended in a GPU memory fault twice, while the shipping build passes every test and TSL_SEQ_VERIFY's brute-force recount) -- use the per-item
trace instead: TSL_EXTRA_FLAGS=-DTSL_SEQ_TRACE + tools/seq_trace_probe.py."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd import _lib
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH); m.set_option("semantics", 1)
n = 48
frames = list(syn.sphere_room_stream(n))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
L = _lib.lib(); L.tsl_tsdf_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
out = np.zeros((16384, 16), np.int64)
for i in range(16): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
L.tsl_tsdf_debug_counters(m.h, out.ctypes.data_as(ctypes.c_void_p), 1)
for i in range(16, n): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
L.tsl_tsdf_debug_counters(m.h, out.ctypes.data_as(ctypes.c_void_p), 1)
c = out.reshape(-1)[:16]
names = ["clear + load keys", "bitonic sort", "prefix of steps + reserve", "walk", "voxel scan + csr", "counting sort"]
tot = float(c[:6].sum())
print(f"{n - 16} frames: items {c[8]}, segments {c[9]} ({c[9] / max(1, c[8]):.0f} per item), tuples {c[10]} ({c[10] / max(1, c[8]):.0f} per item)")
for k, nm in enumerate(names):
    print(f"  {nm:28s} {c[k] / 1e6:10.1f} Mcycles  {100.0 * c[k] / tot:5.1f} %   {c[k] / max(1, c[8]):9.0f} cycles per item")
