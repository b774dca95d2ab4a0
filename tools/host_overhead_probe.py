"""Developer probe: host enqueue time vs GPU completion time per frame."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
for a in sys.argv[1:]:
    k, v = a.split("="); m.set_option(k, int(v))
N = 220
frames = list(syn.sphere_room_stream(N))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
for i in range(20): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
m.sync()
import gc; gc.collect(); gc.disable()
t0 = time.perf_counter()
for i in range(20, N): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
t1 = time.perf_counter(); m.sync(); t2 = time.perf_counter()
n = N - 20
print(f"host enqueue {1e6*(t1-t0)/n:.1f} us/frame, total {1e6*(t2-t0)/n:.1f} us/frame -> {n/(t2-t0):.0f} fps")
