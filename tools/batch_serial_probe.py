"""Developer probe: batches issued one at a time (sync after every batch) so kernel times are uncontended."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
for a in sys.argv[1:]:
    k, v = a.split("="); m.set_option(k, int(v))
N = 120
frames = list(syn.sphere_room_stream(N))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
t0 = time.perf_counter()
for i in range(N):
    m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
    if i % 4 == 3: m.sync()
m.sync()
print(f"{1e6*(time.perf_counter()-t0)/N:.1f} us/frame")
