"""Developer probe: esdf_stats() of a few updates of the config-4 stream (run under rocprofv3 --kernel-trace to see the per-round durations)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
md = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 24
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
frames = list(syn.sphere_room_stream(N))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
for i in range(N):
    m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
    t0 = time.perf_counter(); m.update_esdf(max_dist=md); dt = time.perf_counter() - t0
    if i >= N - 4: print(i, f"{dt*1e3:.2f} ms", m.esdf_stats())
