"""Developer probe: where the host time of a queued frame goes (cProfile of the shim over 200 frames of the benchmark stream)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
N = 240
frames = list(syn.sphere_room_stream(N))
depth = torch.from_numpy(np.stack([d for _, _, d in frames]).view(np.int16)).cuda()
poses = [(np.ascontiguousarray(R), np.ascontiguousarray(T)) for R, T, _ in frames]
for i in range(40): m.recast_depth_to_map(poses[i][0], poses[i][1], depth[i], None)
m.sync()
import gc; gc.collect(); gc.disable()
t0 = time.perf_counter()
for i in range(40, 140): m.recast_depth_to_map(poses[i][0], poses[i][1], depth[i], None)
t1 = time.perf_counter(); m.sync(); t2 = time.perf_counter()
print(f"plain: host enqueue {1e6*(t1-t0)/100:.1f} us/frame, total {1e6*(t2-t0)/100:.1f} us/frame -> {100/(t2-t0):.0f} fps")
pr = cProfile.Profile(); pr.enable()
for i in range(140, 240): m.recast_depth_to_map(poses[i][0], poses[i][1], depth[i], None)
pr.disable(); m.sync()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
