"""Developer probe: one DenseTSDF per process; argv: [prof] [submap] [stacked] [N=..]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd import _lib
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
a = sys.argv[1:]
N = 330
frames = list(syn.sphere_room_stream(N))
dev = torch.from_numpy(np.stack([d for _, _, d in frames]).view(np.int16)).cuda() if "stacked" in a else [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
kw = dict(max_submap_num=9, max_bricks=32768) if "submap" in a else {}
m = DenseTSDF(**C2, **kw); m.set_dep_camera_intrinsic(syn.K_DEPTH)
if "submap" in a:
    m.active_submap_id[None] = 0; m.set_base_pose_submap(0, frames[0][0], frames[0][1])
for i in range(30): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
m.sync(); torch.cuda.synchronize()
if "prof" in a: m.enable_profiling(True, only=[_lib.K_INTEGRATE])
t0 = time.perf_counter()
for i in range(30, N): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
t1 = time.perf_counter(); m.sync(); t2 = time.perf_counter()
print(" ".join(a) or "plain", f"{(N-30)/(t2-t0):.0f} fps enqueue {1e6*(t1-t0)/(N-30):.0f} us/frame", m.kernel_time(_lib.K_INTEGRATE) if "prof" in a else "")
