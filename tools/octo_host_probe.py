import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np
from taichislam_amd.mapping import Octomap
from taichislam_amd.utils import synthetic as syn
oc = Octomap(map_scale=[51.2, 51.2], voxel_scale=0.05, K=2, max_ray_length=5.0, max_submap_num=4)
oc.set_dep_camera_intrinsic(syn.K_DEPTH)
fr = list(syn.sphere_room_stream(60))
for R, T, d in fr[:10]: oc.recast_depth_to_map(R, T, d, None)
oc.sync()
t0 = time.perf_counter()
for i in range(300):
    R, T, d = fr[i % 60]; oc.recast_depth_to_map(R, T, d, None)
oc.sync(); dt = time.perf_counter() - t0
print("octomap host numpy images:", 300 / dt, "frames/s", 1e6 * dt / 300, "us per call")
