#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json (not the bench.py headline): C1 sphere + marching cubes, C3 Octomap 1024^3,
C4 TSDF + ESDF + marching cubes every 10th frame.  Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd.mapping import DenseTSDF, MarchingCubeMesher, Octomap
from taichislam_amd.utils import synthetic as syn

out = {}
N = 120
host = list(syn.sphere_room_stream(N))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in host]

# C1: analytic sphere 128^3, mesh once
g = DenseTSDF(map_scale=[6.4, 6.4], voxel_scale=0.05, num_voxel_per_blk_axis=16)
g.init_sphere(voxels=100, radius=1.5)
ms = MarchingCubeMesher(g, 1000000)
ms.generate_mesh(1); g.sync()
t = time.perf_counter()
for _ in range(20): ms.generate_mesh(1)
g.sync()
out["C1_sphere128_mc"] = {"ms_per_mesh": 1000 * (time.perf_counter() - t) / 20, "triangles": ms.num_facelets[None], "voxels": g.count_active()}

# C3: octomap 1024^3 / 5 cm
oc = Octomap(map_scale=[51.2, 51.2], voxel_scale=0.05, K=2, max_ray_length=5.0, max_submap_num=4)
oc.set_dep_camera_intrinsic(syn.K_DEPTH)
for f in range(20): oc.recast_depth_to_map(host[f][0], host[f][1], dev[f], None)
oc.sync()
t = time.perf_counter()
for f in range(20, N): oc.recast_depth_to_map(host[f][0], host[f][1], dev[f], None)
oc.sync()
out["C3_octomap_1024"] = {"frames_per_s": (N - 20) / (time.perf_counter() - t), "leaves": int(oc.export_leaves()[0].shape[0])}

# C4: TSDF + ESDF every frame? (the reference's ESDF hook ran per frame; here every 10th with the mesh, full recompute)
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
mesher = MarchingCubeMesher(m, 4000000, tsdf_surface_thres=0.1)
for f in range(10): m.recast_depth_to_map(host[f][0], host[f][1], dev[f], None)
mesher.generate_mesh(1); m.update_esdf(max_dist=1.0); m.sync()
t = time.perf_counter(); t_mesh = t_esdf = 0.0
for f in range(10, N):
    m.recast_depth_to_map(host[f][0], host[f][1], dev[f], None)
    if f % 10 == 9:
        m.sync(); a = time.perf_counter(); mesher.generate_mesh(1); b = time.perf_counter(); it = m.update_esdf(max_dist=1.0); c = time.perf_counter()
        t_mesh += b - a; t_esdf += c - b
m.sync()
dt = time.perf_counter() - t
out["C4_tsdf_esdf_mc"] = {"frames_per_s": (N - 10) / dt, "ms_per_mesh": 1000 * t_mesh / 11, "ms_per_esdf": 1000 * t_esdf / 11, "triangles": mesher.num_facelets[None],
                          "esdf_launches": it, "voxels": m.count_active()}
print(json.dumps(out))
