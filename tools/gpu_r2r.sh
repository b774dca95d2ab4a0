#!/bin/bash
cd $GRAFT_REPO_ROOT
python - << 'PY'
import sys, time, numpy as np, torch, collections, cProfile, pstats
sys.path.insert(0, ".")
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd import _lib
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
N = 120
frames = list(syn.sphere_room_stream(N))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
for variant in ("first", "second"):
    m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
    for i in range(20): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
    m.sync(); torch.cuda.synchronize()
    pr = cProfile.Profile()
    if variant == "second": pr.enable()
    t0 = time.perf_counter()
    for i in range(20, N): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
    t1 = time.perf_counter(); m.sync(); t2 = time.perf_counter()
    pr.disable()
    print(variant, f"{(N-20)/(t2-t0):.0f} fps enqueue {1e6*(t1-t0)/(N-20):.0f} us/frame")
    if variant == "second": pstats.Stats(pr).sort_stats("tottime").print_stats(8)
PY
