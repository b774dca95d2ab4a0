"""Developer probe: the driver's 20-frame burst (5 warm-up frames, sync, 20 frames, sync) with and without the per-launch HIP events bench.py uses for
roofline.achieved -- what the events themselves cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd import _lib
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
frames = list(syn.sphere_room_stream(65))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
for prof in (0, 1, 0, 1):
    m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
    for a in sys.argv[1:]:
        k, v = a.split("="); m.set_option(k, int(v))
    for i in range(5): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
    m.sync()
    if prof: m.enable_profiling(True, only=[_lib.K_INTEGRATE])
    res = []
    for rep in range(3):
        b = 5 + rep * 20
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(b, b + 20): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
        m.sync(); torch.cuda.synchronize()
        res.append(20 / (time.perf_counter() - t0))
    print(f"profiling events {'on ' if prof else 'off'}: bursts of 20 frames at", " ".join(f"{r:8.0f}" for r in res), "frames/s")
    del m
