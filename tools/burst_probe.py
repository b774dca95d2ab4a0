"""Developer probe: where the wall time of a 20-frame burst goes (the driver's bench command line uses --steps 20 --warmup 5)."""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
m = DenseTSDF(**C2, max_submap_num=9, max_bricks=32768); m.set_dep_camera_intrinsic(syn.K_DEPTH)
for a in sys.argv[1:]:
    k, v = a.split("="); m.set_option(k, int(v))
N = 25 + 5 * 20
host = list(syn.sphere_room_stream(N))
depth = torch.from_numpy(np.stack([d for _, _, d in host]).view(np.int16)).cuda()
fr = [depth[i] for i in range(N)]
poses = [(np.ascontiguousarray(R), np.ascontiguousarray(T)) for R, T, _ in host]
for i in range(5): m.recast_depth_to_map(poses[i][0], poses[i][1], fr[i], None)
m.sync(); torch.cuda.synchronize()
gc.collect(); gc.disable()
for rep in range(6):
    base = 5 + rep * 20
    torch.cuda.synchronize(); m.sync()
    t0 = time.perf_counter()
    for i in range(base, base + 20): m.recast_depth_to_map(poses[i][0], poses[i][1], fr[i], None)
    t1 = time.perf_counter()
    m.sync()
    t2 = time.perf_counter()
    torch.cuda.synchronize(); m.sync()
    t3 = time.perf_counter()
    print(f"burst {rep}: queue loop {1e6*(t1-t0):.0f} us, sync {1e6*(t2-t1):.0f} us, barrier {1e6*(t3-t2):.0f} us -> {20/(t3-t0):.0f} fps")
