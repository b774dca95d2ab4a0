"""Developer probe: config 4 (TSDF + ESDF every frame + mesh every 10th) -- where does the wall time go: host enqueue or GPU?"""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd.mapping import DenseTSDF, MarchingCubeMesher
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
for a in sys.argv[1:]:
    k, v = a.split("="); m.set_option(k, int(v))
mesher = MarchingCubeMesher(m, 4000000, tsdf_surface_thres=5 * C2["voxel_scale"])
N = 110
frames = list(syn.sphere_room_stream(N))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
for f in range(10):
    m.recast_depth_to_map(frames[f][0], frames[f][1], dev[f], None); m.update_esdf(max_dist=1.0, wait=False)
m.sync(); m.esdf_totals()
gc.collect(); gc.disable()
tr = te = tm = 0.0
t0 = time.perf_counter()
for f in range(10, N):
    a = time.perf_counter(); m.recast_depth_to_map(frames[f][0], frames[f][1], dev[f], None)
    b = time.perf_counter(); m.update_esdf(max_dist=1.0, wait=False)
    c = time.perf_counter()
    if f % 10 == 9: mesher.generate_mesh(1)
    d = time.perf_counter()
    tr += b - a; te += c - b; tm += d - c
t1 = time.perf_counter()
m.esdf_totals(); m.sync()
t2 = time.perf_counter()
n = N - 10
print(f"{' '.join(sys.argv[1:]) or 'default'}: host enqueue {1e6*(t1-t0)/n:.0f} us/frame (recast {1e6*tr/n:.0f}, update_esdf {1e6*te/n:.0f}, mesh {1e6*tm/n:.0f}); drain {1e6*(t2-t1)/n:.0f} us/frame; total {1e6*(t2-t0)/n:.0f} us/frame = {n/(t2-t0):.0f} frames/s")
# host cost of the calls with an empty queue behind them: the first calls after a sync cannot be held back by anything in flight
gc.enable(); m.sync(); m.esdf_totals()
per = []
for f in range(20, 28):
    a = time.perf_counter(); m.recast_depth_to_map(frames[f][0], frames[f][1], dev[f], None)
    b = time.perf_counter(); m.update_esdf(max_dist=1.0, wait=False)
    c = time.perf_counter(); per.append((1e6 * (b - a), 1e6 * (c - b)))
m.sync(); m.esdf_totals()
print("  first calls after a sync (recast us, update_esdf us):", " ".join(f"({x:.0f},{y:.0f})" for x, y in per))
