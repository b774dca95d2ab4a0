"""Developer probe: per-kernel times with batches of eight issued ONE AT A TIME (a sync after every batch: nothing shares the device
with the kernel being timed), next to the pipelined rate of the same options.   python tools/phase_probe.py [opt=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd import _lib
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
m.set_option("ramp", 0)
for a in sys.argv[1:]:
    k, v = a.split("="); m.set_option(k, int(v))
N = 160
frames = list(syn.sphere_room_stream(N))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
for i in range(32):
    m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
m.sync()
m.enable_profiling(True)
t0 = time.perf_counter()
for i in range(32, 128):
    m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
    if i % 8 == 7: m.sync()
m.sync()
ser = 1e6 * (time.perf_counter() - t0) / 96
out = []
for kid, name in _lib.KERNEL_NAMES.items():
    ms, n = m.kernel_time(kid)
    if n: out.append(f"{name} {1000.0 * ms / n:.1f}")
m.enable_profiling(False)
print(" ".join(sys.argv[1:]) or "default", "| serial batches of 8, us per launch:", ", ".join(out), f"| {ser:.1f} us/frame incl. sync")
try:
    for i in range(128, 136):
        m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
    print("   last batch: heavy bricks", m.get_option("last_heavy_bricks"), "slab slots", m.get_option("last_slab_slots"), "frame bricks", m.last_frame_stats()["bricks"])
except Exception as e:
    print("   (no batch statistics:", e, ")")
