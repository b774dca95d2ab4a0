"""Developer probe: host enqueue timeline (when does the host start to block?)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 330
frames = list(syn.sphere_room_stream(N))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
for a in sys.argv[2:]:
    k, v = a.split("="); m.set_option(k, int(v))
for i in range(30): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
m.sync(); torch.cuda.synchronize()
ts = np.zeros(N)
import collections
mx = collections.defaultdict(float); orig = m._call
def timed(name, *a):
    t = time.perf_counter(); r = orig(name, *a); d = time.perf_counter() - t
    if d > mx[name]: mx[name] = d
    return r
m._call = timed
import gc; gc.collect(); gc.disable()
t0 = time.perf_counter()
for i in range(30, N):
    m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
    ts[i] = time.perf_counter() - t0
t1 = time.perf_counter(); m.sync(); t2 = time.perf_counter()
d = np.diff(ts[30:]) * 1e6
print(f"N={N} {(N-30)/(t2-t0):.0f} fps; enqueue us per frame by block of 40:", [int(d[k:k+40].mean()) for k in range(0, len(d), 40)], "max gap us", int(d.max()), "at", int(d.argmax()) + 31, "final sync ms %.2f" % (1e3 * (t2 - t1)), {k: int(v * 1e6) for k, v in mx.items()})
