"""Rate of tsl_tsdf_integrate_depth with HOST images (the reference API's calling convention, taichislam_node.py:381-382), as a stream.
python tools/host_input_probe.py [calls]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
from util import C2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
host = list(syn.sphere_room_stream(40))
g = DenseTSDF(**C2); g.set_dep_camera_intrinsic(syn.K_DEPTH)
for R, T, d in host[:16]:
    g.recast_depth_to_map(R, T, d, None)
g.sync()
t0 = time.perf_counter()
for i in range(n):
    R, T, d = host[i % 40]
    g.recast_depth_to_map(R, T, d, None)
t1 = time.perf_counter()
g.sync()
t2 = time.perf_counter()
print(f"{n} host images: {n / (t2 - t0):.0f} frames/s ({1e6 * (t1 - t0) / n:.1f} us per call on the host, {1e3 * (t2 - t1):.2f} ms drain)")
