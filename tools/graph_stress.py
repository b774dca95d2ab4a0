"""Developer stress test for the opt-in hipGraph replay path: many handles, shapes and option changes in one process."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
from util import SMALL, small_stream
K, frames = small_stream(6)
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
if "t" in mode:
    import torch
    keep = torch.zeros((1 << 21, 2), dtype=torch.int64, device="cuda")
    del keep
for it in range(40):
    cfg = dict(SMALL, max_submap_num=(16 if it % 3 == 0 else 1024))
    if it % 4 == 1: cfg = dict(cfg, map_scale=[5.12, 5.12])
    m = DenseTSDF(**cfg); m.set_dep_camera_intrinsic(K); m.set_option("graph", int(os.environ.get("G", "1")))
    other = DenseTSDF(**dict(cfg, is_global_map=True)) if it % 2 == 0 else None
    for f, (R, T, d) in enumerate(frames):
        m.recast_depth_to_map(R, T, d, None)
        if "b" in mode and f == 2: m.export_submap()
        if "b" in mode and f == 3 and other is not None:
            m.switch_to_next_submap(); other.fuse_submaps(m)
    n = m.count_active()
    print(it, n, flush=True)
    del m, other
    if it % 5 == 0: gc.collect()
print("done")
