"""Round-5 soak: long back-to-back streams against the oracle at the benchmark size -- the default path vs BATCHED, the literal mode vs FAITHFUL -- frames queued without a
synchronisation, host images and device tensors alternating in blocks of 40.   python tools/soak_r5.py [frames_default=400] [frames_literal=200]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import BATCHED, FAITHFUL, OracleTSDF
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
from util import C2, sort_export
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 200
frames = list(syn.sphere_room_stream(max(nd, nl)))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
for name, n, sem, mode in (("default vs BATCHED", nd, 0, BATCHED), ("literal vs FAITHFUL", nl, 1, FAITHFUL)):
    g = DenseTSDF(**C2, max_bricks=65536); g.set_dep_camera_intrinsic(syn.K_DEPTH)
    if sem: g.set_option("semantics", 1)
    t0 = time.perf_counter()
    for f in range(n):
        R, T, d = frames[f]
        g.recast_depth_to_map(R, T, d if (f // 40) % 2 else dev[f], None)
    g.sync(); dt = time.perf_counter() - t0
    e = sort_export(g.export_submap())
    ov = g.get_option("overlapped_launches")
    o = OracleTSDF(**C2); o.set_intrinsics(syn.K_DEPTH)
    for R, T, d in frames[:n]: o.integrate_depth(R, T, d, mode=mode)
    w = sort_export(o.export_sparse())
    same = all(e[k].shape == w[k].shape and np.array_equal(e[k], w[k]) for k in ("indices", "TSDF", "W_TSDF", "occupy"))
    print(f"SOAK {name}: {n} frames back to back ({n / dt:.0f} frames/s, {ov} batches into a busy pipeline), {w['indices'].shape[0]} voxels, bit-exact: {same}", flush=True)
    del g
