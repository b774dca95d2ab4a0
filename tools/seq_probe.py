"""Throughput (and optionally parity) of the reference-literal sequential semantics on the GPU: BASELINE configs[1] stream, semantics = 1.

  python tools/seq_probe.py [--frames 72] [--warmup 8] [--impl 1] [--check 12] [--opt name=value ...]

--check N integrates N further frames on a fresh map beside the oracle's FAITHFUL replay and compares every TSDF / W bit."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=72)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--impl", type=int, default=1)
    ap.add_argument("--check", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    import torch
    from taichislam_amd.mapping import DenseTSDF
    from taichislam_amd.utils import synthetic as syn
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from util import C2
    n = a.warmup + a.frames
    host = list(syn.sphere_room_stream(n))
    dev = torch.from_numpy(np.stack([d for _, _, d in host]).view(np.int16)).cuda()
    g = DenseTSDF(**C2)
    g.set_dep_camera_intrinsic(syn.K_DEPTH)
    g.set_option("semantics", 1)
    g.set_option("seq_impl", a.impl)
    for kv in a.opt:
        k, v = kv.split("="); g.set_option(k, int(v))
    for f in range(a.warmup):
        g.recast_depth_to_map(host[f][0], host[f][1], dev[f], None)
    g.sync()
    t0 = time.perf_counter()
    for f in range(a.warmup, n):
        g.recast_depth_to_map(host[f][0], host[f][1], dev[f], None)
    g.sync()
    dt = time.perf_counter() - t0
    print(f"seq_impl {a.impl}: {a.frames} frames in {1e3 * dt:.1f} ms = {a.frames / dt:.0f} frames/s; last frame {g.last_frame_stats()}", flush=True)
    if a.impl == 1:
        print(f"  last batch: {g.get_option('seq_long_voxels')} voxels with a wave of their own, longest runs summed over its frames {g.get_option('seq_longest_run')} updates", flush=True)
    if a.check:
        from oracle import FAITHFUL, OracleTSDF
        from util import assert_export_equal
        g2 = DenseTSDF(**C2); g2.set_dep_camera_intrinsic(syn.K_DEPTH); g2.set_option("semantics", 1); g2.set_option("seq_impl", a.impl)
        o = OracleTSDF(**C2); o.set_intrinsics(syn.K_DEPTH, syn.K_DEPTH)
        for f in range(a.check):
            g2.recast_depth_to_map(host[f][0], host[f][1], dev[f], None)
            o.integrate_depth(host[f][0], host[f][1], host[f][2], mode=FAITHFUL)
        assert_export_equal(g2.export_submap(), o.export_sparse(), f"seq_impl {a.impl} vs FAITHFUL, {a.check} frames")
        print(f"seq_impl {a.impl}: {a.check} frames bit-exact with oracle FAITHFUL", flush=True)


if __name__ == "__main__":
    main()
