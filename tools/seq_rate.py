"""Literal mode (semantics = 1) in a stream: exactness against the oracle's FAITHFUL map on the first frames, then the steady rate and the per-batch kernel
times (HIP events) on the frames that follow.   python tools/seq_rate.py [frames_checked=40] [frames_timed=400] [opt=value ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from oracle import FAITHFUL, OracleTSDF
    from taichislam_amd import _lib
    from taichislam_amd.mapping import DenseTSDF
    from taichislam_amd.utils import synthetic as syn
    from util import C2, sort_export
    nchk = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    ntim = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    sync_every = max([int(a.split("=")[1]) for a in sys.argv[3:] if a.startswith("sync_every=")] + [0])      # a synchronisation every n timed frames: every batch alone on the device
    opts = [a.split("=") for a in sys.argv[3:] if not a.startswith("late:") and not a.startswith("sync_every=")]
    late = [a[5:].split("=") for a in sys.argv[3:] if a.startswith("late:")]      # options set behind the exactness check (e.g. late:phases=1: phase A alone)
    frames = list(syn.sphere_room_stream(nchk + ntim))
    dd = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
    g = DenseTSDF(**C2)
    g.set_dep_camera_intrinsic(syn.K_DEPTH)
    g.set_option("semantics", 1)
    for k, v in opts:
        g.set_option(k, int(v))
    for (R, T, _), d in zip(frames[:nchk], dd[:nchk]):
        g.recast_depth_to_map(R, T, d, None)
    g.sync()
    e = sort_export(g.export_submap())
    o = OracleTSDF(**C2); o.set_intrinsics(syn.K_DEPTH)
    for R, T, d in frames[:nchk]:
        o.integrate_depth(R, T, d, mode=FAITHFUL)
    w = sort_export(o.export_sparse())
    exact = all(e[k].shape == w[k].shape and np.array_equal(e[k], w[k]) for k in ("indices", "TSDF", "W_TSDF", "occupy"))
    print(f"exact vs FAITHFUL after {nchk} frames: {exact} ({w['indices'].shape[0]} voxels)", flush=True)
    for k, v in late:
        g.set_option(k, int(v))
    g.enable_profiling(True, only=[_lib.K_INTEGRATE, _lib.K_RAYS, _lib.K_SORT])
    t0 = time.perf_counter()
    for f, ((R, T, _), d) in enumerate(zip(frames[nchk:], dd[nchk:])):
        g.recast_depth_to_map(R, T, d, None)
        if sync_every and f % sync_every == sync_every - 1:
            g.sync()
    g.sync()
    dt = time.perf_counter() - t0
    kt = {k: g.kernel_time(i) for k, i in (("replay", _lib.K_INTEGRATE), ("split+group+classify", _lib.K_RAYS), ("rank", _lib.K_SORT))}
    print(f"steady {ntim / dt:.0f} frames/s over {ntim} frames; per batch: " + ", ".join(f"{k} {1000.0 * a / max(1, b):.0f} us ({b})" for k, (a, b) in kt.items()), flush=True)
    if not exact:
        sys.exit(3)


if __name__ == "__main__":
    main()
