"""Developer probe (GPU): the raise / lower wavefront (esdf_mode 1) against a full recompute (esdf_mode 0 + esdf_full) after every frame of a stream."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
from util import C2, SMALL, lin, small_stream

def srt(m):
    i, e = m.export_esdf(); o = np.argsort(lin(i)); return i[o], e[o]

def run(name, cfg, K, frames, md):
    a = DenseTSDF(**cfg); a.set_dep_camera_intrinsic(K); a.set_option("esdf_mode", 1)
    b = DenseTSDF(**cfg); b.set_dep_camera_intrinsic(K); b.set_option("esdf_mode", 0); b.set_option("esdf_full", 1)
    bad = 0
    for f, (R, T, d) in enumerate(frames):
        for m in (a, b):
            m.recast_depth_to_map(R, T, d, None); m.update_esdf(max_dist=md)
        (ia, ea), (ib, eb) = srt(a), srt(b)
        same = np.array_equal(ia, ib) and np.array_equal(ea.view(np.uint32), eb.view(np.uint32))
        st = a.esdf_stats()
        nd = int((ea.view(np.uint32) != eb.view(np.uint32)).sum()) if ia.shape == ib.shape else -1
        print(f"{name} frame {f}: {'OK ' if same else 'DIFF'} voxels {ia.shape[0]} differing {nd} orphans {a.get_option('esdf_orphans')} | inc {st['incremental']} dirty {st['dirty_bricks']} changed {st['changed_bricks']} reached {st['region_bricks']} visits {st['brick_relaxations']} pushes {st['voxel_pushes']} rounds {st['rounds']} sets {st['passes']} (full: visits {b.esdf_stats()['brick_relaxations']} pushes {b.esdf_stats()['voxel_pushes']})", flush=True)
        if not same:
            bad += 1
            if nd > 0:
                w = np.nonzero(ea.view(np.uint32) != eb.view(np.uint32))[0][:5]
                print("   first:", [(ia[k].tolist(), float(ea[k]), float(eb[k])) for k in w])
    return bad

which = sys.argv[1:] or ["small", "c2"]
bad = 0
if "small" in which:
    K, frames = small_stream(20)
    bad += run("small md0.5", SMALL, K, frames, 0.5)
    bad += run("small md2.0", SMALL, K, frames[:8], 2.0)
if "c2" in which:
    bad += run("c2 md1.0", C2, syn.K_DEPTH, list(syn.sphere_room_stream(8)), 1.0)
print("FRAMES_WITH_DIFFERENCES", bad)
