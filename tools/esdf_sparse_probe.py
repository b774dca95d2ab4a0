"""Developer probe (GPU): the two forms of the incremental ESDF update when a frame changes LITTLE -- a static camera in the sphere room and a ball of 15 cm radius that
moves 3 cm per frame, 1.5 m in front of the wall.  The room's TSDF does not move (identical measurements), only the bricks around the ball do.  Prints ms per
update (HIP events around the update's launches), bricks reached and voxel writes for esdf_mode 1 (wavefront) and 0 (regional recompute), and checks the two maps
against each other."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from taichislam_amd import _lib
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
from util import C2, lin

def ball_depth(R, T, base, centre, rad):
    fx, fy, cx, cy = syn.K_DEPTH[0], syn.K_DEPTH[4], syn.K_DEPTH[2], syn.K_DEPTH[5]
    h, w = base.shape
    ii, jj = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    d = np.stack([(ii - cx) / fx, (jj - cy) / fy, np.ones_like(ii)], axis=-1) @ R.T
    oc = T - centre
    a = (d * d).sum(-1); b = 2.0 * (d @ oc); c = float(oc @ oc) - rad * rad
    disc = b * b - 4 * a * c
    t = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
    mm = np.where((t > 0) & np.isfinite(t), np.rint(1000.0 * t), 65535.0)
    return np.minimum(base.astype(np.float64), mm).astype(np.uint16)

N, WARM = 60, 20
R, T = syn.camera_pose(0)
base = syn.sphere_room_depth(R, T)
fwd = R @ np.array([0.0, 0.0, 1.0]); right = R @ np.array([1.0, 0.0, 0.0])
frames = [ball_depth(R, T, base, T + 1.5 * fwd + (0.03 * f - 0.9) * right, 0.15) for f in range(N)]
dev = [torch.from_numpy(d.view(np.int16)).cuda() for d in frames]
out = {}
for mode in (1, 0):
    m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH); m.set_option("esdf_mode", mode)
    for f in range(WARM):
        m.recast_depth_to_map(R, T, dev[f], None); m.update_esdf(max_dist=1.0)
    t0 = m.esdf_totals()
    m.enable_profiling(True, only=[_lib.K_ESDF])
    for f in range(WARM, N):
        m.recast_depth_to_map(R, T, dev[f], None); m.update_esdf(max_dist=1.0, wait=False)
    t1 = m.esdf_totals(); m.sync()
    ms, n = m.kernel_time(_lib.K_ESDF)
    i, e = m.export_esdf(); o = np.argsort(lin(i)); out[mode] = (i[o], e[o])
    k = N - WARM
    print(f"esdf_mode {mode}: {ms / n:.3f} ms per update, bricks reached {(t1['region_bricks'] - t0['region_bricks']) / k:.0f}, visits {(t1['brick_relaxations'] - t0['brick_relaxations']) / k:.0f}, "
          f"voxel writes {(t1['voxel_pushes'] - t0['voxel_pushes']) / k:.0f} per update; last {m.esdf_stats()}", flush=True)
print("maps equal:", np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32)))
