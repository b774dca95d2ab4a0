"""Developer probe: per-workgroup part timeline of k_integrate_bricks3 (TSL_TIMING build: TSL_LIB=.../libtaichislam_hip_timing.so)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taichislam_amd import _lib
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH); m.set_option("ramp", 0)
for a in sys.argv[1:]:
    k, v = a.split("="); m.set_option(k, int(v))
frames = list(syn.sphere_room_stream(14))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
L = _lib.lib(); L.tsl_tsdf_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
out = np.zeros((16384, 16), np.int64)
NF = 16
for i in range(16): m.recast_depth_to_map(frames[i % 14][0], frames[i % 14][1], dev[i % 14], None)
m.sync()
L.tsl_tsdf_debug_counters(m.h, out.ctypes.data_as(ctypes.c_void_p), 1)
NB = int(os.environ.get("NB", "8"))
for i in range(NB): m.recast_depth_to_map(frames[(4 + i) % 14][0], frames[(4 + i) % 14][1], dev[(4 + i) % 14], None)
m.sync()
L.tsl_tsdf_debug_counters(m.h, out.ctypes.data_as(ctypes.c_void_p), 1)
# per workgroup 16 item records of 8: [0] start [1] unit | np << 8 | frame mask << 32 | kdep << 40 [2] segments walked [3] end of the last walk [4] item end [5] end of the wait for the previous frame (0: none)
rec = out.reshape(-1)[131072:131072 + 512 * 128].reshape(512, 16, 8)
used = rec[:, :, 0] != 0
t0 = rec[:, :, 0][used].min()
nitems = used.sum(1)
print("workgroups with work", (nitems > 0).sum(), "items", used.sum(), "items per wg histogram", np.bincount(nitems))
end = np.where(used, rec[:, :, 4], 0).max(1)
print("wg end time us: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile((end[nitems > 0] - t0) / 100.0, [10, 50, 90, 99, 100])))
order = np.argsort(-end)
print("slowest / fastest workgroups: wg | per item: U|P frames-mask segs np kdep @start walk(to the end of the last walk) apply(wait)")
for w in list(order[:8]) + list(order[-3:]):
    s = f"  wg {w:3d} |"
    for t in range(16):
        if not used[w, t]: break
        r = rec[w, t]
        wt = (r[5] - r[3]) / 100 if r[5] else 0.0
        s += f" [{'U' if r[1] & 1 else 'P'}{(r[1] >> 32) & 15:x} {r[2]:5d} np{(r[1] >> 8) & 0xfff:2d} @{(r[0]-t0)/100:5.1f} w{(r[3]-r[0])/100:5.1f} a{(r[4]-r[3])/100:5.1f}({wt:.1f})]"
    print(s)
isu = (rec[:, :, 1] & 1)[used] == 1; ns = rec[:, :, 2][used]; npp = ((rec[:, :, 1] >> 8) & 0xfff)[used]
walk = (rec[:, :, 3] - rec[:, :, 0])[used] / 100.0; fl = (rec[:, :, 4] - rec[:, :, 3])[used] / 100.0; st = (rec[:, :, 0][used] - t0) / 100.0
wait = np.where(rec[:, :, 5] != 0, rec[:, :, 5] - rec[:, :, 3], 0)[used] / 100.0
print(f"  units {isu.sum()}  parts {(~isu).sum()} (split {(npp[~isu] > 1).sum()})  late waits {np.count_nonzero(wait > 0)} mean {wait[wait > 0].mean() if (wait > 0).any() else 0:.2f} us")
for kind, selk in (("unit", isu), ("part whole", (~isu) & (npp == 1)), ("part split", (~isu) & (npp > 1))):
    for lo, hi in ((0, 256), (256, 1024), (1024, 2048), (2048, 4097), (4097, 1 << 30)):
        sel = selk & (ns >= lo) & (ns < hi)
        if sel.any():
            print(f"  {kind:10s} segs [{lo},{hi}): {sel.sum():4d} items  start p50 {np.median(st[sel]):5.1f}  walk mean {walk[sel].mean():5.1f} p90 {np.percentile(walk[sel],90):5.1f}  apply mean {fl[sel].mean():5.1f} p90 {np.percentile(fl[sel],90):5.1f} max {fl[sel].max():5.1f}")
busy = np.where(used, rec[:, :, 4] - rec[:, :, 0], 0).sum(1) / 100.0
print("busy us per workgroup: mean %.1f min %.1f max %.1f; launch span %.1f us for the last batch" % (busy.mean(), busy.min(), busy.max(), (end.max() - t0) / 100.0))
ph = out.reshape(-1)[196608:196608 + 512 * 128].reshape(512, 16, 8)
for kind, selk in (("unit", isu), ("part", ~isu)):
    for lo, hi in ((0, 256), (256, 1024), (1024, 2048), (2048, 4097)):
        sel = selk & (ns >= lo) & (ns < hi)
        if sel.any():
            a = ph[used][sel] / 100.0
            print(f"  {kind} segs [{lo},{hi}): steps/item {ph[used][sel][:, 4].mean():.1f}  per item us: top+walk {a[:, 0].mean():5.1f}  sort {a[:, 1].mean():5.1f}  apply {a[:, 2].mean():5.1f}  tail {a[:, 3].mean():5.1f}")
