"""Developer probe: why host images out of a torch-pinned tensor integrate SLOWER than pageable numpy images over 200 cache-cold frames.
Times the host side alone: reading every second pixel of every second row of 200 distinct 640 x 480 u16 images (what tsl_tsdf_integrate_depth's pick does) from
pageable arrays and from slices of one pinned tensor, and prints the NUMA node the pages of each live on (/proc/self/numa_maps) and the node this thread runs on."""
import os, re, time, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def node_of(addr):
    best = None
    for line in open("/proc/self/numa_maps"):
        a = int(line.split()[0], 16)
        if a <= addr and (best is None or a > best[0]):
            best = (a, line)
    if not best: return "?"
    return " ".join(x for x in best[1].split() if re.match(r"N\d+=\d+", x) or x.startswith("bind") or x.startswith("prefer") or x in ("default", "interleave"))

def cpu_node():
    try:
        cpu = os.sched_getcpu()
        for n in os.listdir("/sys/devices/system/node"):
            if n.startswith("node") and os.path.exists(f"/sys/devices/system/node/{n}/cpu{cpu}"):
                return f"cpu {cpu} on {n}"
        return f"cpu {cpu}"
    except Exception as e:
        return repr(e)

N = 200
rng = np.random.default_rng(0)
page = [rng.integers(300, 5000, size=(480, 640), dtype=np.uint16) for _ in range(N)]
pin_t = torch.from_numpy(np.stack(page).view(np.int16)).pin_memory()
pinned = pin_t.numpy().view(np.uint16)
dst = np.empty((240, 320), np.uint16)
print("thread:", cpu_node(), "| nodes:", sorted(n for n in os.listdir("/sys/devices/system/node") if n.startswith("node")))
print("pageable image 0 pages:", node_of(page[0].ctypes.data))
print("pinned tensor pages   :", node_of(pinned.ctypes.data))
for label, src in (("pageable", page), ("pinned", [pinned[i] for i in range(N)])):
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(N):
            np.copyto(dst, src[i][::2, ::2])
        dt = time.perf_counter() - t0
        print(f"{label}: pick of the visited pixels {1e6 * dt / N:.1f} us per image (pass {rep})")
    t0 = time.perf_counter()
    s = 0
    for i in range(N):
        s += int(src[i].sum(dtype=np.uint64))
    print(f"{label}: full read {1e6 * (time.perf_counter() - t0) / N:.1f} us per image")
# the library itself, host time per call
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
frames = list(syn.sphere_room_stream(N))
pg = [d for _, _, d in frames]
pt = torch.from_numpy(np.stack(pg).view(np.int16)).pin_memory().numpy().view(np.uint16)
for label, src in (("pageable", pg), ("pinned", [pt[i] for i in range(N)])):
    m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
    for i in range(40): m.recast_depth_to_map(frames[i][0], frames[i][1], src[i], None)
    m.sync()
    for rep in range(2):
        t0 = time.perf_counter(); th = 0.0
        for i in range(300):
            a = time.perf_counter(); m.recast_depth_to_map(frames[i % N][0], frames[i % N][1], src[i % N], None); th += time.perf_counter() - a
        m.sync(); dt = time.perf_counter() - t0
        print(f"library, {label}: {300 / dt:.0f} frames/s, host {1e6 * th / 300:.1f} us per call")
