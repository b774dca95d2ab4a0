"""Dev-box tool: the ESDF DEFINITIONS of the reference, from the reference's own source.

taichi_slam/mapping/dense_esdf.py cannot be constructed at HEAD (its __init__ fails, its propagation is incomplete, it indexes 4-d
fields with three indices), so the package takes its DEFINITIONS from it -- which voxels are fixed (`is_fixed`, :228-230), what an observed voxel
starts from (`propogate_esdf`, :313-330) and how a value moves to a neighbour (`process_lower_queue`, :275-299).  Here those three functions are
EXECUTED, unmodified, on the sequential Taichi stand-in (tools/ti_seq): the module is loaded by path, an instance is made without its broken
__init__ (object.__new__) and given exactly the attributes the three functions read -- 3-d fields on a pointer / dense tree, the queues, the 26
neighbour vectors built as dense_esdf.py:141-146 builds them.  Scene: an analytic sphere SDF (values rounded to f16, the package's storage type)
over the central 16^3 block of a 48^3 grid -- every voxel of an active block is observed (the reference treats an unobserved voxel of an
active block as a source of distance 0), the blocks around stay inactive (the reference indexes neighbours without a range check).  Committed as tests/golden/ref_esdf_defs.npz:
   init      ESDF after the initialisation branches alone (process_lower_queue stubbed out on the instance)
   one_pass  ESDF after propogate_esdf as written: initialisation + ONE pass over the lower queue, and the queue's order
   fixed     ESDF after the lower-queue pass has been repeated (every voxel queued again, the reference's own insert_lower /
             process_lower_queue) until nothing changes: the fixed point of the reference's relaxation rule
tests/test_ref_esdf.py compares the package (oracle Dijkstra, HIP incremental update) with them.

    python tools/gen_ref_esdf.py            # ~15 minutes (the stand-in interprets every operation); needs /root/reference"""
import importlib.util
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/taichi_slam/mapping"
sys.path.insert(0, os.path.join(ROOT, "tools", "ti_seq")); sys.path.insert(0, ROOT)
N, BLK, VS, RADIUS, MAXRAY = 48, 16, 0.1, 0.45, 0.6
LO, HI = 16, 32                                    # the observed region: the central block, every voxel of it (the blocks around stay inactive: ti.is_active skips them)


def load():
    import taichi as ti
    assert "ti_seq" in ti.__file__
    pkg = types.ModuleType("taichi_slam"); pkg.__path__ = []
    sub = types.ModuleType("taichi_slam.mapping"); sub.__path__ = [REF]; sub.__package__ = "taichi_slam.mapping"
    sys.modules.update({"taichi_slam": pkg, "taichi_slam.mapping": sub})
    out = {}
    for name in ("mapping_common", "dense_esdf"):
        spec = importlib.util.spec_from_file_location(f"taichi_slam.mapping.{name}", os.path.join(REF, name + ".py"))
        mod = importlib.util.module_from_spec(spec); sys.modules[spec.name] = mod; spec.loader.exec_module(mod); out[name] = mod
    return ti, out["dense_esdf"].DenseSDF


def sphere_tsdf():
    """f32 values that are exactly representable in f16: signed distance to a sphere around the grid centre"""
    i = np.arange(LO, HI, dtype=np.float32)
    x, y, z = np.meshgrid(i, i, i, indexing="ij")
    c = np.float32(N / 2 - 0.25)                     # off the lattice: no symmetric ties between neighbours
    d = np.sqrt(((x - c) * np.float32(VS)) ** 2 + ((y - c) * np.float32(VS)) ** 2 + ((z - c) * np.float32(VS)) ** 2).astype(np.float32) - np.float32(RADIUS)
    return d.astype(np.float16).astype(np.float32)


def make(ti, DenseSDF, tsdf):
    o = object.__new__(DenseSDF)                     # the reference's __init__ cannot run (Q18); the functions below read only what is set here
    o.voxel_scale, o.gamma, o.max_ray_length, o.N, o.Nz = VS, VS, MAXRAY, N, N
    o.N_ = ti.Vector([N, N, N], ti.f32)
    Broot = ti.root.pointer(ti.ijk, (N // BLK, N // BLK, N // BLK))
    B = Broot.dense(ti.ijk, (BLK, BLK, BLK))
    o.TSDF, o.ESDF = ti.field(dtype=ti.f32), ti.field(dtype=ti.f32)
    o.observed, o.fixed = ti.field(dtype=ti.i8), ti.field(dtype=ti.i8)
    o.parent_dir = ti.Vector.field(3, dtype=ti.i32)
    B.place(o.TSDF, o.ESDF, o.observed, o.fixed, o.parent_dir)
    Troot = ti.root.pointer(ti.ijk, (N // BLK, N // BLK, N // BLK))
    T = Troot.dense(ti.ijk, (BLK, BLK, BLK))
    o.updated_TSDF = ti.field(dtype=ti.i32)
    T.place(o.updated_TSDF)
    o.B, o.Broot = B, B                              # ti.is_active(self.Broot, ijk): the stand-in keeps a tree's active blocks at the node the fields are placed in
    o.max_queue_size = 4 * N * N * N
    o.raise_queue = ti.Vector.field(3, dtype=ti.i32, shape=o.max_queue_size)
    o.lower_queue = ti.Vector.field(3, dtype=ti.i32, shape=o.max_queue_size)
    o.num_raise_queue, o.num_lower_queue = ti.field(dtype=ti.i32, shape=()), ti.field(dtype=ti.i32, shape=())
    o.head_lower_queue, o.head_raise_queue = ti.field(dtype=ti.i32, shape=()), ti.field(dtype=ti.i32, shape=())
    o.neighbors = []                                 # dense_esdf.py:141-146
    for _di in range(-1, 2):
        for _dj in range(-1, 2):
            for _dk in range(-1, 2):
                if _di != 0 or _dj != 0 or _dk != 0:
                    o.neighbors.append(ti.Vector([_di, _dj, _dk], ti.f32))
    for idx in np.ndindex(HI - LO, HI - LO, HI - LO):
        g = tuple(LO + x for x in idx)
        o.TSDF[g] = float(tsdf[idx]); o.updated_TSDF[g] = 1
    return o


def esdf_of(o):
    out = np.zeros((HI - LO,) * 3, np.float32)
    for idx in np.ndindex(HI - LO, HI - LO, HI - LO):
        v = o.ESDF[tuple(LO + x for x in idx)]
        out[idx] = np.float32(getattr(v, "v", v))
    return out


if __name__ == "__main__":
    assert os.path.exists(REF), "the reference tree is needed"
    ti, DenseSDF = load()
    tsdf = sphere_tsdf()

    @ti.kernel
    def propagate(o: ti.template()):
        o.propogate_esdf()

    @ti.kernel
    def queue_all_and_lower(o: ti.template()):
        o.num_lower_queue[None] = 0
        o.head_lower_queue[None] = 0
        for i, j, k in o.updated_TSDF:
            o.insert_lower(ti.Vector([i, j, k], ti.i32))
        o.process_lower_queue()

    t0 = time.time()
    a = make(ti, DenseSDF, tsdf)
    a.process_lower_queue = lambda: None             # the initialisation branches alone
    a.process_raise_queue = lambda: None
    propagate(a)
    init = esdf_of(a)
    nq = int(a.num_lower_queue[None]); nr = int(a.num_raise_queue[None])
    print(f"init: {time.time() - t0:.0f} s, lower queue {nq}, raise queue {nr}")
    b = make(ti, DenseSDF, tsdf)
    propagate(b)                                     # as written: initialisation, (empty) raise queue, ONE pass over the lower queue
    one = esdf_of(b)
    qn = int(b.num_lower_queue[None])
    queue = np.array([[int(getattr(c, "v", c)) for c in b.lower_queue[t]] for t in range(qn)], np.int16)
    print(f"one pass: {time.time() - t0:.0f} s, queue {qn}")
    cur, passes = one, 1
    while True:
        queue_all_and_lower(b)
        nxt = esdf_of(b); passes += 1
        changed = int((nxt != cur).sum())
        print(f"pass {passes}: {changed} voxels changed, {time.time() - t0:.0f} s")
        cur = nxt
        if changed == 0:
            break
    path = os.path.join(ROOT, "tests", "golden", "ref_esdf_defs.npz")
    np.savez_compressed(path, tsdf=tsdf, init=init, one_pass=one, fixed=cur, queue=queue, passes=np.int32(passes),
                        params=np.array([N, BLK, VS, RADIUS, MAXRAY, LO, HI], np.float64))
    print(f"-> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
