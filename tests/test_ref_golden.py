"""Golden vectors produced by the REFERENCE'S OWN SOURCE (tests/golden/ref_*.npz, tools/gen_ref_golden.py: taichi_slam/mapping/dense_tsdf.py +
mapping_common.py imported unmodified and executed, one loop iteration after the other, on the sequential Taichi stand-in of tools/ti_seq).

  * CPU: the oracle's FAITHFUL mode -- the reference-literal sequential replay -- reproduces every vector bit for bit.  This is what pins the
    oracle: its restatement of the reference's control flow, index arithmetic, type promotion, rounding points and struct-for order agrees with
    the reference's text as executed by an independent interpreter (what stays an assumption of both is Taichi's own back-end behaviour: f16
    arithmetic through f32, ti.round half away from zero, the cast in front of an atomic add -- tools/ti_seq/taichi/__init__.py header).
  * GPU: the HIP path with semantics = 1 reproduces the integration vectors bit for bit; the default (order-free) HIP path gives the same voxel
    set, the same occupancy and values within the deviation BASELINE.md section 5 reports for it."""
import glob
import json
import os

import numpy as np
import pytest

from util import lin

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["depth_stream", "point_clouds", "textured", "textured_points", "weight_clamp", "two_submaps_fused", "aligned_submap_fused"]
FUSED = ("two_submaps_fused", "aligned_submap_fused")
# the same two submaps with num_voxel_per_blk_axis = 10, the block size of the reference's own configuration: volumes of 70 / 130 voxels per axis (not
# multiples of the 16^3 storage bricks), a sensor grid of 10^3 blocks (another ray order), a struct-for over 10^3 blocks in fuse_submaps (another splat order)
BLK10 = ["blk10_two_submaps", "blk10_two_submaps_fused"]
# a submap with a tilted base pose, a display window, recast_step = 3 on an image of 61 rows, internal_voxels = 3, voxels of 12 cm: map, particle exports
# (positions through the submap's pose), ray casts, meshes at step 1 and 2
POSED = ["posed_exports"]


def load(name):
    z = np.load(os.path.join(GOLD, f"ref_{name}.npz"))
    cfg = json.loads(str(z["cfg"]))
    steps = json.loads(str(z["steps"]))
    for n, s in enumerate(steps):
        for k in list(s):
            if s[k] is None and f"s{n}_{k}" in z.files:
                s[k] = z[f"s{n}_{k}"]
    out = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    return cfg, z["K"], (z["Kc"] if bool(z["has_Kc"]) else None), steps, out


def sorted_bits(e):
    o = np.argsort(lin(np.asarray(e["indices"])), kind="stable")
    out = {"indices": np.asarray(e["indices"])[o].astype(np.int16), "TSDF": np.asarray(e["TSDF"])[o].view(np.uint16), "W_TSDF": np.asarray(e["W_TSDF"])[o].view(np.uint16),
           "occupy": np.asarray(e["occupy"])[o].astype(np.int8)}
    if getattr(e.get("color", None), "size", 0):
        out["color"] = np.asarray(e["color"])[o].view(np.uint16)
    return out


def replay(make, steps, K, Kc, mode_kw, fuse):
    """Drives an implementation (oracle or HIP shim) through the recorded steps.  `make(cfg_overrides)` builds a map."""
    m = make({})
    g = None
    for s in steps:
        if s["kind"] == "base":
            m.set_base_pose_submap(s["sid"], np.ascontiguousarray(s["R"], dtype=np.float64), np.ascontiguousarray(s["T"], dtype=np.float64))
        elif s["kind"] == "depth":
            m.integrate("depth", s, **mode_kw)
        elif s["kind"] == "pcl":
            m.integrate("pcl", s, **mode_kw)
        elif s["kind"] == "next_submap":
            m.next_submap()
        elif s["kind"] == "select":
            m.select(s["sid"])
        elif s["kind"] == "fuse":
            g = make({"is_global_map": True, "map_scale": s["global_map_scale"]})
            for b in steps:
                if b["kind"] == "base":
                    g.set_base_pose_submap(b["sid"], np.ascontiguousarray(b["R"], dtype=np.float64), np.ascontiguousarray(b["T"], dtype=np.float64))
            fuse(g, m)
    return sorted_bits((g if g is not None else m).export())


def assert_bits_equal(got, want, what):
    assert got["indices"].shape == want["indices"].shape, f"{what}: {got['indices'].shape[0]} voxels, the reference has {want['indices'].shape[0]}"
    assert np.array_equal(got["indices"], want["indices"]), f"{what}: voxel sets differ"
    for k in ("TSDF", "W_TSDF", "occupy") + (("color",) if "color" in want else ()):
        bad = np.nonzero(np.atleast_1d((got[k] != want[k]).reshape(got[k].shape[0], -1).any(axis=1)))[0]
        assert bad.size == 0, f"{what}: {k} differs at {bad.size} of {want[k].shape[0]} voxels, first {want['indices'][bad[0]]}: {got[k][bad[0]]} vs {want[k][bad[0]]}"


# ------------------------------------------------------------------------------------------------------------------ oracle (CPU)
class _Ora:
    def __init__(self, cfg, K, Kc):
        from oracle import OracleTSDF
        self.o = OracleTSDF(**cfg)
        self.o.set_intrinsics(K, Kc if Kc is not None else K)

    def set_base_pose_submap(self, sid, R, T): self.o.set_base_pose_submap(sid, R, T)
    def next_submap(self): self.o.set_active_submap(self.o.get_active_submap() + 1)
    def select(self, sid): self.o.set_active_submap(sid)
    def export(self): return self.o.export_sparse()

    def integrate(self, kind, s, mode):
        if kind == "depth":
            self.o.integrate_depth(s["R"], s["T"], s["depth"], s.get("texture"), mode=mode)
        else:
            self.o.integrate_points(s["R"], s["T"], s["xyz"], s.get("rgb"), mode=mode)


@pytest.mark.parametrize("name", NAMES + BLK10 + POSED)
def test_oracle_faithful_reproduces_the_reference_source_bit_for_bit(name):
    from oracle import FAITHFUL
    cfg, K, Kc, steps, want = load(name)
    got = replay(lambda over: _Ora({**cfg, **over}, K, Kc), steps, K, Kc, {"mode": FAITHFUL}, lambda g, m: g.o.fuse_submaps(m.o, mode=FAITHFUL))
    assert want["indices"].shape[0] > 1000
    assert_bits_equal(got, want, f"oracle FAITHFUL vs reference source, {name}")


def test_the_vectors_cover_what_they_claim():
    """weights reach the clamp, colours are present, several rays share voxels, the fused map is larger than either submap's footprint"""
    _, _, _, _, w = load("weight_clamp")
    assert (w["W_TSDF"].view(np.float16) == np.float16(1000.0)).sum() >= 20 and (w["W_TSDF"].view(np.float16) < np.float16(1000.0)).sum() > 50
    _, _, _, _, t = load("textured")
    assert "color" in t and (t["color"] != 0).any()
    _, _, _, steps, p = load("point_clouds")
    assert p["indices"].shape[0] > 10000 and sum(s["kind"] == "pcl" for s in steps) == 2
    assert sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLD, "ref_*.npz"))) == sorted(f"ref_{n}.npz" for n in NAMES + BLK10 + POSED + ["octomap", "session", "session_blk10", "esdf_defs"])


# ------------------------------------------------------------------------------------------------------------------ HIP (GPU)
class _Hip:
    def __init__(self, cfg, K, Kc, semantics):
        from taichislam_amd.mapping import DenseTSDF
        self.m = DenseTSDF(**cfg)
        self.m.set_dep_camera_intrinsic(K)
        self.m.set_color_camera_intrinsic(Kc if Kc is not None else K)
        if semantics:
            self.m.set_option("semantics", semantics)

    def set_base_pose_submap(self, sid, R, T): self.m.set_base_pose_submap(sid, R, T)
    def next_submap(self): self.m.switch_to_next_submap()
    def select(self, sid): self.m.active_submap_id[None] = sid
    def export(self): return self.m.export_submap()

    def integrate(self, kind, s):
        if kind == "depth":
            self.m.recast_depth_to_map(s["R"], s["T"], s["depth"], s.get("texture"))
        else:
            self.m.recast_pcl_to_map(s["R"], s["T"], s["xyz"], s["rgb"] if s.get("rgb") is not None else np.array([]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["depth_stream", "point_clouds", "weight_clamp", "textured", "textured_points", "blk10_two_submaps"])
def test_hip_sequential_mode_reproduces_the_reference_source_bit_for_bit(hip_lib, name):
    cfg, K, Kc, steps, want = load(name)
    got = replay(lambda over: _Hip({**cfg, **over}, K, Kc, 1), steps, K, Kc, {}, None)
    assert_bits_equal(got, want, f"HIP semantics = 1 vs reference source, {name}")


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_default_path_against_the_reference_source(hip_lib, name):
    """The order-free default: the same voxels observed, the same occupancy; TSDF within the band BASELINE.md section 5 measures against the literal
    replay (per-frame sums applied once instead of ray by ray in f16: a few f16 ulp per frame)."""
    cfg, K, Kc, steps, want = load(name)
    got = replay(lambda over: _Hip({**cfg, **over}, K, Kc, 0), steps, K, Kc, {}, lambda g, m: g.m.fuse_submaps(m.m))
    assert np.array_equal(got["indices"], want["indices"]), "voxel sets differ"
    assert np.array_equal(got["occupy"], want["occupy"])
    t_g, t_w = got["TSDF"].view(np.float16).astype(np.float64), want["TSDF"].view(np.float16).astype(np.float64)
    # An axis-aligned base pose makes six of the seven splat weights exactly 0.  In the reference's fusion a voxel whose FIRST splat has weight 0 is
    # 0 / 0 and stays NaN whatever comes later (58 % of aligned_submap_fused); the order-free sums are NaN only where EVERY splat has weight 0.
    fin = np.isfinite(t_g) & np.isfinite(t_w)
    assert not (np.isnan(t_g) & ~np.isnan(t_w)).any(), "a voxel that is NaN in the order-free fusion is NaN in the literal one"
    assert fin.all() if name != "aligned_submap_fused" else (0.3 < fin.mean() < 0.6 and np.isnan(t_w).mean() > 0.5)
    d = np.abs(t_g - t_w)[fin]
    vs = cfg["voxel_scale"]
    assert np.percentile(d, 50) <= 0.02 * vs and np.percentile(d, 99) <= 0.6 * vs, np.percentile(d, [50, 99])
    assert d.max() <= 4.0 * vs or name == "aligned_submap_fused", d.max()      # (there the literal fusion also divides by sums that underflow: single outliers)
    w_g, w_w = got["W_TSDF"].view(np.float16).astype(np.float64), want["W_TSDF"].view(np.float16).astype(np.float64)
    assert np.percentile((np.abs(w_g - w_w) / np.maximum(w_w, 1e-3))[fin], 99) <= 0.02


# ------------------------------------------------------------------------------------------------------------------ Octomap (taichi_octomap.py)
def _octo_replay(o, steps, depth_fn, pcl_fn):
    for s in steps:
        if s["kind"] == "base":
            o.set_base_pose_submap(s["sid"], np.ascontiguousarray(s["R"], dtype=np.float64), np.ascontiguousarray(s["T"], dtype=np.float64))
        elif s["kind"] == "depth":
            depth_fn(s)
        elif s["kind"] == "pcl":
            pcl_fn(s)
    idx, cnt = o.export_leaves()
    order = np.argsort(lin(idx), kind="stable")
    return idx[order].astype(np.int16), cnt[order].astype(np.float32)


def test_oracle_octomap_reproduces_the_reference_source_bit_for_bit():
    from oracle import OracleOctomap
    cfg, K, _, steps, want = load("octomap")
    o = OracleOctomap(**cfg)
    o.set_intrinsics(K, K)
    idx, cnt = _octo_replay(o, steps, lambda s: o.integrate_depth(s["R"], s["T"], s["depth"], None), lambda s: o.integrate_points(s["R"], s["T"], s["xyz"], None))
    assert want["indices"].shape[0] > 1000 and want["occupy"].max() > 5
    assert np.array_equal(idx, want["indices"]) and np.array_equal(cnt.view(np.uint32), want["occupy"].view(np.uint32))


@pytest.mark.parametrize("textured", [False, True])
def test_oracle_octomap_exports_reproduce_the_reference_source(textured):
    """cvt_occupy_to_voxels(level) of the reference (taichi_octomap.py:90-102; the node calls it with level 0) on a submap with a tilted base pose: the leaves above
    min_occupy_thres, positions through the pose, colours of the textured tree -- the same rows, bit for bit, at level 0 and 1."""
    from oracle import OracleOctomap
    cfg, K, _, steps, want = load("octomap")
    o = OracleOctomap(**{**cfg, "texture_enabled": textured})
    o.set_intrinsics(K, K)
    o.set_base_pose_submap(0, want["posed_R"], want["posed_T"])
    for s in steps[1:3]:
        o.integrate_depth(s["R"], s["T"], s["depth"], want["tex_texture"] if textured else None)
    for level in (0, 1):
        if textured:
            xyz, rgb, n = o.occupied_voxels(level, with_color=True)
            got = _rows(xyz, rgb)
        else:
            xyz, n = o.occupied_voxels(level)
            got = _rows(xyz)
        w = want[("posedtex" if textured else "posed") + f"_export{level}"]
        assert n == w.shape[0] > 500 and np.array_equal(got.view(np.uint32), w.view(np.uint32)), level


@pytest.mark.gpu
def test_hip_octomap_reproduces_the_reference_source_bit_for_bit(hip_lib):
    from taichislam_amd.mapping import Octomap
    cfg, K, _, steps, want = load("octomap")
    o = Octomap(**cfg)
    o.set_dep_camera_intrinsic(K)
    idx, cnt = _octo_replay(o, steps, lambda s: o.recast_depth_to_map(s["R"], s["T"], s["depth"], None), lambda s: o.recast_pcl_to_map(s["R"], s["T"], s["xyz"], None, s["xyz"].shape[0]))
    assert np.array_equal(idx, want["indices"]) and np.array_equal(cnt.view(np.uint32), want["occupy"].view(np.uint32))


# ------------------------------------------------------------------------------------------------------------------ marching cubes (marching_cube_mesher.py)
def _canon(v, n):
    tri = np.concatenate([np.asarray(v, np.float32).reshape(-1, 9), np.asarray(n, np.float32).reshape(-1, 9)], axis=1)
    return tri[np.lexsort(tri.view(np.uint32).T[::-1])]


def test_oracle_mesh_reproduces_the_reference_source_bit_for_bit():
    """generate_mesh(1) of the reference's mesher on the reference's map == the oracle's marching cubes on the oracle's FAITHFUL map: the same
    triangles, vertices and normals bit for bit.  (This comparison found the one restatement slip so far: `valp2 - valp1` in vertexInterp is an f16
    operation; the oracle and the HIP kernel had taken the difference in f32 -- vertices off by up to 3e-5 voxels.)"""
    from oracle import FAITHFUL, OracleTSDF
    cfg, K, _, steps, want = load("depth_stream")
    o = OracleTSDF(**cfg)
    o.set_intrinsics(K, K)
    for s in steps:
        if s["kind"] == "base":
            o.set_base_pose_submap(s["sid"], s["R"], s["T"])
        else:
            o.integrate_depth(s["R"], s["T"], s["depth"], None, mode=FAITHFUL)
    v, n, _, cnt = o.generate_mesh(1, float(want["mesh_thres"]), 20000)
    assert cnt == want["mesh"].shape[0] > 1000
    assert np.array_equal(_canon(v, n).view(np.uint32), want["mesh"].view(np.uint32))


def _hip_map(name, **extra):
    from taichislam_amd.mapping import DenseTSDF
    cfg, K, Kc, steps, want = load(name)
    m = DenseTSDF(**cfg, **extra)
    m.set_dep_camera_intrinsic(K)
    m.set_color_camera_intrinsic(Kc if Kc is not None else K)
    m.set_option("semantics", 1)
    for s in steps:
        if s["kind"] == "base":
            m.set_base_pose_submap(s["sid"], s["R"], s["T"])
        else:
            m.recast_depth_to_map(s["R"], s["T"], s["depth"], s.get("texture"))
    return cfg, m, want


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["depth_stream", "textured"])
def test_hip_mesh_reproduces_the_reference_source_bit_for_bit(hip_lib, name):
    from taichislam_amd.mapping import MarchingCubeMesher
    cfg, m, want = _hip_map(name)
    me = MarchingCubeMesher(m, 20000, tsdf_surface_thres=float(want["mesh_thres"]))
    me.generate_mesh(1)
    v, n, c = me.get_mesh()
    assert me.num_facelets[None] == want["mesh"].shape[0]
    cols = [v.reshape(-1, 9), n.reshape(-1, 9)] + ([c.reshape(-1, 9)] if cfg.get("texture_enabled") else [])
    assert np.array_equal(_rows(*cols).view(np.uint32), want["mesh"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["depth_stream", "textured"])
def test_hip_coarse_mesh_and_the_gather_kernel_reproduce_the_reference_source(hip_lib, name):
    """generate_mesh(2) -- cubes of edge 2 anchored at every voxel, corners up to two voxels beyond the brick: every value is read through the brick
    table (k_marching_cubes) -- and the same kernel at step 1 (option mesh_gather) against the reference's own meshes."""
    from taichislam_amd.mapping import MarchingCubeMesher
    cfg, m, want = _hip_map(name)
    for step, gather, key in ((2, 0, "mesh_step2"), (1, 1, "mesh")):
        m.set_option("mesh_gather", gather)
        me = MarchingCubeMesher(m, 60000, tsdf_surface_thres=float(want["mesh_thres"]))
        me.generate_mesh(step)
        v, n, c = me.get_mesh()
        assert me.num_facelets[None] == want[key].shape[0], (step, gather)
        cols = [v.reshape(-1, 9), n.reshape(-1, 9)] + ([c.reshape(-1, 9)] if cfg.get("texture_enabled") else [])
        assert np.array_equal(_rows(*cols).view(np.uint32), want[key].view(np.uint32)), (step, gather)


# ------------------------------------------------------------------------------------------------------------------ exports, raycast, coloured mesh
def _rows(*cols):
    t = np.concatenate([np.asarray(c, np.float32).reshape(len(c), -1) for c in cols], axis=1).astype(np.float32)
    return t[np.lexsort(t.view(np.uint32).T[::-1])]


def _oracle_map(name):
    from oracle import FAITHFUL, OracleTSDF
    cfg, K, Kc, steps, want = load(name)
    o = OracleTSDF(**cfg)
    o.set_intrinsics(K, Kc if Kc is not None else K)
    for s in steps:
        if s["kind"] == "base":
            o.set_base_pose_submap(s["sid"], s["R"], s["T"])
        else:
            o.integrate_depth(s["R"], s["T"], s["depth"], s.get("texture"), mode=FAITHFUL)
    return cfg, o, want


@pytest.mark.parametrize("name", ["depth_stream", "textured"] + POSED)
def test_oracle_exports_raycast_and_mesh_reproduce_the_reference_source(name):
    """On the reference's map: cvt_TSDF_surface_to_voxels (positions + colours: the jet colour map of matplotlib, or the stored colours), cvt_TSDF_to_voxels_slice,
    BaseMap.raycast for 24 rays, and generate_mesh(1) with vertex colours -- everything bit for bit, as sets of rows."""
    cfg, o, want = _oracle_map(name)
    xyz, rgb, n = o.surface_voxels()
    assert n == want["surface"].shape[0] > 1000 and np.array_equal(_rows(xyz, rgb).view(np.uint32), want["surface"].view(np.uint32))
    z, dz = (float(x) for x in want["slice_args"])
    xyz, val, rgb, n = o.slice_voxels(z, dz)
    assert n == want["slice"].shape[0] > 20 and np.array_equal(_rows(xyz, val, rgb).view(np.uint32), want["slice"].view(np.uint32))
    hit, end, ln = o.raycast(want["ray_pos"], want["ray_dir"], float(want["ray_max"]))
    assert np.array_equal(hit, want["ray_hit"]) and np.array_equal(end.view(np.uint32), want["ray_end"].view(np.uint32)) and np.array_equal(ln.view(np.uint32), want["ray_len"].view(np.uint32))
    v, nr, col, cnt = o.generate_mesh(1, float(want["mesh_thres"]), 20000)
    cols = [v.reshape(-1, 9), nr.reshape(-1, 9)] + ([col.reshape(-1, 9)] if cfg.get("texture_enabled") else [])
    assert cnt == want["mesh"].shape[0] > 500 and np.array_equal(_rows(*cols).view(np.uint32), want["mesh"].view(np.uint32))
    # the planner's point queries: is_pos_occupy, is_pos_unobserved, is_near_pos_occupy(xyz, 2) for 200 points (mapping_common.py:178-201)
    qp = want["q_pos"]
    assert 50 < want["q_occ"].sum() < 190 and 30 < want["q_unobs"].sum() < 150
    assert np.array_equal(o.query_points(0, qp), want["q_occ"]) and np.array_equal(o.query_points(1, qp), want["q_unobs"]) and np.array_equal(o.query_points(2, qp, 2), want["q_near2"])
    # generate_mesh(2): every voxel below the threshold anchors a cube of edge 2 (the anchors are not thinned out: overlapping cubes, more triangles than at step 1)
    v, nr, col, cnt = o.generate_mesh(2, float(want["mesh_thres"]), 60000)
    cols = [v.reshape(-1, 9), nr.reshape(-1, 9)] + ([col.reshape(-1, 9)] if cfg.get("texture_enabled") else [])
    assert cnt == want["mesh_step2"].shape[0] > want["mesh"].shape[0] and np.array_equal(_rows(*cols).view(np.uint32), want["mesh_step2"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["depth_stream", "textured"])
def test_hip_exports_and_raycast_reproduce_the_reference_source(hip_lib, name):
    """The HIP map of semantics = 1 is the reference's map bit for bit (colours included); so are its particle exports and ray casts."""
    cfg, m, want = _hip_map(name, max_disp_particles=40000)
    m.cvt_TSDF_surface_to_voxels()
    n = m.num_TSDF_particles[None]
    xyz, rgb, _ = m._read_exports(n)
    assert n == want["surface"].shape[0] and np.array_equal(_rows(xyz, rgb).view(np.uint32), want["surface"].view(np.uint32))
    z, dz = (float(x) for x in want["slice_args"])
    m.cvt_TSDF_to_voxels_slice(z, dz)
    n = m.num_TSDF_particles[None]
    xyz, rgb, val = m._read_exports(n)
    assert n == want["slice"].shape[0] and np.array_equal(_rows(xyz, val, rgb).view(np.uint32), want["slice"].view(np.uint32))
    hit, end, ln = m.raycast(want["ray_pos"], want["ray_dir"], float(want["ray_max"]))
    assert np.array_equal(np.asarray(hit, bool), want["ray_hit"]) and np.array_equal(np.asarray(end, np.float32).view(np.uint32), want["ray_end"].view(np.uint32))
    assert np.array_equal(np.asarray(ln, np.float32).view(np.uint32), want["ray_len"].view(np.uint32))


# ------------------------------------------------------------------------------------------------------------------ a whole session (submap_mapping.py)
class OraMap:
    """The oracle behind the slice of the DenseTSDF surface that SubmapMapping drives (FAITHFUL everywhere)."""

    def __init__(self, **kw):
        from oracle import OracleTSDF
        kw = dict(kw)
        self.max_disp_particles = kw.pop("max_disp_particles", 1 << 20)
        self.kw = kw
        self.o = OracleTSDF(**kw)
        self.enable_texture = bool(kw.get("texture_enabled", False))
        self.max_submap_num = kw.get("max_submap_num", 1024)
        self.remote = 0
        self.export_color = self.export_TSDF_xyz = self.num_TSDF_particles = None
        self.clear_last_TSDF_exporting = False

    def set_dep_camera_intrinsic(self, K): self.o.set_intrinsics(K, K)
    def set_color_camera_intrinsic(self, K): pass
    def get_active_submap_id(self): return self.o.get_active_submap()
    def set_base_pose_submap(self, sid, R, T): self.o.set_base_pose_submap(int(sid), np.asarray(R, np.float64), np.asarray(T, np.float64))
    def saveMap(self, filename): pass
    def finalization_current_submap(self): pass

    def switch_to_next_submap(self):
        self.o.set_active_submap(self.o.get_active_submap() + 1)
        return self.o.get_active_submap()

    def recast_depth_to_map(self, R, T, depth, texture):
        from oracle import FAITHFUL
        self.o.integrate_depth(R, T, depth, None, mode=FAITHFUL)

    def fuse_submaps(self, submaps):
        from oracle import FAITHFUL
        self.o.fuse_submaps(submaps.o, mode=FAITHFUL)

    def export_submap(self):
        e = self.o.export_sparse()
        return {"indices": e["indices"], "TSDF": e["TSDF"], "W_TSDF": e["W_TSDF"], "color": np.array([]), "occupy": e["occupy"],
                "map_scale": [self.kw["map_scale"][0], self.kw["map_scale"][1]], "voxel_scale": self.kw["voxel_scale"], "texture_enabled": False,
                "num_voxel_per_blk_axis": self.kw.get("num_voxel_per_blk_axis", 16)}

    def input_remote_submap(self, submap):                     # dense_tsdf.py:499-516
        self.remote += 1
        idx = self.max_submap_num - self.remote
        self.o.import_sparse(idx, submap["indices"], submap["TSDF"], submap["W_TSDF"], submap["occupy"])
        self.set_base_pose_submap(idx, *submap["pose"])
        return idx


def _session(map_cls, SM, name="session"):
    import submap_trace as st
    z = np.load(os.path.join(GOLD, f"ref_{name}.npz"))
    p = json.loads(str(z["params"]))
    saved = st.H, st.W, st.OPTS
    st.H, st.W, st.OPTS = p["H"], p["W"], dict(p["OPTS"])
    try:
        assert (st.NFRAMES, st.KEYFRAME_STEP) == (p["NFRAMES"], p["KEYFRAME_STEP"])
        sm, sent = st.drive(SM, map_cls)
        smb = SM(map_cls, keyframe_step=st.KEYFRAME_STEP, sub_opts=dict(st.OPTS), global_opts=dict(st.OPTS))
        smb.map_send_handle = smb.traj_send_handle = lambda b: None
        for buf in sent:
            smb.input_remote_submap(buf)
    finally:
        st.H, st.W, st.OPTS = saved
    return z, sm.global_map, smb.global_map


@pytest.mark.parametrize("name", ["session", "session_blk10"])
def test_the_reference_orchestration_on_the_reference_maps_is_reproduced_by_the_package_on_the_oracle(name):
    """tests/golden/ref_session.npz (and ref_session_blk10.npz: the same with blocks of 10 voxels, the reference's own configuration): the reference's submap_mapping.py driving the reference's DenseTSDF (all of it run on tools/ti_seq) through eight frames,
    three submaps, a pose-graph update, local_to_global, and a second agent fed from the wire.  The package's SubmapMapping on the FAITHFUL oracle ends with
    the same two global maps, bit for bit: poses through convert_by_pgo / convert_by_base, the export -> zlib -> load_numpy path, slots from the top for
    remote submaps, fusion in struct-for order."""
    from taichislam_amd.mapping.submap_mapping import SubmapMapping
    z, ga, gb = _session(OraMap, SubmapMapping, name)
    for tag, g in (("A", ga), ("B", gb)):
        got = sorted_bits(g.export_submap())
        want = {k[2:]: z[k] for k in z.files if k.startswith(tag + "_")}
        assert want["indices"].shape[0] > 20000
        assert_bits_equal(got, want, f"agent {tag}'s global map")


@pytest.mark.gpu
def test_the_reference_session_on_the_hip_maps(hip_lib):
    """The same session through the package's SubmapMapping on the HIP maps (default, order-free path): both agents' global maps hold exactly the reference's
    voxels; values within the band of the order-free sums against the literal replay."""
    from taichislam_amd.mapping import DenseTSDF
    from taichislam_amd.mapping.submap_mapping import SubmapMapping
    z, ga, gb = _session(DenseTSDF, SubmapMapping)
    for tag, g in (("A", ga), ("B", gb)):
        got = sorted_bits(g.export_submap())
        want = {k[2:]: z[k] for k in z.files if k.startswith(tag + "_")}
        assert np.array_equal(got["indices"], want["indices"]), f"agent {tag}: voxel sets differ"
        assert np.array_equal(got["occupy"], want["occupy"])
        t_g, t_w = got["TSDF"].view(np.float16).astype(np.float64), want["TSDF"].view(np.float16).astype(np.float64)
        fin = np.isfinite(t_g) & np.isfinite(t_w)
        # (submap 0 has an axis-aligned base pose: the literal fusion's first-come 0 / 0 leaves 31 % of agent A's voxels NaN, see aligned_submap_fused)
        # A voxel whose only splats have weights below 2^-24 keeps a finite value in the literal fusion (w * t / w) but is 0 / 0 in the fixed-point sums: 2 of 42 855.
        assert fin.mean() > 0.6 and (np.isnan(t_g) & ~np.isnan(t_w)).sum() <= 1e-4 * t_g.size
        d = np.abs(t_g - t_w)[fin]
        assert np.percentile(d, 50) <= 0.02 * 0.08 and np.percentile(d, 99) <= 0.6 * 0.08, np.percentile(d, [50, 99])


@pytest.mark.skipif(not os.path.exists("/root/reference/taichi_slam/mapping/submap_mapping.py"), reason="needs the reference tree (dev box)")
def test_the_reference_orchestration_itself_on_the_oracle_maps():
    """The third side of the triangle, where the reference tree exists: the REFERENCE's submap_mapping.py (loaded by path, unmodified) driving the oracle-backed
    maps ends with the same two global maps as the reference's orchestration on the reference's maps (the vector) and as the package's orchestration on the
    oracle (the test above)."""
    from test_reference_callers import load_reference_submap_mapping

    class _Base:            # (the reference module imports BaseMap only for annotations)
        pass
    RefSM = load_reference_submap_mapping(OraMap, type("NoOcto", (), {}), _Base)
    z, ga, gb = _session(OraMap, RefSM)
    for tag, g in (("A", ga), ("B", gb)):
        want = {k[2:]: z[k] for k in z.files if k.startswith(tag + "_")}
        assert_bits_equal(sorted_bits(g.export_submap()), want, f"reference orchestration on the oracle, agent {tag}")


def _octo_textured(o, z, depth_fn, export):
    o.set_base_pose_submap(0, np.eye(3), np.zeros(3))
    depth_fn(z["out_tex_R"], z["out_tex_T"], z["out_tex_depth"], z["out_tex_texture"])
    idx, cnt, rgb = export()
    order = np.argsort(lin(idx), kind="stable")
    assert np.array_equal(idx[order].astype(np.int16), z["out_tex_indices"]) and np.array_equal(cnt[order].astype(np.float32).view(np.uint32), z["out_tex_occupy"].view(np.uint32))
    assert np.array_equal(rgb[order].astype(np.float32).view(np.uint32), z["out_tex_color"].view(np.uint32)), "leaf colours (last writer, BGR -> RGB, / 255)"


def test_oracle_textured_octomap_reproduces_the_reference_source():
    from oracle import OracleOctomap
    cfg, K, _, _, _ = load("octomap")
    z = np.load(os.path.join(GOLD, "ref_octomap.npz"))
    o = OracleOctomap(**{**cfg, "texture_enabled": True})
    o.set_intrinsics(K, K)
    _octo_textured(o, z, lambda R, T, d, t: o.integrate_depth(R, T, d, t), lambda: o.export_leaves(with_color=True))


@pytest.mark.gpu
def test_hip_textured_octomap_reproduces_the_reference_source(hip_lib):
    from taichislam_amd.mapping import Octomap
    cfg, K, _, _, _ = load("octomap")
    z = np.load(os.path.join(GOLD, "ref_octomap.npz"))
    o = Octomap(**{**cfg, "texture_enabled": True})
    o.set_dep_camera_intrinsic(K); o.set_color_camera_intrinsic(K)
    _octo_textured(o, z, lambda R, T, d, t: o.recast_depth_to_map(R, T, d, t), lambda: o.export_leaves(with_color=True))


# ------------------------------------------------------------------------------------------------------------------ the literal fusion on the GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(FUSED) + ["blk10_two_submaps_fused"])
def test_hip_sequential_fusion_reproduces_the_reference_source_bit_for_bit(hip_lib, name):
    """semantics = 1 on the submaps AND on the global map: integration and fusion both replay the reference literally (csrc/tsl_sequential.hip) -- the fused
    global maps of the reference's own source, NaNs of the axis-aligned case included, bit for bit.  blk10: blocks of 10 voxels, the reference's own
    configuration (submap_mapping.py:21,:67) -- fuse_submaps' struct-for walks 10^3 blocks while the map is stored in 16^3 bricks (round 4: the replay key
    carries the source cell's place in the reference's block order)."""
    cfg, K, Kc, steps, want = load(name)
    got = replay(lambda over: _Hip({**cfg, **over}, K, Kc, 1), steps, K, Kc, {}, lambda g, m: g.m.fuse_submaps(m.m))
    assert_bits_equal(got, want, f"HIP semantics = 1 (integration + fusion) vs reference source, {name}")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["session", "session_blk10"])
def test_the_reference_session_on_the_hip_maps_with_the_literal_semantics(hip_lib, name):
    """The whole session (eight frames, three submaps, pose-graph update, fusion, a second agent fed from the wire) on HIP maps with semantics = 1:
    both agents' global maps equal the reference's, bit for bit -- with blocks of 16 and with blocks of 10 voxels (the reference's own configuration)."""
    from taichislam_amd.mapping import DenseTSDF
    from taichislam_amd.mapping.submap_mapping import SubmapMapping

    class SeqTSDF(DenseTSDF):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            self.set_option("semantics", 1)
    z, ga, gb = _session(SeqTSDF, SubmapMapping, name)
    for tag, g in (("A", ga), ("B", gb)):
        want = {k[2:]: z[k] for k in z.files if k.startswith(tag + "_")}
        assert_bits_equal(sorted_bits(g.export_submap()), want, f"agent {tag}'s global map, literal semantics on the GPU, {name}")


# ------------------------------------------------------------------------------------------------------------------ provenance of the vectors
@pytest.mark.skipif(not os.path.exists("/root/reference/taichi_slam/mapping/dense_tsdf.py"), reason="needs the reference tree (dev box)")
def test_committed_vectors_regenerate_from_the_reference_source():
    """Where the reference tree exists (the dev box, the driver's CPU tier): run the generator again for two of the vectors -- the reference's dense_tsdf.py /
    taichi_octomap.py imported unmodified and executed on tools/ti_seq, ~12 s -- and compare with the committed files.  The vectors are what that source produces."""
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, os, numpy as np\n"
        f"sys.path.insert(0, {os.path.join(root, 'tools')!r})\n"
        "import gen_ref_golden as G\n"
        "D, O = G.load_reference()\n"
        "name, cfg, K, Kc, steps = [x for x in G.scenarios() if x[0] == 'weight_clamp'][0]\n"
        "res = G.run(D, name, cfg, K, Kc, steps)\n"
        "_, _, _, oc = G.run_octomap(O)\n"
        "np.savez(sys.argv[1], **{'w_' + k: v for k, v in res.items()}, **{'o_' + k: v for k, v in oc.items()})\n")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "regen.npz")
        r = subprocess.run([sys.executable, "-c", code, out], capture_output=True, text=True, timeout=600)          # a process of its own: `import taichi` must not leak
        assert r.returncode == 0, r.stderr[-2000:]
        z = np.load(out)
        w = np.load(os.path.join(GOLD, "ref_weight_clamp.npz"))
        o = np.load(os.path.join(GOLD, "ref_octomap.npz"))
        for k in ("indices", "TSDF", "W_TSDF", "occupy"):
            assert np.array_equal(z["w_" + k], w["out_" + k]), k
        for k in ("indices", "occupy", "tex_indices", "tex_occupy", "tex_color"):
            assert np.array_equal(z["o_" + k], o["out_" + k]), k
