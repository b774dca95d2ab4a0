"""Dev-box tool: golden vectors from the REFERENCE'S OWN SOURCE.

/root/reference/taichi_slam/mapping/{mapping_common,dense_tsdf}.py are loaded by path, unmodified, with `import taichi` resolved to the
sequential stand-in of tools/ti_seq (Taichi is not installable here; the stand-in's header says exactly what it does and does not model),
driven through the reference's public API -- set_base_pose_submap, recast_depth_to_map / recast_pcl_to_map, switch_to_next_submap,
fuse_submaps, export_submap -- on small seeded inputs, and what export_submap returns is committed, with the inputs, as
tests/golden/ref_*.npz.  tests/test_ref_golden.py then checks, where the reference tree does not exist:
  * oracle FAITHFUL (the reference-literal sequential replay) == these maps, bit for bit           (CPU, -m "not gpu")
  * HIP semantics = 1 == these maps, bit for bit; the default HIP path: same voxel set, bounded deviation   (GPU)

    python tools/gen_ref_golden.py            # ~3 minutes; needs /root/reference"""
import importlib.util
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/taichi_slam/mapping"
sys.path.insert(0, os.path.join(ROOT, "tools", "ti_seq")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


Mesher = None


def load_reference():
    """the reference's mapping_common + dense_tsdf by path (its package __init__ pulls in modules that are not part of the path)"""
    import taichi as ti
    assert "ti_seq" in ti.__file__
    pkg = types.ModuleType("taichi_slam"); pkg.__path__ = []
    sub = types.ModuleType("taichi_slam.mapping"); sub.__path__ = [REF]; sub.__package__ = "taichi_slam.mapping"
    sys.modules.update({"taichi_slam": pkg, "taichi_slam.mapping": sub})
    out = {}
    for name in ("mapping_common", "dense_tsdf", "taichi_octomap", "marching_cube_mesher"):
        spec = importlib.util.spec_from_file_location(f"taichi_slam.mapping.{name}", os.path.join(REF, name + ".py"))
        mod = importlib.util.module_from_spec(spec); sys.modules[spec.name] = mod; spec.loader.exec_module(mod); out[name] = mod
    global Mesher
    Mesher = out["marching_cube_mesher"].MarchingCubeMesher
    return out["dense_tsdf"].DenseTSDF, out["taichi_octomap"].Octomap


def lin(idx):
    i = idx.astype(np.int64)
    return ((i[:, 0] + 32768) << 32) | ((i[:, 1] + 32768) << 16) | (i[:, 2] + 32768)


def sorted_export(obj):
    o = np.argsort(lin(np.asarray(obj["indices"])), kind="stable")
    out = {"indices": np.asarray(obj["indices"])[o].astype(np.int16), "TSDF": np.asarray(obj["TSDF"], np.float16)[o].view(np.uint16),
           "W_TSDF": np.asarray(obj["W_TSDF"], np.float16)[o].view(np.uint16), "occupy": np.asarray(obj["occupy"])[o].astype(np.int8)}
    if np.asarray(obj["color"]).size:
        out["color"] = np.asarray(obj["color"], np.float16)[o].view(np.uint16)
    return out


def scenarios():
    from taichislam_amd.utils import synthetic as syn
    rng = np.random.default_rng(20260926)
    eye, zero = np.eye(3), np.zeros(3)
    out = []
    # 1. three frames of the sphere-room stream (the benchmark's scene, scaled down): rays of 30-45 steps, voxels hit by several rays per frame and by all frames
    h, w = 48, 64
    K = syn.scaled_intrinsics(h, w)
    cfg = dict(map_scale=[10.24, 10.24], voxel_scale=0.08, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=6, recast_step=2, max_submap_num=4)
    fr = []
    for f in range(3):
        R, T = syn.camera_pose(4 * f)
        fr.append(dict(kind="depth", R=R, T=T, depth=syn.sphere_room_depth(R, T, h, w, radius=3.0, K=K)))
    out.append(("depth_stream", cfg, K, None, [dict(kind="base", sid=0, R=eye, T=zero)] + fr))
    # 2. point clouds: random directions and ranges, a cluster inside a few sensor voxels, a point beyond max_ray_length; a rotated, shifted pose
    cfg = dict(map_scale=[5.12, 5.12], voxel_scale=0.04, num_voxel_per_blk_axis=16, max_ray_length=2.4, min_ray_length=0.3, internal_voxels=5, recast_step=2, max_submap_num=4)
    fr = [dict(kind="base", sid=0, R=eye, T=zero)]
    for f in range(2):
        d = rng.normal(size=(260, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        pts = d * rng.uniform(0.3, 2.2, size=(260, 1))
        pts = np.concatenate([pts, np.array([[0.5, 0.2, 0.7]]) + rng.uniform(-0.03, 0.03, size=(40, 3)), np.array([[2.0, 2.0, 2.0]])]).astype(np.float32)
        a = 0.3 * (f + 1)
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
        fr.append(dict(kind="pcl", R=R, T=np.array([0.1 * f, -0.05, 0.02]), xyz=pts))
    out.append(("point_clouds", cfg, syn.scaled_intrinsics(30, 40), None, fr))
    # 3. colour: depth + texture through the same projection (color_same_proj), two frames
    h, w = 36, 48
    K = syn.scaled_intrinsics(h, w)
    cfg = dict(map_scale=[5.12, 5.12], voxel_scale=0.08, num_voxel_per_blk_axis=16, max_ray_length=3.0, min_ray_length=0.3, internal_voxels=5, recast_step=1, max_submap_num=4, texture_enabled=True)
    fr = [dict(kind="base", sid=0, R=eye, T=zero)]
    for f in range(2):
        R, T = syn.camera_pose(6 * f, orbit=0.2)
        depth = syn.sphere_room_depth(R, T, h, w, radius=1.6, K=K)
        fr.append(dict(kind="depth", R=R, T=T, depth=depth, texture=rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)))
    out.append(("textured", cfg, K, K, fr))
    # 3b. coloured point clouds (recast_pcl_to_map with rgb_array: the colours are cast to f16 before they are summed, dense_tsdf.py:176-181)
    cfgp = dict(map_scale=[5.12, 5.12], voxel_scale=0.05, num_voxel_per_blk_axis=16, max_ray_length=2.4, min_ray_length=0.3, internal_voxels=5, recast_step=2, max_submap_num=4, texture_enabled=True)
    frp = [dict(kind="base", sid=0, R=eye, T=zero)]
    for f in range(2):
        d = rng.normal(size=(220, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        pts = np.concatenate([d * rng.uniform(0.3, 2.2, size=(220, 1)), np.array([[0.4, -0.3, 0.6]]) + rng.uniform(-0.04, 0.04, size=(60, 3))]).astype(np.float32)
        frp.append(dict(kind="pcl", R=eye, T=np.array([0.02 * f, 0.01, 0.0]), xyz=pts, rgb=rng.integers(0, 256, size=(pts.shape[0], 3), dtype=np.uint8)))
    out.append(("textured_points", cfgp, syn.scaled_intrinsics(30, 40), syn.scaled_intrinsics(30, 40), frp))
    # 4. the weight clamp: a wall 0.35 m in front of a fixed camera, 14 frames -- w = 1 / z^2 = 8 per ray, W reaches Wmax = 1000 inside the run
    h, w = 24, 32
    K = syn.scaled_intrinsics(h, w)
    cfg = dict(map_scale=[2.56, 2.56], voxel_scale=0.04, num_voxel_per_blk_axis=16, max_ray_length=2.0, min_ray_length=0.3, internal_voxels=4, recast_step=1, max_submap_num=4)
    fr = [dict(kind="base", sid=0, R=eye, T=zero)]
    for f in range(14):
        fr.append(dict(kind="depth", R=eye, T=zero, depth=np.full((h, w), 350 + (f % 3), np.uint16)))
    out.append(("weight_clamp", cfg, K, None, fr))
    # 5. two submaps with their own base poses, then fuse_submaps into a global map (dense_tsdf.py:272-318)
    h, w = 30, 40
    K = syn.scaled_intrinsics(h, w)
    cfg = dict(map_scale=[5.12, 5.12], voxel_scale=0.08, num_voxel_per_blk_axis=16, max_ray_length=4.0, min_ray_length=0.3, internal_voxels=5, recast_step=2, max_submap_num=4)
    def tilt(R, a, b):                                   # off the lattice: with an axis-aligned base pose six of the seven splat weights are exactly 0 and the
        ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)      # reference's first-come 0 / 0 turns half of the fused voxels into NaN (kept in: test below)
        return R @ np.array([[ca, -sa, 0], [sa, ca, 0], [0, 0, 1.0]]) @ np.array([[1.0, 0, 0], [0, cb, -sb], [0, sb, cb]])
    R0, T0 = syn.camera_pose(0)
    R1, T1 = syn.camera_pose(20)
    R0, T0 = tilt(R0, 0.17, 0.05), T0 + np.array([0.013, 0.021, -0.037])
    R1, T1 = tilt(R1, -0.11, 0.08), T1 + np.array([-0.027, 0.009, 0.031])
    fr = [dict(kind="base", sid=0, R=R0, T=T0), dict(kind="depth", R=R0, T=T0, depth=syn.sphere_room_depth(R0, T0, h, w, radius=2.5, K=K)),
          dict(kind="next_submap"), dict(kind="base", sid=1, R=R1, T=T1), dict(kind="depth", R=R1, T=T1, depth=syn.sphere_room_depth(R1, T1, h, w, radius=2.5, K=K)),
          dict(kind="next_submap"), dict(kind="fuse", global_map_scale=[10.24, 10.24])]      # (the splat must stay inside the global volume: outside is undefined in the reference)
    out.append(("two_submaps_fused", cfg, K, None, fr))
    # 6. the same with ONE axis-aligned submap: the literal fusion's 0 / 0 (weights exactly 0, whichever splat comes first decides) as it is
    Ra, Ta = syn.camera_pose(0)
    fr = [dict(kind="base", sid=0, R=Ra, T=Ta), dict(kind="depth", R=Ra, T=Ta, depth=syn.sphere_room_depth(Ra, Ta, h, w, radius=2.5, K=K)),
          dict(kind="next_submap"), dict(kind="fuse", global_map_scale=[10.24, 10.24])]
    out.append(("aligned_submap_fused", cfg, K, None, fr))
    # 7. blocks of 10 voxels per axis, the block size of the reference's own configuration (submap_mapping.py:33-36, taichislam_node.py): the volume is
    #    rounded up to whole blocks (N = 70 and 130 here, not a multiple of 16), the sensor grid is made of 10^3 blocks (another ray order) and
    #    fuse_submaps walks the submaps' 10^3 blocks lexicographically (another splat order) -- none of which may change with the 16^3 bricks the maps are stored in
    cfg10 = {**cfg, "num_voxel_per_blk_axis": 10}
    tilted = [s for s in out[-2][4]]
    # 8. exports of a submap with a tilted base pose, a display window (disp_floor / disp_ceiling) and recast_step = 3 on an image whose height it does not
    #    divide: the particle positions go through the submap's pose (dense_tsdf.py:352-355,:379-382), the mesh stays in the map's own frame
    h3, w3 = 61, 80
    K3 = syn.scaled_intrinsics(h3, w3)
    cfg3 = dict(map_scale=[5.12, 5.12], voxel_scale=0.12, num_voxel_per_blk_axis=16, max_ray_length=4.0, min_ray_length=0.3, internal_voxels=3, recast_step=3, max_submap_num=4,
                disp_floor=-0.6, disp_ceiling=0.9)
    fr3 = [dict(kind="base", sid=0, R=R0, T=T0), dict(kind="depth", R=R0, T=T0, depth=syn.sphere_room_depth(R0, T0, h3, w3, radius=2.5, K=K3)),
           dict(kind="depth", R=R0, T=T0 + np.array([0.04, -0.02, 0.03]), depth=syn.sphere_room_depth(R0, T0 + np.array([0.04, -0.02, 0.03]), h3, w3, radius=2.5, K=K3))]
    out.append(("posed_exports", cfg3, K3, None, fr3))
    out.append(("blk10_two_submaps_fused", cfg10, K, None, tilted))
    out.append(("blk10_two_submaps", cfg10, K, None, [s for s in tilted if s["kind"] != "fuse"] + [dict(kind="select", sid=1)]))
    return out


def canon_rows(*cols):
    t = np.concatenate([np.asarray(c, np.float32).reshape(len(c), -1) for c in cols], axis=1).astype(np.float32)
    return t[np.lexsort(t.view(np.uint32).T[::-1])]


WITH_EXPORTS = ("depth_stream", "textured", "posed_exports")


def run(DenseTSDF, name, cfg, K, Kc, steps):
    m = DenseTSDF(**cfg, max_disp_particles=40000 if name in WITH_EXPORTS else 64)
    m.set_dep_camera_intrinsic(K)
    if Kc is not None:
        m.set_color_camera_intrinsic(Kc)
    g, t0 = None, time.time()
    for s in steps:
        if s["kind"] == "base":
            m.set_base_pose_submap(s["sid"], np.ascontiguousarray(s["R"], dtype=np.float64), np.ascontiguousarray(s["T"], dtype=np.float64))
        elif s["kind"] == "depth":
            m.recast_depth_to_map(s["R"], s["T"], s["depth"], s.get("texture", np.zeros((1, 1, 3), np.uint8)))
        elif s["kind"] == "pcl":
            m.recast_pcl_to_map(s["R"], s["T"], s["xyz"], s.get("rgb", np.zeros((1, 3), np.uint8)))
        elif s["kind"] == "next_submap":
            m.switch_to_next_submap()
        elif s["kind"] == "select":                       # export_submap() writes the ACTIVE submap: go back to an earlier one
            m.active_submap_id[None] = s["sid"]
        elif s["kind"] == "fuse":
            g = DenseTSDF(**{**cfg, "is_global_map": True, "map_scale": s["global_map_scale"]}, max_disp_particles=64)
            g.set_dep_camera_intrinsic(K)
            for b in steps:                                   # the global map needs the submaps' base poses (submap_mapping.py sets them the same way)
                if b["kind"] == "base":
                    g.set_base_pose_submap(b["sid"], np.ascontiguousarray(b["R"], dtype=np.float64), np.ascontiguousarray(b["T"], dtype=np.float64))
            g.fuse_submaps(m)
    res = sorted_export((g if g is not None else m).export_submap())
    print(f"{name}: {res['indices'].shape[0]} voxels, {time.time() - t0:.1f} s")
    if name in WITH_EXPORTS:      # marching_cube_mesher.py on the map just built: generate_mesh(1), triangles as rows in a canonical order
        import taichi as ti
        t0 = time.time()
        thres = 5 * cfg["voxel_scale"]
        me = Mesher(m, max_triangles=20000, tsdf_surface_thres=thres)
        me.generate_mesh(1)
        n = int(me.num_facelets[None])
        cols = [me.mesh_vertices.to_numpy()[:n * 3].reshape(n, 9), me.mesh_normals.to_numpy()[:n * 3].reshape(n, 9)]
        if cfg.get("texture_enabled"):
            cols.append(me.mesh_colors.to_numpy()[:n * 3].reshape(n, 9))
        res["mesh"], res["mesh_thres"] = canon_rows(*cols), np.float32(thres)
        print(f"  mesh: {n} triangles, {time.time() - t0:.1f} s")
        # generate_mesh(2): EVERY voxel below the threshold anchors a cube of edge 2 (marching_cube_mesher.py:133-136,:156-159: the corner offsets are
        # multiplied by `step`, the anchors are not thinned out), so the cubes overlap and reach two voxels beyond the anchor's block
        t0 = time.time()
        me2 = Mesher(m, max_triangles=60000, tsdf_surface_thres=thres)
        me2.generate_mesh(2)
        n2 = int(me2.num_facelets[None])
        assert n2 < 60000
        cols2 = [me2.mesh_vertices.to_numpy()[:n2 * 3].reshape(n2, 9), me2.mesh_normals.to_numpy()[:n2 * 3].reshape(n2, 9)]
        if cfg.get("texture_enabled"):
            cols2.append(me2.mesh_colors.to_numpy()[:n2 * 3].reshape(n2, 9))
        res["mesh_step2"] = canon_rows(*cols2)
        print(f"  mesh, step 2: {n2} triangles, {time.time() - t0:.1f} s")
        # the particle exports: cvt_TSDF_surface_to_voxels (dense_tsdf.py:339-366) and cvt_TSDF_to_voxels_slice (:368-391); rows in a canonical order
        m.cvt_TSDF_surface_to_voxels()
        ns = int(m.num_TSDF_particles[None])
        assert ns < m.max_disp_particles
        res["surface"] = canon_rows(m.export_TSDF_xyz.to_numpy()[:ns], m.export_color.to_numpy()[:ns])
        m.cvt_TSDF_to_voxels_slice(0.17, 1.5)
        nz = int(m.num_TSDF_particles[None])
        assert nz < m.max_disp_particles
        res["slice"] = canon_rows(m.export_TSDF_xyz.to_numpy()[:nz], m.export_TSDF.to_numpy()[:nz], m.export_color.to_numpy()[:nz])
        res["slice_args"] = np.array([0.17, 1.5], np.float64)
        # BaseMap.raycast (mapping_common.py:159-173), the planner's query: rays from the last camera position into the scene
        last = [s for s in steps if s["kind"] == "depth"][-1]
        rng = np.random.default_rng(3)
        d = rng.normal(size=(24, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        pos = np.tile(np.asarray(m.input_T[None].to_list(), dtype=np.float64).astype(np.float32), (24, 1)) + rng.uniform(-0.05, 0.05, size=(24, 3)).astype(np.float32)
        hit, end, ln = [], [], []
        for a, b in zip(pos, d.astype(np.float32)):
            succ, x_, _len = m.raycast(ti.Vector([float(a[0]), float(a[1]), float(a[2])], ti.f32), ti.Vector([float(b[0]), float(b[1]), float(b[2])], ti.f32), 3.0)
            hit.append(bool(succ)); end.append([float(e.v) for e in x_]); ln.append(float(_len.v))
        res["ray_pos"], res["ray_dir"], res["ray_max"] = pos.astype(np.float32), d.astype(np.float32), np.float32(3.0)
        res["ray_hit"], res["ray_end"], res["ray_len"] = np.array(hit), np.array(end, np.float32), np.array(ln, np.float32)
        # the planner's point queries (mapping_common.py:178-201 with dense_tsdf.py:148-155): is_pos_occupy, is_pos_unobserved, is_near_pos_occupy(xyz, 2)
        # for points scattered through the observed part of the map and around it
        obs_idx = res["indices"][rng.integers(0, res["indices"].shape[0], size=160)].astype(np.float32) * np.float32(cfg["voxel_scale"])
        qp = np.concatenate([obs_idx + rng.uniform(-0.3, 0.3, size=obs_idx.shape).astype(np.float32), rng.uniform(-1.0, 1.0, size=(40, 3)).astype(np.float32)]).astype(np.float32)
        qo, qu, qn = [], [], []
        for a in qp:
            v = ti.Vector([float(a[0]), float(a[1]), float(a[2])], ti.f32)
            qo.append(bool(m.is_pos_occupy(v))); qu.append(bool(m.is_pos_unobserved(v))); qn.append(bool(m.is_near_pos_occupy(v, 2)))
        res["q_pos"], res["q_occ"], res["q_unobs"], res["q_near2"] = qp, np.array(qo), np.array(qu), np.array(qn)
        print(f"  surface {ns} particles, slice {nz}, rays hit {sum(hit)} of {len(hit)}; of {len(qp)} query points {sum(qo)} occupied, {sum(qu)} unobserved, {sum(qn)} near an occupied voxel")
    return res


def run_octomap(Octomap):
    """taichi_octomap.py: two depth frames and a point cloud into the occupancy tree; the leaves (index, count) are read from the stand-in's
    storage (the reference's Octomap exports nothing but display particles).  Its fuse_submaps is not part of the vector: untextured it
    stops with AttributeError ('Octomap' object has no attribute 'color', taichi_octomap.py:198), textured its result is a last-writer race."""
    from taichislam_amd.utils import synthetic as syn
    rng = np.random.default_rng(7)
    h, w = 30, 40
    K = syn.scaled_intrinsics(h, w)
    cfg = dict(map_scale=[6.4, 6.4], voxel_scale=0.05, min_occupy_thres=1, max_ray_length=3.0, min_ray_length=0.3, K=2, max_submap_num=4, recast_step=1)
    m = Octomap(**cfg, max_disp_particles=64)
    m.set_dep_camera_intrinsic(K)
    m.set_base_pose_submap(0, np.eye(3), np.zeros(3))
    steps, t0 = [dict(kind="base", sid=0, R=np.eye(3), T=np.zeros(3))], time.time()
    for f in range(2):
        R, T = syn.camera_pose(5 * f, orbit=0.2)
        depth = syn.sphere_room_depth(R, T, h, w, radius=1.6, K=K)
        m.recast_depth_to_map(R, T, depth, np.zeros((1, 1, 3), np.uint8))
        steps.append(dict(kind="depth", R=R, T=T, depth=depth))
    d = rng.normal(size=(300, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = np.concatenate([d * rng.uniform(0.3, 2.5, size=(300, 1)), np.array([[0.4, -0.3, 0.6]]) + rng.uniform(-0.02, 0.02, size=(50, 3))]).astype(np.float32)
    Rp = np.array([[0.0, -1.0, 0], [1.0, 0, 0], [0, 0, 1.0]]); Tp = np.array([0.05, 0.1, -0.02])
    m.recast_pcl_to_map(Rp, Tp, pts, np.zeros((1, 3), np.uint8), pts.shape[0])
    steps.append(dict(kind="pcl", R=Rp, T=Tp, xyz=pts))
    def leaves(o, colour=False):
        cells = []
        for b, blk in o.B.blocks.items():
            a = blk[id(o.occupy)]
            for c in np.argwhere(a != 0):
                idx = [bb * s_ + off + cc for bb, s_, off, cc in zip(b, o.B.blk, o.B.offset, c)]
                assert idx[0] == 0
                cells.append((idx[1], idx[2], idx[3], float(a[tuple(c)])) + (tuple(float(x) for x in blk[id(o.color)][tuple(c)]) if colour else ()))
        cells.sort()
        return cells
    cells = leaves(m)
    res = {"indices": np.array([c[:3] for c in cells], np.int16), "occupy": np.array([c[3] for c in cells], np.float32)}
    # the textured tree: colours are stored per point, last writer, channels swapped (taichi_octomap.py:118-121); one frame, so that the
    # last writer of a leaf is the last pixel in raster order whichever way the tree is walked
    mt = Octomap(**{**cfg, "texture_enabled": True}, max_disp_particles=64)
    mt.set_dep_camera_intrinsic(K); mt.set_color_camera_intrinsic(K)
    mt.set_base_pose_submap(0, np.eye(3), np.zeros(3))
    R, T = syn.camera_pose(2, orbit=0.2)
    depth = syn.sphere_room_depth(R, T, h, w, radius=1.6, K=K)
    tex = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    mt.recast_depth_to_map(R, T, depth, tex)
    ct = leaves(mt, True)
    res.update({"tex_R": R, "tex_T": T, "tex_depth": depth, "tex_texture": tex, "tex_indices": np.array([c[:3] for c in ct], np.int16),
                "tex_occupy": np.array([c[3] for c in ct], np.float32), "tex_color": np.array([c[4:] for c in ct], np.float32)})
    # cvt_occupy_to_voxels(level) (taichi_octomap.py:90-102; the node calls it with level 0): the leaves above min_occupy_thres, positions through the
    # submap's base pose (a tilted one here) -- as rows in a canonical order, untextured and textured
    ca, sa = np.cos(0.23), np.sin(0.23)
    Rb, Tb = np.array([[ca, -sa, 0], [sa, ca, 0], [0, 0, 1.0]]) @ np.array([[1.0, 0, 0], [0, np.cos(0.1), -np.sin(0.1)], [0, np.sin(0.1), np.cos(0.1)]]), np.array([0.21, -0.13, 0.07])
    for tag, textured in (("posed", False), ("posedtex", True)):
        mp = Octomap(**{**cfg, "texture_enabled": textured}, max_disp_particles=40000)
        mp.set_dep_camera_intrinsic(K); mp.set_color_camera_intrinsic(K)
        mp.set_base_pose_submap(0, Rb, Tb)
        for s_ in steps[1:3]:
            mp.recast_depth_to_map(s_["R"], s_["T"], s_["depth"], tex if textured else np.zeros((1, 1, 3), np.uint8))
        for level in (0, 1):
            mp.cvt_occupy_to_voxels(level)
            n_ = int(mp.num_export_particles[None])
            assert 100 < n_ < 40000
            cols = [mp.export_x.to_numpy()[:n_]] + ([mp.export_color.to_numpy()[:n_]] if textured else [])
            res[f"{tag}_export{level}"] = canon_rows(*cols)
        print(f"  {tag}: {n_} particles above the threshold")
    res["posed_R"], res["posed_T"] = Rb, Tb
    print(f"octomap: {len(cells)} leaves, {time.time() - t0:.1f} s")
    return cfg, K, steps, res


SESSION = dict(H=48, W=64, OPTS=dict(map_scale=[10.24, 10.24], voxel_scale=0.08, num_voxel_per_blk_axis=16, max_ray_length=5.0, max_submap_num=16, max_disp_particles=64))


def run_session(DenseTSDF, Octomap, name="session", opts=None):
    """The reference's ORCHESTRATION on the reference's maps: taichi_slam/mapping/submap_mapping.py (loaded by path, unmodified; only its hard-coded
    autosave path is turned off) drives eight depth frames through tests/submap_trace.drive -- a new submap every three keyframes, finished submaps
    exported and put on the wire, a pose-graph update, local_to_global -- and a second agent receives the wire buffers (input_remote_submap:
    zlib + np.load, load_numpy into a slot from the top, fuse).  Committed: both agents' global maps."""
    import io
    import zlib
    import submap_trace as st
    from test_reference_callers import load_reference_submap_mapping
    import taichi_slam.mapping.mapping_common as mc
    st.H, st.W, st.OPTS = SESSION["H"], SESSION["W"], {**SESSION["OPTS"], **(opts or {})}
    RefSM = load_reference_submap_mapping(DenseTSDF, Octomap, mc.BaseMap)
    t0 = time.time()
    sm, sent = st.drive(RefSM, DenseTSDF)
    a = sorted_export(sm.global_map.export_submap())
    print(f"session: agent A global map {a['indices'].shape[0]} voxels, {len(sent)} submaps sent, {time.time() - t0:.1f} s")
    smb = RefSM(DenseTSDF, keyframe_step=st.KEYFRAME_STEP, sub_opts=dict(st.OPTS), global_opts=dict(st.OPTS))
    smb.map_send_handle = lambda b: None
    smb.traj_send_handle = lambda b: None
    smb.saveMap = lambda filename: None
    for buf in sent:
        smb.input_remote_submap(buf)
    b = sorted_export(smb.global_map.export_submap())
    print(f"session: agent B (received {len(sent)} submaps) global map {b['indices'].shape[0]} voxels, {time.time() - t0:.1f} s")
    res = {"A_" + k: v for k, v in a.items()}
    res.update({"B_" + k: v for k, v in b.items()})
    res["params"] = np.array(json.dumps({"H": st.H, "W": st.W, "OPTS": st.OPTS, "NFRAMES": st.NFRAMES, "KEYFRAME_STEP": st.KEYFRAME_STEP}))
    path = os.path.join(ROOT, "tests", "golden", f"ref_{name}.npz")
    np.savez_compressed(path, **res)
    print(f"  -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def save(name, cfg, K, Kc, steps, res):
    arrays = {"cfg": np.array(json.dumps(cfg)), "K": np.asarray(K), "Kc": np.asarray(K if Kc is None else Kc), "has_Kc": np.array(Kc is not None),
              "steps": np.array(json.dumps([{k: (None if isinstance(v, np.ndarray) else v) for k, v in s.items()} for s in steps]))}
    for n_, s in enumerate(steps):
        for k, v in s.items():
            if isinstance(v, np.ndarray):
                arrays[f"s{n_}_{k}"] = v
    for k, v in res.items():
        arrays["out_" + k] = v
    path = os.path.join(ROOT, "tests", "golden", f"ref_{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"  -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    assert os.path.exists(REF), "the reference tree is needed to generate the vectors"
    DenseTSDF, Octomap = load_reference()
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    only = sys.argv[1:]
    for name, cfg, K, Kc, steps in scenarios():
        if only and name not in only:
            continue
        save(name, cfg, K, Kc, steps, run(DenseTSDF, name, cfg, K, Kc, steps))
    if not only or "octomap" in only:
        cfg, K, steps, res = run_octomap(Octomap)
        save("octomap", cfg, K, None, steps, res)
    if not only or "session" in only:
        run_session(DenseTSDF, Octomap)
    if not only or "session_blk10" in only:     # the same session with the block size of the reference's own configuration (submap_mapping.py:33-36)
        run_session(DenseTSDF, Octomap, "session_blk10", dict(num_voxel_per_blk_axis=10))
