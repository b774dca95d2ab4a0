"""GPU parity for submap fusion (dense_tsdf.py:272-318) and marching cubes (marching_cube_mesher.py:44-187)."""
import numpy as np
import pytest

from taichislam_amd.utils import synthetic as syn
from util import SMALL, assert_export_equal, make_pair, small_stream, sort_export, sorted_rows

pytestmark = pytest.mark.gpu


def _tri_keys(v, n):
    """Canonical, order-independent representation of a triangle soup: rows of 18 floats sorted lexicographically."""
    t = np.concatenate([v.reshape(-1, 9), n.reshape(-1, 9)], axis=1)
    return sorted_rows(t)


def test_marching_cubes_on_integrated_map(hip_lib):
    from oracle import BATCHED
    from taichislam_amd.mapping import MarchingCubeMesher
    K, frames = small_stream(3)
    g, o = make_pair(SMALL, K)
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None)
        o.integrate_depth(R, T, d, mode=BATCHED)
    thres = 5 * SMALL["voxel_scale"]                      # scripts/taichislam_node.py:209
    mesher = MarchingCubeMesher(g, 400000, tsdf_surface_thres=thres)
    mesher.generate_mesh(1)
    ov, on, _, ontri = o.generate_mesh(1, thres, 400000)
    assert mesher.num_facelets[None] == ontri > 1000 and mesher.vertice_num() == 3 * ontri
    gv, gn, _ = mesher.get_mesh()
    a, b = _tri_keys(gv, gn), _tri_keys(ov, on)
    assert np.array_equal(a[:, :9], b[:, :9]), "mesh vertices differ"
    assert np.array_equal(np.isnan(a[:, 9:]), np.isnan(b[:, 9:]))
    assert np.array_equal(np.nan_to_num(a[:, 9:]), np.nan_to_num(b[:, 9:])), "mesh normals differ"
    assert mesher.num_vertices[None] == 3 * ontri


def test_marching_cubes_sphere_config1(hip_lib):
    """BASELINE configs[0]: analytic sphere SDF in a 128^3 map (tests/marching_cube_test.py scenario); the mesh must
    be a closed surface on the sphere."""
    from taichislam_amd.mapping import DenseTSDF, MarchingCubeMesher
    from oracle import OracleTSDF
    cfg = dict(map_scale=[6.4, 6.4], voxel_scale=0.05, num_voxel_per_blk_axis=16, max_ray_length=10)
    g = DenseTSDF(**cfg, min_occupy_thres=1, enable_esdf=False)          # legacy kwargs of the reference test are accepted
    radius = 1.5
    g.init_sphere(voxels=80, radius=radius)
    e = g.export_submap()
    o = OracleTSDF(**cfg)
    o.import_sparse(0, e["indices"], e["TSDF"], e["W_TSDF"], e["occupy"])
    mesher = MarchingCubeMesher(g, 300000)
    mesher.generate_mesh(1)
    ov, on, _, ontri = o.generate_mesh(1, 0.1, 300000)
    gv, gn, _ = mesher.get_mesh()
    assert mesher.num_facelets[None] == ontri > 5000
    assert np.array_equal(_tri_keys(gv, gn), _tri_keys(ov, on))
    r = np.linalg.norm(gv, axis=1)
    assert np.abs(r - radius).max() < 0.05 * 0.5            # vertices lie on the sphere to O(voxel^2)
    # closed 2-manifold: every edge is shared by exactly two triangles after welding.  The two cells that share an edge walk it in opposite directions,
    # and vertexInterp's `valp2 - valp1` is an f16 operation (relative error up to 2^-11): mu + mu' = d / fl16(d) != 1, so their two copies of the vertex
    # are up to 5e-4 of a voxel = 2.4e-5 m apart -- the reference's own meshes have these hairline offsets.  Weld by distance, not by rounding.
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from scipy.spatial import cKDTree
    pairs = cKDTree(gv).query_pairs(1.2e-3 * cfg["voxel_scale"], output_type="ndarray")
    _, inv = connected_components(coo_matrix((np.ones(len(pairs)), (pairs[:, 0], pairs[:, 1])), shape=(gv.shape[0],) * 2), directed=False)
    uniq = np.unique(inv)
    tri = inv.reshape(-1, 3)
    tri = tri[(tri[:, 0] != tri[:, 1]) & (tri[:, 1] != tri[:, 2]) & (tri[:, 0] != tri[:, 2])]
    edges = np.sort(np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]), axis=1)
    _, counts = np.unique(edges, axis=0, return_counts=True)
    assert (counts == 2).all()
    assert uniq.shape[0] - edges.shape[0] // 2 + tri.shape[0] == 2        # Euler characteristic of a sphere
    nrm = gn[np.isfinite(gn).all(1)]
    assert np.abs(np.linalg.norm(nrm, axis=1) - 1).max() < 2e-3


def _mesh_pair(g, o, thres=0.1, cap=400000):
    from taichislam_amd.mapping import MarchingCubeMesher
    mesher = MarchingCubeMesher(g, cap, tsdf_surface_thres=thres)
    mesher.generate_mesh(1)
    ov, on, _, ontri = o.generate_mesh(1, thres, cap)
    gv, gn, _ = mesher.get_mesh()
    assert mesher.num_facelets[None] == ontri
    a, b = _tri_keys(gv, gn), _tri_keys(ov, on)
    assert np.array_equal(a[:, :9], b[:, :9]), "mesh vertices differ"
    assert np.array_equal(np.nan_to_num(a[:, 9:]), np.nan_to_num(b[:, 9:])), "mesh normals differ"
    return ontri


@pytest.mark.parametrize("plane", ["face", "edge", "corner", "inside"])
def test_marching_cubes_surface_on_brick_boundaries(hip_lib, plane):
    """The brick sign summary (k_mc_summary) lets the tile kernel skip a brick when it and the seven bricks its cells' corners reach into hold
    values on one side of zero only.  Here the surface lies exactly BETWEEN two storage bricks -- across a face, along an edge, at a corner -- so
    every brick is one-sided by itself and the triangles come from cells whose corners are in the neighbours; then the map changes (the flags
    of the first mesh are stale) and is meshed again.  marching_cube_mesher.py:127-187."""
    from taichislam_amd.mapping import DenseTSDF
    from oracle import OracleTSDF
    cfg = dict(map_scale=[3.2, 3.2], voxel_scale=0.05, num_voxel_per_blk_axis=16)
    g, o = DenseTSDF(**cfg), OracleTSDF(**cfg)
    r = np.arange(-24, 24, dtype=np.int16)                                  # three bricks per axis around the origin (brick faces at multiples of 16 from -32)
    ii, jj, kk = np.meshgrid(r, r, r, indexing="ij")
    idx = np.stack([ii, jj, kk], -1).reshape(-1, 3)
    x, y, z = (idx[:, a].astype(np.float32) for a in range(3))
    c = -0.5 if plane != "inside" else 3.3                                   # the zero crossing between voxels -1 | 0 = between two bricks
    d = {"face": x - c, "edge": np.maximum(x - c, y - c), "corner": np.maximum(np.maximum(x - c, y - c), z - c), "inside": x - c}[plane]
    t = (d * np.float32(0.05)).astype(np.float16)

    def load(tt):
        w, occ = np.ones(len(tt), np.float16), np.zeros(len(tt), np.int8)
        g.reset(); g.load_numpy(0, idx, tt, w, occ, None)
        o.reset(); o.import_sparse(0, idx, tt, w, occ)
    load(t)
    n1 = _mesh_pair(g, o, thres=10.0)
    assert n1 > 500
    # the surface moves by half a brick (other bricks become two-sided, the first ones one-sided): nothing of the first mesh's summary may survive
    load(((d - 8.0) * np.float32(0.05)).astype(np.float16))
    n2 = _mesh_pair(g, o, thres=10.0)
    assert n2 > 500


def test_mesh_capacity_clamp(hip_lib):
    from taichislam_amd.mapping import DenseTSDF, MarchingCubeMesher
    g = DenseTSDF(map_scale=[6.4, 6.4], voxel_scale=0.05, num_voxel_per_blk_axis=16)
    g.init_sphere(voxels=40, radius=0.6)
    big = MarchingCubeMesher(g, 100000); big.generate_mesh(1)
    small = MarchingCubeMesher(g, 100); small.generate_mesh(1)
    assert small.num_facelets[None] == big.num_facelets[None] > 100       # true count is reported (Q10)
    assert small.get_mesh()[0].shape[0] == 300


def _two_submap_collection(cfg, K, frames):
    from oracle import BATCHED
    g, o = make_pair(cfg, K)
    R0, T0, _ = frames[0]
    g.set_base_pose_submap(0, R0, T0); o.set_base_pose_submap(0, R0, T0)
    for R, T, d in frames[:2]:
        g.recast_depth_to_map(R, T, d, None); o.integrate_depth(R, T, d, mode=BATCHED)
    g.switch_to_next_submap(); o.set_active_submap(1)
    R2, T2, _ = frames[2]
    g.set_base_pose_submap(1, R2, T2); o.set_base_pose_submap(1, R2, T2)
    for R, T, d in frames[2:4]:
        g.recast_depth_to_map(R, T, d, None); o.integrate_depth(R, T, d, mode=BATCHED)
    # the reference fuses submaps [0, active) poses but all cells; close submap 1 as create_new_submap would
    g.switch_to_next_submap(); o.set_active_submap(2)
    return g, o


def test_fuse_submaps_bit_exact(hip_lib):
    from oracle import BATCHED, OracleTSDF
    from taichislam_amd.mapping import DenseTSDF
    K, frames = small_stream(4)
    cfg = dict(SMALL, max_submap_num=8)
    gs, os_ = _two_submap_collection(cfg, K, frames)
    gcfg = dict(cfg, is_global_map=True)
    gg = DenseTSDF(**gcfg)
    og = OracleTSDF(**gcfg)
    for sid, f in ((0, 0), (1, 2)):
        R, T, _ = frames[f]
        gg.set_base_pose_submap(sid, R, T); og.set_base_pose_submap(sid, R, T)
    gg.fuse_submaps(gs)
    og.fuse_submaps(os_, mode=BATCHED)
    eg, eo = sort_export(gg.export_submap()), sort_export(og.export_sparse())
    assert eg["indices"].shape[0] > 50000 and np.array_equal(eg["indices"], eo["indices"])
    tg, to = eg["TSDF"].view(np.float16), eo["TSDF"].view(np.float16)
    assert np.array_equal(np.isnan(tg), np.isnan(to))
    ok = ~np.isnan(tg)
    assert np.array_equal(eg["TSDF"][ok], eo["TSDF"][ok]) and np.array_equal(eg["W_TSDF"], eo["W_TSDF"])
    assert np.array_equal(eg["occupy"], eo["occupy"])
    # fusing again gives the same map (the global map is rebuilt from scratch, dense_tsdf.py:313)
    gg.fuse_submaps(gs)
    eg2 = sort_export(gg.export_submap())
    assert np.array_equal(eg2["indices"], eg["indices"]) and np.array_equal(eg2["W_TSDF"], eg["W_TSDF"])


def test_submap_mapping_orchestration(hip_lib):
    """SubmapMapping (submap_mapping.py:126-181): keyframe-stepped submaps, local_to_global fusion, wire format."""
    from taichislam_amd.mapping import DenseTSDF, SubmapMapping
    K, frames = small_stream(7)
    sent = []
    sm = SubmapMapping(DenseTSDF, keyframe_step=3,
                       sub_opts=dict(map_scale=[10.24, 10.24], voxel_scale=0.04, num_voxel_per_blk_axis=16, max_ray_length=5.0, max_submap_num=16),
                       global_opts=dict(map_scale=[10.24, 10.24], voxel_scale=0.04, num_voxel_per_blk_axis=16, max_ray_length=5.0, max_submap_num=16))
    sm.map_send_handle = sent.append
    sm.set_dep_camera_intrinsic(K); sm.set_color_camera_intrinsic(K)
    ext = (np.eye(3), np.zeros(3))
    for f, (R, T, d) in enumerate(frames):
        sm.recast_depth_to_map_by_frame(f, True, (R, T), ext, d, np.array([], dtype=int))
    assert sm.submap_collection.get_active_submap_id() == 2 and len(sm.submaps) == 3 and len(sent) == 2
    assert sm.global_map.count_active() > 10000
    sm.cvt_TSDF_surface_to_voxels()
    n = sm.num_TSDF_particles[None]
    assert n > 100 and sm.export_TSDF_xyz.to_numpy()[:n].shape == (n, 3)
    # a second agent ingests the first agent's submap from the wire buffer
    other = SubmapMapping(DenseTSDF, keyframe_step=3,
                          sub_opts=dict(map_scale=[10.24, 10.24], voxel_scale=0.04, num_voxel_per_blk_axis=16, max_ray_length=5.0, max_submap_num=16),
                          global_opts=dict(map_scale=[10.24, 10.24], voxel_scale=0.04, num_voxel_per_blk_axis=16, max_ray_length=5.0, max_submap_num=16))
    other.input_remote_submap(sent[0])
    assert other.submap_collection.remote_submap_num[None] == 1 and other.global_map.count_active() > 1000




def test_submap_mapping_global_map_matches_the_oracle(hip_lib):
    """The package's SubmapMapping (same call trace as the reference's class, tests/test_reference_callers.py) on the HIP shims: three
    keyframe-stepped submaps; the global map it maintains equals the oracle's fusion of the same submaps bit for bit, and a second
    agent that ingests the submaps from the wire rebuilds the same closed submaps."""
    from oracle import BATCHED, OracleTSDF
    from taichislam_amd.mapping import DenseTSDF, SubmapMapping
    opts = dict(map_scale=[10.24, 10.24], voxel_scale=0.04, num_voxel_per_blk_axis=16, max_ray_length=5.0, max_submap_num=16)
    K, frames = small_stream(8)
    sm = SubmapMapping(DenseTSDF, keyframe_step=3, sub_opts=opts, global_opts=opts)
    sent = []
    sm.map_send_handle = sent.append
    sm.set_dep_camera_intrinsic(K)
    ext = (np.eye(3), np.zeros(3))
    oc = OracleTSDF(**opts); oc.set_intrinsics(K)
    for f, (R, T, d) in enumerate(frames):
        if f and f % 3 == 0:
            oc.set_active_submap(oc.get_active_submap() + 1)
        if f % 3 == 0:
            oc.set_base_pose_submap(oc.get_active_submap(), R, T)
        sm.recast_depth_to_map_by_frame(f, True, (R, T), ext, d, np.array([], dtype=int))
        oc.integrate_depth(R, T, d, mode=BATCHED)
    assert len(sent) == 2 and sorted(sm.submaps.values()) == [0, 1, 2]
    sm.local_to_global()
    og = OracleTSDF(**dict(opts, is_global_map=True))
    for fid, sid in sm.submaps.items():
        og.set_base_pose_submap(sid, frames[fid][0], frames[fid][1])
    og.fuse_submaps(oc, mode=BATCHED)
    a, b = sort_export(sm.global_map.export_submap()), sort_export(og.export_sparse())
    assert np.array_equal(a["indices"], b["indices"]) and a["indices"].shape[0] > 50000
    ok = ~np.isnan(a["TSDF"].view(np.float16))
    assert np.array_equal(a["TSDF"][ok], b["TSDF"][ok]) and np.array_equal(a["W_TSDF"], b["W_TSDF"]) and np.array_equal(a["occupy"], b["occupy"])
    # a second agent ingests the two finished submaps from the wire: its collection holds them as remote submaps with the sender's poses
    other = SubmapMapping(DenseTSDF, keyframe_step=3, sub_opts=opts, global_opts=opts)
    for buf in sent:
        other.input_remote_submap(buf)
    assert other.submap_collection.remote_submap_num[None] == 2 and sorted(other.submaps) == [0, 3]
    first_two = OracleTSDF(**opts); first_two.set_intrinsics(K)
    for f, (R, T, d) in enumerate(frames[:6]):
        if f == 3:
            first_two.set_active_submap(1)
        if f % 3 == 0:
            first_two.set_base_pose_submap(first_two.get_active_submap(), R, T)
        first_two.integrate_depth(R, T, d, mode=BATCHED)
    first_two.set_active_submap(2)
    og2 = OracleTSDF(**dict(opts, is_global_map=True))
    og2.set_base_pose_submap(0, frames[0][0], frames[0][1]); og2.set_base_pose_submap(1, frames[3][0], frames[3][1])
    og2.fuse_submaps(first_two, mode=BATCHED)
    a2, b2 = sort_export(other.global_map.export_submap()), sort_export(og2.export_sparse())
    assert np.array_equal(a2["indices"], b2["indices"])
    ok = ~np.isnan(a2["TSDF"].view(np.float16))
    assert np.array_equal(a2["TSDF"][ok], b2["TSDF"][ok]) and np.array_equal(a2["W_TSDF"], b2["W_TSDF"])
