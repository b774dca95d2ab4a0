"""GPU tests of the boundary's device-pointer and slice entry points (include/taichislam_hip.h: *_dev, tsl_esdf_slice,
tsl_octo_pack_pointcloud2): each against the host-buffer form of the same call and, where one exists, against the oracle.

Reference callers these serve: scripts/taichislam_node.py:330-333 (Octomap export_x -> PointCloud2), :342 (mesh buffers), :350-351
(export_TSDF_xyz / export_color .to_numpy()), taichi_slam/mapping/dense_esdf.py:498-509 (cvt_ESDF_to_voxels_slice),
taichi_slam/mapping/topo_graph.py:444-507 (64-128 rays per node expansion)."""
import numpy as np
import pytest

from taichislam_amd.utils import synthetic as syn
from util import SMALL, lin, make_pair, small_stream, sorted_rows

pytestmark = pytest.mark.gpu


def _integrated_pair(n=3, **kw):
    from oracle import BATCHED
    K, frames = small_stream(n)
    g, o = make_pair(dict(SMALL, **kw), K)
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None)
        o.integrate_depth(R, T, d, mode=BATCHED)
    return g, o, frames


def test_esdf_slice_matches_the_oracle_layer(hip_lib):
    """cvt_ESDF_to_voxels_slice(z): the observed voxels with _index - 0.5 < k < _index + 0.5 (k from the bottom of the volume), their ESDF
    and submap-frame positions -- rebuilt here from the oracle's Dijkstra ESDF; host copy and device view must agree."""
    g, o, frames = _integrated_pair()
    vs = SMALL["voxel_scale"]
    g.update_esdf(max_dist=1.0)
    oi, oe = o.esdf(max_dist=1.0)
    R0, T0 = np.eye(3), np.zeros(3)                                   # default base pose of submap 0
    for z in (0.0, 0.37, -0.52):
        g.cvt_ESDF_to_voxels_slice(z)
        n = g.num_export_ESDF_particles[None]
        xyz, val = g.export_ESDF_xyz.to_numpy(n), g.export_ESDF.to_numpy(n)
        index_f = np.float32((z + g.Nz * vs / 2.0) / vs)
        ku = (oi[:, 2].astype(np.int32) + g.Nz // 2).astype(np.float32)
        sel = (index_f - np.float32(0.5) < ku) & (ku < index_f + np.float32(0.5))
        want_xyz = (oi[sel].astype(np.float32) * np.float32(vs)) @ R0.T.astype(np.float32) + T0.astype(np.float32)
        assert n == int(sel.sum()) > 500, (z, n, int(sel.sum()))
        a = sorted_rows(np.concatenate([xyz, val[:, None]], 1)); b = sorted_rows(np.concatenate([want_xyz, oe[sel][:, None]], 1))
        assert np.array_equal(a, b), f"slice z={z}"
        tx, tv = g.export_ESDF_xyz.to_torch(), g.export_ESDF.to_torch()
        assert tx.is_cuda and tuple(tx.shape) == (n, 3) and np.array_equal(tx.cpu().numpy(), xyz) and np.array_equal(tv.cpu().numpy(), val)
    # the whole field as device tensors == the host export
    ti, te = g.export_esdf_torch()
    hi, he = g.export_esdf()
    assert ti.is_cuda and np.array_equal(ti.cpu().numpy(), hi) and np.array_equal(te.cpu().numpy(), he)


def test_particle_and_mesh_buffers_as_device_tensors(hip_lib):
    from taichislam_amd.mapping import MarchingCubeMesher
    g, o, _ = _integrated_pair()
    g.cvt_TSDF_surface_to_voxels()
    n = g.num_TSDF_particles[None]
    assert n > 1000
    x = g.export_TSDF_xyz.to_torch(); c = g.export_color.to_torch()
    assert x.is_cuda and tuple(x.shape) == (n, 3) and tuple(c.shape) == (n, 3)
    assert np.array_equal(x.cpu().numpy(), g.export_TSDF_xyz.to_numpy(n)) and np.array_equal(c.cpu().numpy(), g.export_color.to_numpy(n))
    g.cvt_TSDF_to_voxels_slice(0.1)
    n2 = g.num_TSDF_particles[None]
    assert np.array_equal(g.export_TSDF.to_torch().cpu().numpy(), g.export_TSDF.to_numpy(n2)) and n2 > 100
    mesher = MarchingCubeMesher(g, max_triangles=400000, tsdf_surface_thres=5 * SMALL["voxel_scale"])
    mesher.generate_mesh(1)
    nt = mesher.num_facelets[None]
    v, nr = mesher.mesh_vertices.to_torch(), mesher.mesh_normals.to_torch()
    assert nt > 1000 and tuple(v.shape) == (3 * nt, 3) and v.is_cuda
    hv, hn, _ = mesher.get_mesh()
    assert np.array_equal(v.cpu().numpy(), hv) and np.array_equal(nr.cpu().numpy(), hn)
    # a consumer on the GPU: mean vertex without ever leaving the device
    assert abs(float(v.mean()) - float(hv.mean())) < 1e-4


def test_queries_with_device_tensors_equal_the_host_form_and_the_oracle(hip_lib):
    import torch
    g, o, frames = _integrated_pair()
    rng = np.random.default_rng(5)
    pts = rng.uniform(-3.2, 3.2, size=(4096, 3)).astype(np.float32)
    for mode, param in ((0, 0), (1, 0), (2, 2)):
        host = g._query_points(mode, pts, param)
        dev = g._query_points(mode, torch.from_numpy(pts).cuda(), param)
        assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), host)
        assert np.array_equal(host, o.query_points(mode, pts, param).astype(bool)), f"mode {mode}"
    # 128 rays from the sensor position, as a planner expands a node
    pos = np.tile(frames[-1][1].astype(np.float32), (128, 1))
    dirs = rng.normal(size=(128, 3)).astype(np.float32); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    h0, e0, l0 = g.raycast(pos, dirs, 4.0)
    h1, e1, l1 = g.raycast(torch.from_numpy(pos).cuda(), torch.from_numpy(dirs).cuda(), 4.0)
    assert np.array_equal(h1.cpu().numpy(), h0) and np.array_equal(e1.cpu().numpy(), e0) and np.array_equal(l1.cpu().numpy(), l0)
    oh, oe, ol = o.raycast(pos, dirs, 4.0)
    assert np.array_equal(h0, oh.astype(bool)) and np.array_equal(e0, oe) and np.array_equal(l0, ol) and 0 < h0.sum() <= 128
    # frames queued behind the query's input are ordered before it: integrate, then query without any sync in between
    R, T, d = frames[0]
    g.recast_depth_to_map(R, T, torch.from_numpy(d.view(np.int16)).cuda(), None)
    o.integrate_depth(R, T, d)
    h2, _, _ = g.raycast(torch.from_numpy(pos).cuda(), torch.from_numpy(dirs).cuda(), 4.0)
    assert np.array_equal(h2.cpu().numpy(), o.raycast(pos, dirs, 4.0)[0].astype(bool))


def test_octomap_device_points_and_pointcloud2(hip_lib):
    import torch
    from oracle import OracleOctomap
    from taichislam_amd.mapping import Octomap
    from taichislam_amd.utils import ros_adapters
    cfg = dict(map_scale=[12.8, 12.8], voxel_scale=0.05, min_occupy_thres=1, min_ray_length=0.3, max_ray_length=5.0, K=2, max_submap_num=4)
    rng = np.random.default_rng(11)
    pts = (rng.integers(-40, 40, size=(6000, 3)) * 0.05).astype(np.float32)
    pts = np.concatenate([pts, pts[:3000]])                            # repeated hits: counts above the threshold
    R, T = syn.camera_pose(0)
    for tex in (False, True):
        rgb = rng.integers(0, 255, size=(pts.shape[0], 3)).astype(np.uint8) if tex else None
        a, b = Octomap(**cfg, texture_enabled=tex), Octomap(**cfg, texture_enabled=tex)
        o = OracleOctomap(**cfg, texture_enabled=tex)
        a.recast_pcl_to_map(R, T, pts, rgb)
        b.recast_pcl_to_map(R, T, torch.from_numpy(pts).cuda(), torch.from_numpy(rgb).cuda() if tex else None)
        o.integrate_points(R, T, pts, rgb)
        la, lb, lo = a.export_leaves(with_color=True), b.export_leaves(with_color=True), o.export_leaves(with_color=True)
        ra, rb, ro = (sorted_rows(np.concatenate([x[0].astype(np.float64), x[1][:, None], x[2]], 1)) for x in (la, lb, lo))
        assert ra.shape[0] > 3000 and np.array_equal(ra, rb) and np.array_equal(ra, ro), f"textured {tex}"
        b.cvt_occupy_to_voxels(0)
        n = b.num_export_particles[None]
        xyz, col = b.export_x.to_numpy(n), b.export_color.to_numpy(n)
        assert n > 1000 and np.array_equal(b.export_x.to_torch().cpu().numpy(), xyz)
        pay = b.pointcloud2()
        want = ros_adapters.pointcloud2_payload(np.concatenate([xyz, col], 1) if tex else xyz, tex)
        assert pay["width"] == n and pay["point_step"] == (24 if tex else 12) and pay["data"] == want["data"] and pay["fields"] == want["fields"]


def test_textured_merge_then_fuse_has_clean_colour_sums(hip_lib):
    """A multi-GPU style merge of textured submaps (the exchange carries no colour) followed by a plain fuse_submaps on the SAME global
    handle: the fusion's colour must equal that of a fresh handle -- nothing of the merge may linger in the fusion scratch -- and a
    merge abandoned after merge_begin must not leak into the next fusion either."""
    from taichislam_amd.mapping import DenseTSDF
    from util import sort_export
    cfg = dict(SMALL, map_scale=[5.12, 5.12], max_submap_num=4, texture_enabled=True)
    K, frames = small_stream(3)
    rng = np.random.default_rng(2)
    tex = rng.integers(0, 255, size=(120, 160, 3)).astype(np.uint8)
    sub = DenseTSDF(**cfg); sub.set_dep_camera_intrinsic(K); sub.set_color_camera_intrinsic(K)
    for R, T, d in frames:
        sub.recast_depth_to_map(R, T, d, tex)
    sub.active_submap_id[None] = 1
    fresh = DenseTSDF(**dict(cfg, is_global_map=True)); fresh.fuse_submaps(sub)
    want = sort_export(fresh.export_submap())
    g = DenseTSDF(**dict(cfg, is_global_map=True))
    g.allreduce_merge(sub, None)                                       # one-rank merge: TSDF / W as a fusion, no colour
    m = sort_export(g.export_submap())
    assert np.array_equal(m["indices"], want["indices"]) and np.array_equal(m["TSDF"], want["TSDF"]) and np.array_equal(m["W_TSDF"], want["W_TSDF"])
    g.merge_begin(sub)                                                 # ... and one that is abandoned half way
    g.fuse_submaps(sub)
    got = sort_export(g.export_submap())
    for k in ("indices", "TSDF", "W_TSDF", "occupy", "color"):
        assert np.array_equal(got[k], want[k]), k
    assert want["color"].any()
