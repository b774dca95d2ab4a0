"""GPU parity with texture_enabled=True: colour accumulation (dense_tsdf.py:204-211,233-234), per-voxel colour
(:268-269, order-free winner rule of DESIGN.md), sparse export of colours, point-cloud colours (:178-183)."""
import numpy as np
import pytest

from taichislam_amd.utils import synthetic as syn
from util import SMALL, assert_export_equal, make_pair, small_stream

pytestmark = pytest.mark.gpu
TEX = dict(SMALL, texture_enabled=True)


def _texture(h, w, seed=0):
    jj, ii = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    rng = np.random.default_rng(seed)
    t = np.stack([(ii * 255 // max(w - 1, 1)), (jj * 255 // max(h - 1, 1)), rng.integers(0, 256, size=(h, w))], axis=-1)
    return np.ascontiguousarray(t.astype(np.uint8))


@pytest.mark.parametrize("same_proj", [True, False])
def test_depth_with_texture(hip_lib, same_proj):
    from oracle import BATCHED
    K, frames = small_stream(3)
    cfg = dict(TEX, color_same_proj=same_proj)
    g, o = make_pair(cfg, K)
    if not same_proj:
        Kc = K.copy(); Kc[0] *= 1.1; Kc[4] *= 1.1; Kc[2] += 3.0
        g.set_color_camera_intrinsic(Kc); o.set_intrinsics(K, Kc)
    for f, (R, T, d) in enumerate(frames):
        tex = _texture(d.shape[0], d.shape[1], f)
        g.recast_depth_to_map(R, T, d, tex)
        o.integrate_depth(R, T, d, tex, mode=BATCHED)
    eg, eo = g.export_submap(), o.export_sparse()
    assert eg["color"].shape == (eg["TSDF"].shape[0], 3) and np.abs(eg["color"].astype(np.float32)).max() <= 1.0
    assert (eg["color"].astype(np.float32) > 0).mean() > 0.5
    assert_export_equal(eg, eo, f"texture same_proj={same_proj}")
    # surface export carries the stored colours (dense_tsdf.py:360-362)
    g.cvt_TSDF_surface_to_voxels()
    n = g.num_TSDF_particles[None]
    oxyz, orgb, on = o.surface_voxels()
    assert n == on > 0
    a = np.concatenate([g.export_TSDF_xyz.to_numpy()[:n], g.export_color.to_numpy()[:n]], 1)
    b = np.concatenate([oxyz, orgb], 1)
    assert np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])


@pytest.mark.parametrize("unit,ramp", [(64, 0), (64, 2), (0, 0)])
def test_textured_batches_through_units_and_parts(hip_lib, unit, ramp):
    """Ten textured frames with nothing read in between: one full batch of eight + a rest (ramp 0) or half batches first.  With the
    unit limit forced down, every brick with more than 64 segments is split into parts: the colour winners then travel through the
    per-(frame, brick) slots of the merge slab and are applied in frame order by the last arriver; unit 0 sends everything there."""
    from oracle import BATCHED
    K, frames = small_stream(10)
    g, o = make_pair(TEX, K)
    g.set_option("unit", unit); g.set_option("ramp", ramp)
    for f, (R, T, d) in enumerate(frames):
        tex = _texture(d.shape[0], d.shape[1], f)
        g.recast_depth_to_map(R, T, d, tex)
        o.integrate_depth(R, T, d, tex, mode=BATCHED)
    assert_export_equal(g.export_submap(), o.export_sparse(), f"textured batches, unit {unit}, ramp {ramp}")


@pytest.mark.parametrize("group", [0, 1])
def test_points_with_colour(hip_lib, group):
    from oracle import BATCHED
    rng = np.random.default_rng(11)
    g, o = make_pair(TEX, syn.K_DEPTH)
    g.set_option("group", group)
    R, T = syn.camera_pose(2)
    d = rng.normal(size=(15000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = (d * rng.uniform(0.5, 4.0, size=(15000, 1))).astype(np.float32)
    rgb = rng.integers(0, 256, size=(15000, 3)).astype(np.uint8)
    g.recast_pcl_to_map(R, T, pts, rgb)
    o.integrate_points(R, T, pts, rgb, mode=BATCHED)
    assert_export_equal(g.export_submap(), o.export_sparse(), "points with colour")


def test_texture_round_trip_and_untextured_input(hip_lib):
    from taichislam_amd.mapping import DenseTSDF
    K, frames = small_stream(1)
    g, _ = make_pair(TEX, K)
    R, T, d = frames[0]
    g.recast_depth_to_map(R, T, d, _texture(*d.shape))
    e = g.export_submap()
    g2 = DenseTSDF(**TEX)
    g2.load_numpy(0, e["indices"], e["TSDF"], e["W_TSDF"], e["occupy"], e["color"])
    assert_export_equal(g2.export_submap(), e, "colour round trip")
    # a textured map fed without a texture integrates geometry only (colours stay as they were)
    g.recast_depth_to_map(R, T, d, np.array([], dtype=int))
    assert g.count_active() == e["TSDF"].shape[0]


def test_textured_mesh_and_fusion(hip_lib):
    """vertexInterp_color (marching_cube_mesher.py:62-82) and colour fusion (dense_tsdf.py:276-277)."""
    from oracle import BATCHED, OracleTSDF
    from taichislam_amd.mapping import DenseTSDF, MarchingCubeMesher
    from util import sort_export, sorted_rows
    K, frames = small_stream(3)
    cfg = dict(TEX, max_submap_num=4)
    g, o = make_pair(cfg, K)
    g.set_base_pose_submap(0, frames[0][0], frames[0][1]); o.set_base_pose_submap(0, frames[0][0], frames[0][1])
    for f, (R, T, d) in enumerate(frames):
        tex = _texture(d.shape[0], d.shape[1], f)
        g.recast_depth_to_map(R, T, d, tex); o.integrate_depth(R, T, d, tex, mode=BATCHED)
    mesher = MarchingCubeMesher(g, 300000, tsdf_surface_thres=0.2)
    mesher.generate_mesh(1)
    ov, on, oc, ontri = o.generate_mesh(1, 0.2, 300000)
    gv, gn, gc = mesher.get_mesh()
    assert mesher.num_facelets[None] == ontri > 1000 and gc is not None
    a = sorted_rows(np.concatenate([gv.reshape(-1, 9), gc.reshape(-1, 9)], 1))
    b = sorted_rows(np.concatenate([ov.reshape(-1, 9), oc.reshape(-1, 9)], 1))
    assert np.array_equal(a, b)
    g.switch_to_next_submap(); o.set_active_submap(1)
    gcfg = dict(cfg, is_global_map=True)
    gg, og = DenseTSDF(**gcfg), OracleTSDF(**gcfg)
    gg.set_base_pose_submap(0, frames[0][0], frames[0][1]); og.set_base_pose_submap(0, frames[0][0], frames[0][1])
    gg.fuse_submaps(g); og.fuse_submaps(o, mode=BATCHED)
    eg, eo = sort_export(gg.export_submap()), sort_export(og.export_sparse())
    assert np.array_equal(eg["indices"], eo["indices"]) and eg["indices"].shape[0] > 10000
    ok = ~np.isnan(eg["TSDF"].view(np.float16))
    assert np.array_equal(eg["TSDF"][ok], eo["TSDF"][ok]) and np.array_equal(eg["W_TSDF"], eo["W_TSDF"])
    cg, co = eg["color"].view(np.float16), eo["color"].view(np.float16)
    okc = ~(np.isnan(cg).any(1) | np.isnan(co).any(1))
    assert np.array_equal(np.isnan(cg), np.isnan(co)) and np.array_equal(eg["color"][okc], eo["color"][okc])
