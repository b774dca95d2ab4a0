"""GPU parity for the Octomap hit counter (taichi_slam/mapping/taichi_octomap.py:116-189)."""
import numpy as np
import pytest

from taichislam_amd.utils import synthetic as syn
from util import small_stream, sorted_rows

pytestmark = pytest.mark.gpu
CFG = dict(map_scale=[12.8, 12.8], voxel_scale=0.05, min_occupy_thres=2, min_ray_length=0.3, max_ray_length=5.0, K=2, max_submap_num=8)


def _pair(**kw):
    from oracle import OracleOctomap
    from taichislam_amd.mapping import Octomap
    cfg = dict(CFG, **kw)
    return Octomap(**cfg), OracleOctomap(**{k: v for k, v in cfg.items() if k != "max_disp_particles"})


def _leaves_equal(g, o):
    gi, gc = g.export_leaves()
    oi, oc = o.export_leaves()
    a = sorted_rows(np.concatenate([gi.astype(np.float64), gc[:, None]], 1))
    b = sorted_rows(np.concatenate([oi.astype(np.float64), oc[:, None]], 1))
    assert a.shape == b.shape and a.shape[0] > 100 and np.array_equal(a, b)


def test_octomap_depth_points_and_levels(hip_lib):
    K, frames = small_stream(5)
    g, o = _pair()
    assert (g.N, g.Nz, g.Rxy, g.Rz) == (o.N, o.Nz, o.Rxy, o.Rz) == (256, 256, 8, 8)
    g.set_dep_camera_intrinsic(K); o.set_intrinsics(K)
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, np.array([], dtype=int))
        so = o.integrate_depth(R, T, d)
        sg = g.last_frame_stats()
        assert (sg["p_used"], sg["p_valid"], sg["p_oob"]) == (so["p_used"], so["p_valid"], so["p_oob"])
    rng = np.random.default_rng(3)
    pts = rng.uniform(-4, 4, size=(5000, 3)).astype(np.float32)
    g.recast_pcl_to_map(frames[0][0], frames[0][1], pts, None, 5000)
    o.integrate_points(frames[0][0], frames[0][1], pts)
    _leaves_equal(g, o)
    for level in (0, 1, 2, 3):
        gx, _ = g.get_occupy_voxels(level)
        ox, on = o.occupied_voxels(level)
        assert gx.shape[0] == on and np.array_equal(sorted_rows(gx), sorted_rows(ox)), f"level {level}"
    assert g.get_occupy_voxels(0)[0].shape[0] > g.get_occupy_voxels(3)[0].shape[0] > 0


def test_octomap_c3_shape(hip_lib):
    """BASELINE configs[2] geometry: 1024^3 / 5 cm, K=2 (tree depth 10); full-size frames, counts match the oracle."""
    from oracle import OracleOctomap
    from taichislam_amd.mapping import Octomap
    cfg = dict(map_scale=[51.2, 51.2], voxel_scale=0.05, K=2, max_ray_length=5.0, max_submap_num=4)
    g, o = Octomap(**cfg), OracleOctomap(**cfg)
    assert (g.N, g.Rxy) == (1024, 10)
    g.set_dep_camera_intrinsic(syn.K_DEPTH); o.set_intrinsics(syn.K_DEPTH)
    for R, T, d in syn.sphere_room_stream(3):
        g.recast_depth_to_map(R, T, d, None); o.integrate_depth(R, T, d)
    _leaves_equal(g, o)


def test_octomap_submaps_and_fusion(hip_lib):
    from oracle import OracleOctomap
    from taichislam_amd.mapping import Octomap
    K, frames = small_stream(4)
    g, o = _pair(min_occupy_thres=0)
    g.set_dep_camera_intrinsic(K); o.set_intrinsics(K)
    for sid, fr in ((0, frames[:2]), (1, frames[2:])):
        g.set_base_pose_submap(sid, fr[0][0], fr[0][1]); o.set_base_pose_submap(sid, fr[0][0], fr[0][1])
        for R, T, d in fr:
            g.recast_depth_to_map(R, T, d, None); o.integrate_depth(R, T, d)
        _leaves_equal(g, o)
        g.switch_to_next_submap(); o.set_active_submap(sid + 1)
    gg, og = Octomap(**dict(CFG, is_global_map=True, min_occupy_thres=0)), OracleOctomap(**dict(CFG, is_global_map=True, min_occupy_thres=0))
    for sid, f in ((0, 0), (1, 2)):
        gg.set_base_pose_submap(sid, frames[f][0], frames[f][1]); og.set_base_pose_submap(sid, frames[f][0], frames[f][1])
    gg.fuse_submaps(g); og.fuse_submaps(o)
    _leaves_equal(gg, og)
    gg.reset()
    assert gg.export_leaves()[0].shape[0] == 0


def _coloured_leaves_equal(g, o, what):
    gi, gc, gr = g.export_leaves(with_color=True)
    oi, oc, orgb = o.export_leaves(with_color=True)
    a = sorted_rows(np.concatenate([gi.astype(np.float64), gc[:, None], gr], 1))
    b = sorted_rows(np.concatenate([oi.astype(np.float64), oc[:, None], orgb], 1))
    assert a.shape == b.shape and a.shape[0] > 100 and np.array_equal(a, b), what
    assert np.any(gr > 0)


@pytest.mark.parametrize("same_proj", [True, False])
def test_octomap_colour(hip_lib, same_proj):
    """Leaf colours (taichi_octomap.py:120-124, BGR -> RGB / 255): the pixel / point with the largest index colours a leaf, later
    frames overwrite, exports carry the colour (:100-101) and the fusion takes the colour of the largest source leaf (:189)."""
    K, frames = small_stream(3)
    Kc = K.copy(); Kc[0] *= 0.9; Kc[4] *= 0.9; Kc[2] += 3.0
    g, o = _pair(texture_enabled=True, color_same_proj=same_proj, max_disp_particles=200000)
    g.set_dep_camera_intrinsic(K); g.set_color_camera_intrinsic(Kc); o.set_intrinsics(K, Kc)
    rng = np.random.default_rng(5)
    for R, T, d in frames:
        tex = rng.integers(0, 256, size=(d.shape[0], d.shape[1], 3)).astype(np.uint8)
        g.recast_depth_to_map(R, T, d, tex)
        o.integrate_depth(R, T, d, tex)
    _coloured_leaves_equal(g, o, "depth + texture")
    pts = rng.uniform(-3, 3, size=(20000, 3)).astype(np.float32)
    pts[:4000] = pts[4000:8000]                                               # several points per leaf: the last one wins
    rgb = rng.integers(0, 256, size=(20000, 3)).astype(np.uint8)
    g.recast_pcl_to_map(frames[0][0], frames[0][1], pts, rgb)
    o.integrate_points(frames[0][0], frames[0][1], pts, rgb)
    _coloured_leaves_equal(g, o, "points + rgb")
    gx, gcol = g.get_occupy_voxels(0)
    ox, ocol, on = o.occupied_voxels(0, with_color=True)
    assert gx.shape[0] == on > 0
    assert np.array_equal(sorted_rows(np.concatenate([gx, gcol], 1)), sorted_rows(np.concatenate([ox, ocol], 1)))


def test_octomap_coloured_fusion(hip_lib):
    from oracle import OracleOctomap
    from taichislam_amd.mapping import Octomap
    K, frames = small_stream(4)
    cfg = dict(CFG, texture_enabled=True, min_occupy_thres=0)
    gs, os_ = Octomap(**cfg), OracleOctomap(**cfg)
    gg, og = Octomap(**dict(cfg, is_global_map=True)), OracleOctomap(**dict(cfg, is_global_map=True))
    gs.set_dep_camera_intrinsic(K); gs.set_color_camera_intrinsic(K); os_.set_intrinsics(K)
    rng = np.random.default_rng(9)
    for s in range(2):
        Rb, Tb = syn.camera_pose(s * 2)
        for m in (gs, gg):
            m.set_base_pose_submap(s, Rb, Tb)
        for m in (os_, og):
            m.set_base_pose_submap(s, Rb, Tb)
        gs.active_submap_id[None] = s; os_.set_active_submap(s)
        for R, T, d in frames[s * 2:s * 2 + 2]:
            tex = rng.integers(0, 256, size=(d.shape[0], d.shape[1], 3)).astype(np.uint8)
            gs.recast_depth_to_map(R, T, d, tex); os_.integrate_depth(R, T, d, tex)
    gs.active_submap_id[None] = 2; os_.set_active_submap(2)
    gg.fuse_submaps(gs); og.fuse_submaps(os_)
    _coloured_leaves_equal(gg, og, "fused colours")


def test_octomap_device_frames_queued_eight_per_launch(hip_lib):
    """Round 6: device-resident depth frames are only queued by recast_depth_to_map and inserted up to eight per launch (taichi_octomap.py:147-169: the insert is a count,
    exact in any order).  21 frames at the C3 geometry handed over back to back -- two and a half batches, frames of two sizes in one batch, a switch of the active
    submap and a point cloud in between -- must leave exactly the oracle's leaves; the last frame's statistics are that frame's."""
    import torch
    from oracle import OracleOctomap
    from taichislam_amd.mapping import Octomap
    cfg = dict(map_scale=[51.2, 51.2], voxel_scale=0.05, K=2, max_ray_length=5.0, max_submap_num=4)
    g, o = Octomap(**cfg), OracleOctomap(**cfg)
    g.set_dep_camera_intrinsic(syn.K_DEPTH); o.set_intrinsics(syn.K_DEPTH)
    frames = list(syn.sphere_room_stream(21))
    dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
    half = [torch.from_numpy(np.ascontiguousarray(d[:240]).view(np.int16)).cuda() for _, _, d in frames]
    so = None
    for f, (R, T, d) in enumerate(frames):
        if f == 13:                              # something else on the handle in the middle of a batch: the queue is issued first
            pts = np.random.default_rng(5).uniform(-4, 4, size=(3000, 3)).astype(np.float32)
            g.recast_pcl_to_map(R, T, pts, None, 3000); o.integrate_points(R, T, pts)
        if f % 5 == 3:                           # a frame of another size inside the batch
            g.recast_depth_to_map(R, T, half[f], None); so = o.integrate_depth(R, T, np.ascontiguousarray(d[:240]))
        else:
            g.recast_depth_to_map(R, T, dev[f], None); so = o.integrate_depth(R, T, d)
    sg = g.last_frame_stats()
    assert (sg["p_used"], sg["p_valid"], sg["p_oob"]) == (so["p_used"], so["p_valid"], so["p_oob"])
    _leaves_equal(g, o)
    # host images go through the same insert, one at a time (their staging buffer is reused by the next call)
    g2 = Octomap(**cfg); g2.set_dep_camera_intrinsic(syn.K_DEPTH)
    for f, (R, T, d) in enumerate(frames):
        if f == 13:
            g2.recast_pcl_to_map(R, T, pts, None, 3000)
        g2.recast_depth_to_map(R, T, np.ascontiguousarray(d[:240]) if f % 5 == 3 else d, None)
    _leaves_equal(g2, o)


def test_octomap_reset_and_submap_switch_with_frames_still_queued(hip_lib):
    """Frames queued but not yet launched when reset() / a switch of the active submap / an export comes: the queue is issued first, in order."""
    import torch
    from oracle import OracleOctomap
    from taichislam_amd.mapping import Octomap
    cfg = dict(CFG)
    g, o = Octomap(**cfg), OracleOctomap(**{k: v for k, v in cfg.items() if k != "max_disp_particles"})
    K, frames = small_stream(7)
    g.set_dep_camera_intrinsic(K); o.set_intrinsics(K)
    dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
    for f in range(3):                                  # three frames in the queue, then everything is thrown away
        g.recast_depth_to_map(frames[f][0], frames[f][1], dev[f], None)
    g.reset()
    for f in range(3, 7):                               # four more (still fewer than a batch), read back at once
        g.recast_depth_to_map(frames[f][0], frames[f][1], dev[f] if f % 2 else frames[f][2], None)
        o.integrate_depth(*frames[f])
    _leaves_equal(g, o)
