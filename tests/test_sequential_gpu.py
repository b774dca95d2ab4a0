"""Option "semantics" = 1: the reference-literal sequential replay of process_new_pcl (taichi_slam/mapping/dense_tsdf.py:236-270) on the
GPU (csrc/tsl_sequential.hip) must equal the oracle's FAITHFUL mode -- raster-order f16 pixel sums, rays in Taichi's struct-for order,
every ray step applied on its own with f16 rounding and the W clamp at 1000 -- BIT FOR BIT: voxel sets, TSDF and W bits, occupancy and
the frame counters.  A sequential schedule is a legal outcome of the racy reference, so this is the strongest parity statement available
without Taichi; the default (batched) path is measured against the same yardstick in test_parity_vs_faithful_gpu.py."""
import numpy as np
import pytest

from taichislam_amd.utils import synthetic as syn
from util import sort_export, C2, SMALL, assert_export_equal, make_pair, small_stream

pytestmark = pytest.mark.gpu
STAT_KEYS = ("p_used", "p_valid", "p_oob", "v_pcl", "v_skipped", "steps", "steps_oob", "unique", "bricks")


def test_sequential_mode_equals_faithful_small(hip_lib):
    from oracle import FAITHFUL
    K, frames = small_stream(5)
    g, o = make_pair(SMALL, K)
    g.set_option("semantics", 1)
    assert g.get_option("semantics") == 1
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None)
        so = o.integrate_depth(R, T, d, mode=FAITHFUL)
        sg = g.last_frame_stats()
        assert {k: sg[k] for k in STAT_KEYS} == {k: so[k] for k in STAT_KEYS}
    assert_export_equal(g.export_submap(), o.export_sparse(), "sequential vs FAITHFUL, small stream")
    # back to the default semantics on the same map: the next frames are the batched sums again, applied to the sequential map
    from oracle import BATCHED
    g.set_option("semantics", 0)
    for R, T, d in frames[:2]:
        g.recast_depth_to_map(R, T, d, None)
        o.integrate_depth(R, T, d, mode=BATCHED)
    assert_export_equal(g.export_submap(), o.export_sparse(), "batched frames on top of the sequential map")


def test_sequential_mode_equals_faithful_at_the_benchmark_size(hip_lib):
    """BASELINE configs[1], 12 frames: HIP(sequential) == oracle FAITHFUL on every TSDF / W bit of 1.4 M voxels."""
    import torch
    from oracle import FAITHFUL
    g, o = make_pair(C2, syn.K_DEPTH)
    g.set_option("semantics", 1)
    so = None
    for R, T, d in syn.sphere_room_stream(12):
        g.recast_depth_to_map(R, T, torch.from_numpy(d.view(np.int16)).cuda(), None)
        so = o.integrate_depth(R, T, d, mode=FAITHFUL)
    sg = g.last_frame_stats()
    assert {k: sg[k] for k in STAT_KEYS} == {k: so[k] for k in STAT_KEYS} and sg["steps"] > 3_000_000
    e = g.export_submap()
    assert e["TSDF"].shape[0] > 1_300_000
    assert_export_equal(e, o.export_sparse(), "sequential vs FAITHFUL, C2, 12 frames")


@pytest.mark.parametrize("inp", ["host", "device"])
def test_sequential_mode_long_pipelined_stream_with_readers_in_between(hip_lib, inp):
    """44 frames queued without a synchronisation per frame -- every batch slot and frame working set is used twice over, batches of eight in
    flight -- as host numpy images (the visited pixels staged into the mapped pinned buffers) or as device tensors; a mesh, an ESDF update and a
    surface export read the map in between (each issues the queued frames first).  Map and frame counters == FAITHFUL.  dense_tsdf.py:236-270."""
    import torch
    from oracle import FAITHFUL
    from taichislam_amd.mapping import MarchingCubeMesher
    K, frames = small_stream(44)
    g, o = make_pair(SMALL, K)
    g.set_option("semantics", 1)
    dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames] if inp == "device" else None
    mesher = MarchingCubeMesher(g, 400000, tsdf_surface_thres=5 * SMALL["voxel_scale"])
    for f, (R, T, d) in enumerate(frames):
        g.recast_depth_to_map(R, T, dev[f] if dev else d, None)
        so = o.integrate_depth(R, T, d, mode=FAITHFUL)
        if f == 13:
            mesher.generate_mesh(1)
            ov, on, _, ontri = o.generate_mesh(1, 5 * SMALL["voxel_scale"], 400000)
            assert mesher.num_facelets[None] == ontri > 1000
        if f == 21:
            g.update_esdf(wait=False)
        if f == 30:
            g.cvt_TSDF_surface_to_voxels()
            assert g.num_TSDF_particles[None] > 0
    sg = g.last_frame_stats()
    assert {k: sg[k] for k in STAT_KEYS} == {k: so[k] for k in STAT_KEYS}
    assert_export_equal(g.export_submap(), o.export_sparse(), f"sequential, 44 pipelined frames, {inp} input")


def test_sequential_mode_points_crowded_voxels_and_out_of_volume_rays(hip_lib):
    """recast_pcl_to_map in sequential mode: random directions and ranges (rays that leave the volume, degenerate points at the origin)
    plus a cluster of 3 000 points inside a few sensor voxels (the crowded-voxel replay of k_segments); two frames, FAITHFUL bit for bit."""
    from oracle import FAITHFUL
    rng = np.random.default_rng(11)
    g, o = make_pair(SMALL, syn.K_DEPTH)
    g.set_option("semantics", 1)
    for f in range(2):
        R, T = syn.camera_pose(2 + f)
        d = rng.normal(size=(15000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        pts = (d * rng.uniform(0.2, 6.0, size=(15000, 1))).astype(np.float32)
        pts[:5] = 0.0
        cluster = (np.array([[0.8, 0.3, 1.1]]) + rng.uniform(-0.05, 0.05, size=(3000, 3))).astype(np.float32)
        pts = np.concatenate([pts, cluster])
        g.recast_pcl_to_map(R, T, pts, np.array([]))
        so = o.integrate_points(R, T, pts, None, mode=FAITHFUL)
        sg = g.last_frame_stats()
        assert {k: sg[k] for k in STAT_KEYS} == {k: so[k] for k in STAT_KEYS}
        assert so["v_skipped"] >= 1 and so["steps_oob"] >= 0
    assert_export_equal(g.export_submap(), o.export_sparse(), "sequential, point input")


def test_sequential_mode_textured_equals_faithful(hip_lib):
    """Colour under the sequential semantics: every ray step stores its ray's colour (dense_tsdf.py:268-269), so a voxel ends a frame with the colour of
    the last ray of its replay run.  Four 120 x 160 frames with random textures, the point-cloud form with colours as well: map and colours == FAITHFUL."""
    from oracle import FAITHFUL
    rng = np.random.default_rng(5)
    cfg = dict(SMALL, texture_enabled=True)
    K, frames = small_stream(4)
    g, o = make_pair(cfg, K)
    g.set_option("semantics", 1)
    for R, T, d in frames:
        tex = rng.integers(0, 256, size=d.shape + (3,), dtype=np.uint8)
        g.recast_depth_to_map(R, T, d, tex)
        so = o.integrate_depth(R, T, d, tex, mode=FAITHFUL)
        sg = g.last_frame_stats()
        assert {k: sg[k] for k in STAT_KEYS} == {k: so[k] for k in STAT_KEYS}
    R, T = syn.camera_pose(3)
    dd = rng.normal(size=(4000, 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    pts = (dd * rng.uniform(0.3, 4.0, size=(4000, 1))).astype(np.float32)
    rgb = rng.integers(0, 256, size=(4000, 3), dtype=np.uint8)
    g.recast_pcl_to_map(R, T, pts, rgb)
    o.integrate_points(R, T, pts, rgb, mode=FAITHFUL)
    e = g.export_submap()
    assert e["color"].shape[0] == e["TSDF"].shape[0] > 100000 and (np.asarray(e["color"]).view(np.uint16) != 0).any()
    assert_export_equal(e, o.export_sparse(), "sequential, textured")


def test_sequential_fusion_textured_equals_faithful(hip_lib):
    """semantics = 1 on a global map: fuse_submaps replays the reference's racy running average sequentially (submap cells in struct-for order, the seven
    corners in loop order, colours before TSDF with the old weight).  Two textured 120 x 160 submaps with tilted base poses: the fused map, colours included,
    == the FAITHFUL fusion of the FAITHFUL submaps, bit for bit (NaN bit patterns included); fusing twice gives the same map."""
    from oracle import FAITHFUL, OracleTSDF
    from taichislam_amd.mapping import DenseTSDF
    from util import sort_export
    rng = np.random.default_rng(9)
    cfg = dict(SMALL, texture_enabled=True, max_submap_num=8)
    K, frames = small_stream(4)
    g, o = make_pair(cfg, K)
    g.set_option("semantics", 1)

    def tilt(R, a):
        return R @ np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    bases = [(tilt(frames[0][0], 0.13), frames[0][1] + np.array([0.011, -0.007, 0.003])), (tilt(frames[2][0], -0.21), frames[2][1] + np.array([-0.02, 0.013, 0.017]))]
    for sid, fr in ((0, frames[:2]), (1, frames[2:])):
        g.set_base_pose_submap(sid, *bases[sid]); o.set_base_pose_submap(sid, *bases[sid])
        for R, T, d in fr:
            tex = rng.integers(0, 256, size=d.shape + (3,), dtype=np.uint8)
            g.recast_depth_to_map(R, T, d, tex); o.integrate_depth(R, T, d, tex, mode=FAITHFUL)
        g.switch_to_next_submap(); o.set_active_submap(sid + 1)
    gcfg = dict(cfg, is_global_map=True)
    gg, og = DenseTSDF(**gcfg), OracleTSDF(**gcfg)
    gg.set_option("semantics", 1)
    for sid in (0, 1):
        gg.set_base_pose_submap(sid, *bases[sid]); og.set_base_pose_submap(sid, *bases[sid])
    og.fuse_submaps(o, mode=FAITHFUL)
    eo = sort_export(og.export_sparse())
    for _ in range(2):
        gg.fuse_submaps(g)
        eg = sort_export(gg.export_submap())
        assert eg["indices"].shape[0] > 50000 and np.array_equal(eg["indices"], eo["indices"])
        for k in ("TSDF", "W_TSDF", "occupy", "color"):
            assert np.array_equal(eg[k], eo[k]), k


@pytest.mark.parametrize("opts", [{"seq_impl": 0}, {"overlap": 0}, {"overlap": 3}])
def test_sequential_mode_variants_equal_faithful(hip_lib, opts):
    """The same small stream through round 3's global-sort form (seq_impl 0, kept as a cross-check), one frame at a time on the main stream
    (overlap 0) and batches of three: every form ends with the FAITHFUL map."""
    from oracle import FAITHFUL
    K, frames = small_stream(7)
    g, o = make_pair(SMALL, K)
    g.set_option("semantics", 1)
    for k, v in opts.items():
        g.set_option(k, v)
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None)
        so = o.integrate_depth(R, T, d, mode=FAITHFUL)
    sg = g.last_frame_stats()
    assert {k: sg[k] for k in STAT_KEYS} == {k: so[k] for k in STAT_KEYS}
    assert_export_equal(g.export_submap(), o.export_sparse(), f"sequential, {opts}")


def test_sequential_mode_outside_the_division_free_range(hip_lib):
    """Rays of up to 75 m into 50 cm voxels: signed distances beyond the +-60 the division-free update is proven for, so every (frame, brick) is
    flagged and both replay roles walk the literal expression (IEEE division) -- the path no benchmark scene ever takes.  Still FAITHFUL, bit for bit."""
    from oracle import FAITHFUL
    cfg = dict(map_scale=[204.8, 204.8], voxel_scale=0.4, num_voxel_per_blk_axis=16, max_ray_length=80.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
    h, w = 120, 160
    K = syn.scaled_intrinsics(h, w)
    g, o = make_pair(cfg, K)
    g.set_option("semantics", 1)
    for f in range(4):
        R, T = syn.camera_pose(3 * f, orbit=5.0)
        d = syn.sphere_room_depth(R, T, h, w, radius=69.0, K=K)        # optical-axis depth up to 64 m (uint16 millimetres hold 65.5 m); ranges beyond that
        assert d.max() > 60000
        g.recast_depth_to_map(R, T, d, None)
        so = o.integrate_depth(R, T, d, mode=FAITHFUL)
    sg = g.last_frame_stats()
    assert {k: sg[k] for k in STAT_KEYS} == {k: so[k] for k in STAT_KEYS} and sg["steps"] > 100000
    e = g.export_submap()
    assert np.abs(np.asarray(e["TSDF"], np.float32)).max() > 60.0
    assert_export_equal(e, o.export_sparse(), "sequential, long rays")


def test_sequential_mode_tuple_capacity_is_loud(hip_lib):
    """A frame that yields more ray steps than seq_tuple_cap is dropped as a whole and reported (TSL_ERR_CAPACITY), the map keeps the frames before it."""
    from oracle import FAITHFUL
    from taichislam_amd._lib import TslError
    K, frames = small_stream(3)
    g, o = make_pair(SMALL, K)
    g.set_option("semantics", 1)
    g.set_option("seq_tuple_cap", 1 << 16)               # a 120 x 160 frame of the stream has ~330 k steps
    with pytest.raises(TslError, match="capacity"):
        g.recast_depth_to_map(*frames[0], None)
        g.sync()
    assert g.count_active() == 0


def test_tuple_capacity_overflow_in_a_queued_batch_leaves_the_other_frames_exact(hip_lib):
    """k_seq_group's persistent workgroups read the frame's fail word while other workgroups of the same launch may be setting it at their
    tuple-cap check; the decision has to be one per workgroup (thread 0 reads, a barrier, everybody takes its answer), or the waves of a workgroup meet
    different barriers.  Eight frames queued back to back, every second one too big for seq_tuple_cap: the big ones are dropped as wholes and reported,
    the small ones -- same batch, same launch -- must come out exactly as the oracle integrates them."""
    from oracle import FAITHFUL
    from taichislam_amd._lib import TslError
    K, frames = small_stream(8)
    for rep in range(3):
        g, o = make_pair(SMALL, K)
        g.set_option("semantics", 1)
        g.set_option("seq_tuple_cap", 1 << 16)           # a full 120 x 160 frame has ~330 k steps; an 18-row band of it ~50 k
        kept = 0
        for f, (R, T, d) in enumerate(frames):
            if f % 2 == 0:
                d = d.copy(); d[:50] = 0; d[68:] = 0     # rows 50..67 only
                st = o.integrate_depth(R, T, d, mode=FAITHFUL)
                assert 0 < st["steps"] < (1 << 16)
                kept += 1
            g.recast_depth_to_map(R, T, d, None)
        with pytest.raises(TslError, match="capacity"):
            g.sync()
        assert kept == 4
        # (occupancy is marked by phase A, in front of the tuple-capacity check of k_seq_group: a dropped frame leaves its surface voxels' occupy bytes set on
        #  voxels the kept frames observed -- the error is reported, the TSDF / W values and the voxel set are the kept frames' alone, which is what is compared)
        a, b = sort_export(g.export_submap()), sort_export(o.export_sparse())
        assert a["indices"].shape == b["indices"].shape and np.array_equal(a["indices"], b["indices"]), f"repetition {rep}: voxel sets differ"
        assert np.array_equal(a["TSDF"], b["TSDF"]) and np.array_equal(a["W_TSDF"], b["W_TSDF"]), f"repetition {rep}: the frames that did not overflow differ from the oracle"
        assert (a["occupy"] >= b["occupy"]).all()


def test_rebuilding_the_literal_scratch_does_not_double_count_its_bytes(hip_lib):
    """set_option("seq_tuple_cap") on a live handle frees and reallocates the literal scratch; tsl_tsdf_memory_bytes must follow."""
    from taichislam_amd.mapping import DenseTSDF
    g = DenseTSDF(**SMALL)
    b0 = g.memory_bytes()
    g.set_option("semantics", 1)
    b1 = g.memory_bytes()
    assert b1 > b0
    g.set_option("seq_tuple_cap", 1 << 22)
    b2 = g.memory_bytes()
    g.set_option("seq_tuple_cap", 1 << 23)
    assert g.memory_bytes() == b1 and b0 < b2 < b1
    g.set_option("semantics", 0)
