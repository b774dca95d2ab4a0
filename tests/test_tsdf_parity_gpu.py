"""GPU parity: HIP DenseTSDF (through the C-ABI) vs the CPU oracle in BATCHED mode -- bit-exact.

Reference behaviour under test: taichi_slam/mapping/dense_tsdf.py:157-270 (integration), :412-454 (sparse
export/import), :339-389 (particle exports)."""
import numpy as np
import pytest

from taichislam_amd.utils import synthetic as syn
from util import C2, SMALL, assert_export_equal, make_pair, small_stream, sorted_rows

pytestmark = pytest.mark.gpu

STAT_KEYS = ("p_used", "p_valid", "p_oob", "v_pcl", "v_skipped", "steps", "steps_oob", "unique", "bricks")


def _run_both(g, o, frames):
    from oracle import BATCHED
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, np.array([], dtype=int))
        so = o.integrate_depth(R, T, d, mode=BATCHED)
        sg = g.last_frame_stats()
        assert {k: sg[k] for k in STAT_KEYS} == {k: so[k] for k in STAT_KEYS}


def test_small_stream_bit_exact(hip_lib):
    K, frames = small_stream(4)
    g, o = make_pair(SMALL, K)
    _run_both(g, o, frames)
    assert g.count_active() == o.count_active() > 10000
    assert_export_equal(g.export_submap(), o.export_sparse(), "small stream")
    gi, go = g.export_occupied()
    oi, oo = o.export_occupied()
    assert np.array_equal(sorted_rows(gi), sorted_rows(oi)) and gi.shape[0] > 100


@pytest.mark.parametrize("variant,split", [(0, 1), (0, 4), (1, 1), (1, 4), (1, 64), (2, 1), (2, 2), (2, 4), (2, 8)])
def test_kernel_variants_agree(hip_lib, variant, split):
    K, frames = small_stream(2)
    g, o = make_pair(SMALL, K)
    g.set_option("variant", variant)
    g.set_option("split", split)
    _run_both(g, o, frames)
    assert_export_equal(g.export_submap(), o.export_sparse(), f"variant {variant} split {split}")


@pytest.mark.parametrize("wg,spt,chunks", [(256, 4, 1), (256, 4, 2), (256, 4, 8), (512, 4, 1), (512, 4, 3), (512, 2, 1), (512, 2, 2), (512, 2, 5)])
def test_integrate_workgroup_and_part_sizes(hip_lib, wg, spt, chunks):
    """The brick kernel's geometry (threads per workgroup, chunks per part: which bricks are split over workgroups and merged through
    HBM) must not show in the result."""
    K, frames = small_stream(2)
    g, o = make_pair(SMALL, K)
    g.set_option("wg", wg)
    g.set_option("spt", spt)
    g.set_option("chunks", chunks)
    _run_both(g, o, frames)
    assert_export_equal(g.export_submap(), o.export_sparse(), f"wg {wg} spt {spt} chunks {chunks}")


@pytest.mark.parametrize("wg,spt,unit,ramp,split", [(512, 4, 0, 0, 0), (256, 4, 0, 0, 0), (512, 4, 200, 2, 0), (256, 4, 1 << 20, 0, 0), (512, 2, 0, 0, 0), (512, 2, 200, 2, 0),
                                                       (512, 2, 1 << 20, 0, 0), (512, 2, 200, 0, 1), (512, 2, 0, 0, 1), (256, 4, 200, 2, 1)])
def test_units_and_parts_over_full_batches(hip_lib, wg, spt, unit, ramp, split):
    """Eleven frames with nothing read in between (a full batch of eight + three, or half batches first): with the unit limit at 0
    every brick is split into parts and merged through the per-(frame, brick) slab slots, with a huge limit every brick is a unit
    walked by one workgroup over all frames of the batch; any mix in between must give the same map -- also when the parts are walked
    by a launch of their own on the batch's phase-A stream (split_launch)."""
    from oracle import BATCHED
    K, frames = small_stream(11)
    g, o = make_pair(SMALL, K)
    g.set_option("wg", wg); g.set_option("spt", spt); g.set_option("unit", unit); g.set_option("ramp", ramp); g.set_option("split_launch", split)
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None)
        o.integrate_depth(R, T, d, mode=BATCHED)
    assert_export_equal(g.export_submap(), o.export_sparse(), f"wg {wg} spt {spt} unit {unit} ramp {ramp} split_launch {split}")


@pytest.mark.parametrize("opts", [{"adaptive": 1}, {"adaptive": 1, "ramp": 0}, {"bgrid": 100}, {"bgrid": 20}, {"ramp_size": 2}, {"ramp_size": 6, "ramp": 3}, {"unit_floor": 0, "unit": 300},
                                  {"split_launch": 1, "ugrid": 40, "pgrid": 60}])
def test_scheduling_options_do_not_show_in_the_map(hip_lib, opts):
    """How the queued frames are issued (adaptive: as soon as phase A of the previous batch is done; the length and number of the short batches
    after the pipeline ran dry) and how many workgroups the brick kernel is given are scheduling choices: the map must not depend on them."""
    from oracle import BATCHED
    K, frames = small_stream(13)
    g, o = make_pair(SMALL, K)
    for k, v in opts.items():
        g.set_option(k, v)
    so = None
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None)
        so = o.integrate_depth(R, T, d, mode=BATCHED)
    sg = g.last_frame_stats()
    assert {k: sg[k] for k in STAT_KEYS} == {k: so[k] for k in STAT_KEYS}, opts
    assert_export_equal(g.export_submap(), o.export_sparse(), f"options {opts}")


@pytest.mark.parametrize("unit,half,split", [(1 << 20, 300, 0), (3000, 100, 0), (1 << 20, 1 << 20, 0), (2000, 0, 0), (3000, 100, 1)])
def test_middle_tier_first_frames_as_unit_later_frames_as_parts(hip_lib, unit, half, split):
    """Bricks between `unit_half` and `unit` segments per batch: their first frames are walked as a unit, their later frames as parts that
    k_apply_slab applies on top of what the unit wrote.  Any split must give the oracle's map."""
    from oracle import BATCHED
    K, frames = small_stream(11)
    g, o = make_pair(SMALL, K)
    g.set_option("unit", unit); g.set_option("unit_half", half); g.set_option("ramp", 0); g.set_option("split_launch", split)
    so = None
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None)
        so = o.integrate_depth(R, T, d, mode=BATCHED)
    sg = g.last_frame_stats()
    assert {k: sg[k] for k in STAT_KEYS} == {k: so[k] for k in STAT_KEYS}
    assert_export_equal(g.export_submap(), o.export_sparse(), f"unit {unit} unit_half {half}")


@pytest.mark.parametrize("which", [0, 1, 2])
def test_arithmetic_shortcuts_hold_for_every_float(hip_lib, which):
    """The kernels round half away from zero with add+truncate and take square roots without the library's rescaling;
    both must agree with the plain forms (ti.round, sqrtf) for all 2^32 float patterns.  2: the sequential replay divides by Wmax + w
    through a reciprocal product with two FMA corrections once a voxel's weight is saturated -- against IEEE division on 2^32 operand pairs."""
    import ctypes
    from taichislam_amd import _lib
    bad = ctypes.c_int64(-1)
    _lib.check(_lib.lib().tsl_selftest(which, ctypes.byref(bad)))
    assert bad.value == 0


def test_fast_division_is_verified_and_optional(hip_lib):
    """x / voxel_scale is replaced by an fma-refined reciprocal product only after the device has checked it against IEEE
    division for every float; forcing IEEE division must give the same map."""
    from taichislam_amd.mapping import DenseTSDF
    K, frames = small_stream(2)
    for vs in (0.02, 0.04, 0.05, 0.1):
        m = DenseTSDF(map_scale=[5.12, 5.12], voxel_scale=vs, num_voxel_per_blk_axis=16, max_ray_length=3.0)
        assert m.get_option("fastdiv") in (0, 1)
    g, o = make_pair(SMALL, K)
    assert g.get_option("fastdiv") == 1
    g.set_option("fastdiv", 0)
    assert g.get_option("fastdiv") == 0
    _run_both(g, o, frames)
    assert_export_equal(g.export_submap(), o.export_sparse(), "IEEE division path")


def test_full_size_c2_bit_exact(hip_lib):
    """BASELINE.json configs[1]: 640x480 depth into 512^3 @ 2 cm; two frames against the oracle."""
    g, o = make_pair(C2, syn.K_DEPTH)
    frames = list(syn.sphere_room_stream(2))
    _run_both(g, o, frames)
    st = g.last_frame_stats()
    assert st["p_valid"] == 76800 and st["steps"] > 3_000_000 and st["steps_oob"] == 0
    assert_export_equal(g.export_submap(), o.export_sparse(), "C2")


def test_full_size_full_batches_bit_exact(hip_lib):
    """The benchmark configuration with nothing read between the frames: 20 frames go out as [8][8][4] (ramp 0) and as [4][4][8][4]
    (default) -- a full batch walks units over eight frames and merges the bricks next to the sensor through eight slab slots applied
    in frame order by the last arriver.  Both must equal the oracle bit for bit, frame counters included."""
    from oracle import BATCHED
    from taichislam_amd.mapping import DenseTSDF
    frames = list(syn.sphere_room_stream(20))
    g, o = make_pair(C2, syn.K_DEPTH)
    g.set_option("ramp", 0)
    h = DenseTSDF(**C2); h.set_dep_camera_intrinsic(syn.K_DEPTH)
    s = DenseTSDF(**C2); s.set_dep_camera_intrinsic(syn.K_DEPTH); s.set_option("split_launch", 1)
    so = None
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None)
        h.recast_depth_to_map(R, T, d, None)
        s.recast_depth_to_map(R, T, d, None)
        so = o.integrate_depth(R, T, d, mode=BATCHED)
    want = o.export_sparse()
    for m, what in ((g, "[8][8][4]"), (h, "[4][4][8][4]"), (s, "[4][4][8][4], the parts in a launch of their own")):
        sg = m.last_frame_stats()
        assert {k: sg[k] for k in STAT_KEYS} == {k: so[k] for k in STAT_KEYS}, what
        assert_export_equal(m.export_submap(), want, f"C2, 20 frames as {what}")


def test_full_size_properties(hip_lib):
    """Size-independent properties at the benchmark size: determinism across runs and variants, export/import
    round trip, count == export length, W monotone and clamped."""
    from taichislam_amd.mapping import DenseTSDF
    frames = list(syn.sphere_room_stream(6))
    exports = []
    for variant, split in ((2, 2), (1, 4)):
        m = DenseTSDF(**C2)
        m.set_dep_camera_intrinsic(syn.K_DEPTH)
        m.set_option("variant", variant)
        m.set_option("split", split)
        wprev = None
        for R, T, d in frames:
            m.recast_depth_to_map(R, T, d, None)
        e = m.export_submap()
        assert e["TSDF"].shape[0] == m.count_active()
        w = e["W_TSDF"].astype(np.float32)
        assert w.min() > 0 and w.max() <= 1000.0 and np.isfinite(e["TSDF"].astype(np.float32)).all()
        exports.append(e)
    assert_export_equal(exports[0], exports[1], "determinism across variants")
    m2 = DenseTSDF(**C2)
    e = exports[0]
    m2.load_numpy(0, e["indices"], e["TSDF"], e["W_TSDF"], e["occupy"], e["color"])
    assert_export_equal(m2.export_submap(), e, "export -> import -> export")


def test_points_input(hip_lib):
    """recast_pcl_to_map (dense_tsdf.py:167-186): range gate on the norm, z := range."""
    rng = np.random.default_rng(7)
    g, o = make_pair(SMALL, syn.K_DEPTH)
    R, T = syn.camera_pose(3)
    d = rng.normal(size=(20000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = (d * rng.uniform(0.2, 6.0, size=(20000, 1))).astype(np.float32)      # some beyond max_ray_length
    pts[:5] = 0.0                                                               # degenerate: at the sensor origin
    from oracle import BATCHED
    g.recast_pcl_to_map(R, T, pts, np.array([]))
    so = o.integrate_points(R, T, pts, None, mode=BATCHED)
    sg = g.last_frame_stats()
    assert {k: sg[k] for k in STAT_KEYS} == {k: so[k] for k in STAT_KEYS}
    assert so["v_skipped"] >= 1
    assert_export_equal(g.export_submap(), o.export_sparse(), "points")


@pytest.mark.parametrize("overlap", [0, 1, 2, 3, 4])
def test_queued_batches_bit_exact(hip_lib, overlap):
    """Frames are queued and issued in batches (phase A of a batch is one sequence of launches with grid.y = frame); nothing is
    read back between the frames here, the stream mixes image sizes and a point cloud, and its length is not a multiple of the
    batch size."""
    from oracle import BATCHED
    K, frames = small_stream(7)
    g, o = make_pair(SMALL, K)
    g.set_option("overlap", overlap)
    rng = np.random.default_rng(overlap)
    d = rng.normal(size=(4000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = (d * rng.uniform(0.5, 4.0, size=(4000, 1))).astype(np.float32)
    for f, (R, T, dep) in enumerate(frames):
        if f == 2:
            dep = dep[:90, :120].copy()                                   # a smaller image inside a batch
        if f == 4:
            g.recast_pcl_to_map(R, T, pts, np.array([])); o.integrate_points(R, T, pts, None, mode=BATCHED)
            continue
        g.recast_depth_to_map(R, T, dep, np.array([], dtype=int)); o.integrate_depth(R, T, dep, mode=BATCHED)
    assert_export_equal(g.export_submap(), o.export_sparse(), f"queued batches, overlap {overlap}")


@pytest.mark.parametrize("host_input", [False, True])
def test_long_stream_cycles_every_batch_slot(hip_lib, host_input):
    """34 frames without a read in between: every batch slot (four, sharing three phase-A streams) is reused at least twice, the
    last batch is partial; with numpy inputs every frame also goes through the copy stream into its set's staging area, and the
    caller's buffer is overwritten right after each call."""
    from oracle import BATCHED
    import torch
    K, frames = small_stream(34)
    g, o = make_pair(SMALL, K)
    scratch = np.empty_like(frames[0][2])
    for R, T, d in frames:
        if host_input:
            scratch[...] = d
            g.recast_depth_to_map(R, T, scratch, None)
            scratch[...] = 0                                               # the library must have taken its copy
        else:
            g.recast_depth_to_map(R, T, torch.from_numpy(d.view(np.int16)).cuda(), None)
        o.integrate_depth(R, T, d, mode=BATCHED)
    assert_export_equal(g.export_submap(), o.export_sparse(), f"34 queued frames, host_input {host_input}")


@pytest.mark.parametrize("group", [0, 1])
def test_pixel_grouping_paths(hip_lib, group):
    """Pixels of one sensor voxel are summed in raster order with per-add f16 rounding (dense_tsdf.py:230-234) whether the
    groups come from the stable radix sort (group=0) or from the sensor-voxel hash table (group=1)."""
    K, frames = small_stream(3)
    g, o = make_pair(SMALL, K)
    g.set_option("group", group)
    assert g.get_option("group") == group
    _run_both(g, o, frames)
    assert_export_equal(g.export_submap(), o.export_sparse(), f"group {group}")


@pytest.mark.parametrize("group", [0, 1])
def test_wide_sensor_grid_keys(hip_lib, group):
    """max_ray_length / voxel_scale = 500: the sensor-centred grid needs 11 bits per axis, so sensor voxels are keyed with
    64-bit Morton codes (sort path) / 64-bit hash keys, and most of every ray lies outside the 2.56 m map."""
    cfg = dict(map_scale=[2.56, 2.56], voxel_scale=0.01, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3,
               internal_voxels=10, recast_step=2, texture_enabled=False)
    K, frames = small_stream(2, radius=1.6, orbit=0.1)
    g, o = make_pair(cfg, K)
    g.set_option("group", group)
    _run_both(g, o, frames)
    assert g.last_frame_stats()["steps_oob"] > 0
    assert_export_equal(g.export_submap(), o.export_sparse(), f"wide keys, group {group}")


@pytest.mark.parametrize("group", [0, 1])
def test_crowded_sensor_voxels(hip_lib, group):
    """Point clouds that put tens to thousands of points into one sensor voxel: per-thread replay, the workgroup-sorted
    path for big groups, and the f16 saturation of the sums that goes with them."""
    from oracle import BATCHED
    rng = np.random.default_rng(23)
    g, o = make_pair(SMALL, syn.K_DEPTH)
    g.set_option("group", group)
    R, T = syn.camera_pose(1)
    centres = np.array([[1.0, 0.3, 0.2], [-0.8, 1.1, 0.1], [0.5, -1.4, 0.3], [2.0, 2.0, 0.5], [-1.5, -0.7, 0.4]])
    sizes = [40, 49, 300, 5000, 9]
    blobs = [c + rng.uniform(-0.015, 0.015, size=(n, 3)) for c, n in zip(centres, sizes)]
    d = rng.normal(size=(3000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = np.concatenate(blobs + [d * rng.uniform(0.5, 4.0, size=(3000, 1))]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    g.recast_pcl_to_map(R, T, pts, np.array([]))
    so = o.integrate_points(R, T, pts, None, mode=BATCHED)
    sg = g.last_frame_stats()
    assert {k: sg[k] for k in STAT_KEYS} == {k: so[k] for k in STAT_KEYS}
    assert_export_equal(g.export_submap(), o.export_sparse(), f"crowded voxels, group {group}")
    g.recast_pcl_to_map(R, T, pts[::-1].copy(), np.array([]))                   # the hash table must be clean again
    o.integrate_points(R, T, pts[::-1].copy(), None, mode=BATCHED)
    assert_export_equal(g.export_submap(), o.export_sparse(), f"crowded voxels second frame, group {group}")


@pytest.mark.parametrize("h,w,step", [(121, 163, 2), (120, 160, 3), (97, 131, 1)])
def test_ragged_image_sizes(hip_lib, h, w, step):
    cfg = dict(SMALL, recast_step=step)
    K = syn.scaled_intrinsics(h, w)
    g, o = make_pair(cfg, K)
    frames = []
    for f in range(2):
        R, T = syn.camera_pose(f)
        frames.append((R, T, syn.sphere_room_depth(R, T, h, w, K=K)))
    _run_both(g, o, frames)
    assert_export_equal(g.export_submap(), o.export_sparse(), f"ragged {h}x{w}/{step}")


def test_empty_and_gated_depth(hip_lib):
    """All-zero depth, depth beyond max_ray_length and below min_ray_length integrate nothing (dense_tsdf.py:196-199)."""
    K, frames = small_stream(1)
    g, o = make_pair(SMALL, K)
    R, T, d = frames[0]
    for img in (np.zeros_like(d), np.full_like(d, 60000), np.full_like(d, 100)):
        g.recast_depth_to_map(R, T, img, None)
        st = g.last_frame_stats()
        assert st["p_valid"] == 0 and st["v_pcl"] == 0 and st["steps"] == 0
    assert g.count_active() == 0
    half = d.copy()
    half[:, : d.shape[1] // 2] = 0
    _run_both(g, o, [(R, T, half)])
    assert_export_equal(g.export_submap(), o.export_sparse(), "half-empty")


def test_rays_leaving_the_volume(hip_lib):
    """A map smaller than the scene: out-of-volume ray-steps are skipped and counted."""
    cfg = dict(SMALL, map_scale=[2.56, 2.56])
    K, frames = small_stream(2)
    g, o = make_pair(cfg, K)
    _run_both(g, o, frames)
    assert g.last_frame_stats()["steps_oob"] > 0
    assert_export_equal(g.export_submap(), o.export_sparse(), "small volume")


def test_submap_base_pose_and_switch(hip_lib):
    """Integration goes into the active submap in that submap's frame (mapping_common.py:91-100, dense_tsdf.py:238)."""
    K, frames = small_stream(4)
    cfg = dict(SMALL, max_submap_num=8)
    g, o = make_pair(cfg, K)
    R0, T0, _ = frames[0]
    g.set_base_pose_submap(0, R0, T0)
    o.set_base_pose_submap(0, R0, T0)
    _run_both(g, o, frames[:2])
    assert_export_equal(g.export_submap(), o.export_sparse(), "submap 0")
    assert g.switch_to_next_submap() == 1
    o.set_active_submap(1)
    R2, T2, _ = frames[2]
    g.set_base_pose_submap(1, R2, T2)
    o.set_base_pose_submap(1, R2, T2)
    _run_both(g, o, frames[2:])
    assert_export_equal(g.export_submap(), o.export_sparse(), "submap 1")


def test_queue_keeps_every_frames_own_parameters(hip_lib):
    """Frames wait in a queue until a batch is full: intrinsics, base pose and active submap are the ones that were current
    when each frame was queued, and a handle can be dropped with frames still queued."""
    from oracle import BATCHED
    K, frames = small_stream(6)
    K2 = K.copy(); K2[0] *= 1.07; K2[4] *= 0.95; K2[2] -= 2.0
    cfg = dict(SMALL, max_submap_num=8)
    g, o = make_pair(cfg, K)
    for f, (R, T, d) in enumerate(frames):
        if f == 1:
            g.set_dep_camera_intrinsic(K2); o.set_intrinsics(K2, K2)
        if f == 3:
            assert g.switch_to_next_submap() == 1; o.set_active_submap(1)
            g.set_base_pose_submap(1, R, T); o.set_base_pose_submap(1, R, T)
        g.recast_depth_to_map(R, T, d, None); o.integrate_depth(R, T, d, mode=BATCHED)
    assert_export_equal(g.export_submap(), o.export_sparse(), "submap 1 after queued parameter changes")
    g.active_submap_id[None] = 0; o.set_active_submap(0)
    assert_export_equal(g.export_submap(), o.export_sparse(), "submap 0 after queued parameter changes")
    g.recast_depth_to_map(*frames[0][:2], frames[0][2], None)
    g.recast_depth_to_map(*frames[1][:2], frames[1][2], None)
    del g                                                                  # two frames still queued


@pytest.mark.parametrize("queued", [1, 2, 3])
def test_queue_is_flushed_by_every_reader(hip_lib, tmp_path, queued):
    """saveMap-then-exit style callers: 1-3 frames are still queued (a batch is four) when the map is read -- every reader must see them."""
    from oracle import BATCHED
    from taichislam_amd.mapping import DenseTSDF
    K, frames = small_stream(4 + queued)
    g, o = make_pair(SMALL, K)
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None); o.integrate_depth(R, T, d, mode=BATCHED)
    want = o.export_sparse()
    path = str(tmp_path / "map.npy")
    g.saveMap(path)                                                        # first reader after the queued frames
    back = DenseTSDF.loadMap(path)
    assert back.count_active() == want["TSDF"].shape[0] == g.count_active()
    assert_export_equal(back.export_submap(), want, f"saveMap / loadMap with {queued} queued frames")
    assert_export_equal(g.export_submap(), want, f"export_submap with {queued} queued frames")


def test_particle_exports(hip_lib):
    """cvt_TSDF_surface_to_voxels / cvt_TSDF_to_voxels_slice (dense_tsdf.py:339-389) as sorted sets."""
    K, frames = small_stream(3)
    cfg = dict(SMALL, disp_ceiling=1.0, disp_floor=-1.0)
    g, o = make_pair(cfg, K)
    _run_both(g, o, frames)
    g.cvt_TSDF_surface_to_voxels()
    n = g.num_TSDF_particles[None]
    oxyz, orgb, on = o.surface_voxels()
    assert n == on > 100
    gxyz = g.export_TSDF_xyz.to_numpy()[:n]
    grgb = g.export_color.to_numpy()[:n]
    a = sorted_rows(np.concatenate([gxyz, grgb], 1))
    b = sorted_rows(np.concatenate([oxyz, orgb], 1))
    assert np.array_equal(a[:, :3], b[:, :3])
    assert np.allclose(a[:, 3:], b[:, 3:], atol=1e-6)
    g.cvt_TSDF_to_voxels_slice(0.2, dz=1.5)
    n = g.num_TSDF_particles[None]
    sxyz, sval, _, sn = o.slice_voxels(0.2, 1.5)
    assert n == sn > 100
    a = sorted_rows(np.concatenate([g.export_TSDF_xyz.to_numpy()[:n], g.export_TSDF.to_numpy()[:n, None]], 1))
    b = sorted_rows(np.concatenate([sxyz, sval[:, None]], 1))
    assert np.array_equal(a, b)


def test_reset_and_reuse(hip_lib):
    K, frames = small_stream(2)
    g, o = make_pair(SMALL, K)
    _run_both(g, o, frames)
    assert g.bricks_in_use() > 0
    g.reset()
    assert g.count_active() == 0 and g.bricks_in_use() == 0
    o.reset()
    _run_both(g, o, frames[1:])
    assert_export_equal(g.export_submap(), o.export_sparse(), "after reset")


def test_device_pointer_input(hip_lib):
    import torch
    K, frames = small_stream(2)
    g, o = make_pair(SMALL, K)
    from oracle import BATCHED
    for R, T, d in frames:
        dt = torch.from_numpy(d.view(np.int16)).cuda()
        g.recast_depth_to_map(R, T, dt, None)
        g.sync()
        o.integrate_depth(R, T, d, mode=BATCHED)
    assert_export_equal(g.export_submap(), o.export_sparse(), "device pointer")


def test_capacity_error_is_loud(hip_lib):
    from taichislam_amd._lib import TslError
    from taichislam_amd.mapping import DenseTSDF
    K, frames = small_stream(1)
    m = DenseTSDF(**SMALL, max_bricks=8)
    m.set_dep_camera_intrinsic(K)
    R, T, d = frames[0]
    m.recast_depth_to_map(R, T, d, None)
    with pytest.raises(TslError):
        m.last_frame_stats()


def test_temporary_device_tensors_are_safe(hip_lib):
    """A drop-in caller builds a fresh CUDA tensor per frame on torch's stream and lets it go right after the call, as it may with
    the reference's synchronous recast_* (dense_tsdf.py:157-165).  The frames are only queued here: the shim has to order the
    library's stream behind torch's and keep the allocator from recycling the buffers before the queued kernels have read them."""
    import torch
    from oracle import BATCHED
    K, frames = small_stream(7)
    g, o = make_pair(SMALL, K)
    pts = []
    for f, (R, T, d) in enumerate(frames):
        if f % 3 == 2:          # a point-cloud frame between depth frames (forces the queue to be issued early)
            rng = np.random.default_rng(f)
            p = (rng.standard_normal((3000, 3)) * 0.8).astype(np.float32)
            x = torch.from_numpy(p).cuda() * 1.0                      # produced asynchronously on torch's stream
            g.recast_pcl_to_map(R, T, x, None)
            o.integrate_points(R, T, p, mode=BATCHED)
            del x
        else:
            t = torch.from_numpy(d.view(np.int16)).cuda()
            t2 = t + 0                                                  # a second temporary, written by a kernel that may still run
            g.recast_depth_to_map(R, T, t2, None)
            o.integrate_depth(R, T, d, mode=BATCHED)
            del t, t2
        # allocations of the same size right away: without record_stream these get the blocks just released
        junk = [torch.full((120, 160), 7, dtype=torch.int16, device="cuda") for _ in range(4)]
        junk.append(torch.full((3000, 3), 9.0, device="cuda"))
        pts.append(junk[-1].sum())
        del junk
    assert_export_equal(g.export_submap(), o.export_sparse(), "temporaries")


def test_capacity_errors_are_reported_and_do_not_stick(hip_lib):
    """A brick pool that is too small drops bricks -- loudly: the next call that synchronises returns TSL_ERR_CAPACITY once, the
    handle keeps working, and after reset() the same bricks can be allocated again (no poisoned table entries)."""
    from taichislam_amd import _lib
    from taichislam_amd.mapping import DenseTSDF
    K = syn.scaled_intrinsics(120, 160)
    (Ra, Ta), (Rb, Tb) = syn.camera_pose(0), syn.camera_pose(0, start_deg=120.0)
    da, db = syn.sphere_room_depth(Ra, Ta, 120, 160, K=K), syn.sphere_room_depth(Rb, Tb, 120, 160, K=K)
    ref = DenseTSDF(**SMALL); ref.set_dep_camera_intrinsic(K)
    ref.recast_depth_to_map(Ra, Ta, da, None)
    na, ba = ref.count_active(), ref.bricks_in_use()
    ref.recast_depth_to_map(Rb, Tb, db, None)
    nab, bab = ref.count_active(), ref.bricks_in_use()
    assert bab > ba + 20
    cap = ba + 10                                    # room for the first frame, not for the second
    g = DenseTSDF(**SMALL, max_bricks=cap, max_frame_bricks=4096)
    g.set_dep_camera_intrinsic(K)
    g.recast_depth_to_map(Ra, Ta, da, None)
    g.sync()
    assert g.count_active() == na
    g.recast_depth_to_map(Rb, Tb, db, None)
    with pytest.raises(_lib.TslError, match="brick pool"):
        g.sync()
    g.sync()                                         # reported once, the handle keeps working
    assert g.bricks_in_use() == cap and na < g.count_active() < nab
    try:
        g.reset()
    except _lib.TslError:
        pass
    assert g.count_active() == 0 and g.bricks_in_use() == 0
    g.recast_depth_to_map(Rb, Tb, db, None)          # the bricks that did not fit before are allocatable now
    g.sync()
    only_b = DenseTSDF(**SMALL); only_b.set_dep_camera_intrinsic(K)
    only_b.recast_depth_to_map(Rb, Tb, db, None)
    assert_export_equal(g.export_submap(), only_b.export_submap(), "after reset")


def test_pointcloud2_payload_and_export_row_writes(hip_lib):
    """next-4 glue: the PointCloud2 data block interleaved on the device equals what the reference's node assembles from numpy copies
    (utils/ros_pcl_transfer.py:96-136, scripts/taichislam_node.py:420-425); `export_TSDF_xyz[i] = p` (tests/gen_topo_graph.py:64) works."""
    K, frames = small_stream(2)
    g, _ = make_pair(SMALL, K)
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None)
    g.cvt_TSDF_surface_to_voxels()
    n = g.num_TSDF_particles[None]
    assert n > 100
    xyz, rgb = g.export_TSDF_xyz.to_numpy()[:n], g.export_color.to_numpy()[:n]
    for has_rgb in (False, True):
        msg = g.pointcloud2(has_rgb=has_rgb)
        want = np.concatenate((xyz, rgb.astype(float)), axis=1) if has_rgb else xyz          # pub_to_ros :420-425
        assert msg["data"] == want.astype(np.float32).tobytes() and msg["width"] == n and msg["point_step"] == (24 if has_rgb else 12)
        assert [f["name"] for f in msg["fields"]] == list("xyzrgb" if has_rgb else "xyz") and msg["row_step"] == msg["point_step"] * n
    g.export_TSDF_xyz[n] = np.array([1.5, -2.5, 0.25])
    assert np.array_equal(g.export_TSDF_xyz.to_numpy(n + 1)[n], np.array([1.5, -2.5, 0.25], np.float32))
