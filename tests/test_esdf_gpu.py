"""GPU ESDF vs the oracle's Dijkstra over the same graph, and vs brute-force Euclidean distance."""
import numpy as np
import pytest

from util import SMALL, lin, make_pair, small_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[1, 0], ids=["wavefront", "regional"])
def esdf_mode(request, monkeypatch):
    """Every test of this file runs for both forms of the incremental update: esdf_mode 1, the raise / lower wavefront with parent directions (the
    default), and esdf_mode 0, the regional recompute.  The yardstick handles (option esdf_full) always recompute everything with esdf_mode 0."""
    monkeypatch.setenv("TSL_ESDF_MODE", str(request.param))
    return request.param


def test_esdf_matches_oracle_and_brute_force(hip_lib, esdf_mode):
    from oracle import BATCHED
    K, frames = small_stream(2)
    g, o = make_pair(SMALL, K)
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None)
        o.integrate_depth(R, T, d, mode=BATCHED)
    relaxed = g.update_esdf(max_dist=2.0)
    st = g.esdf_stats()
    assert st["incremental"] == 0 and st["total_bricks"] > 50 and relaxed == st["brick_relaxations"] >= st["region_bricks"]
    # (the regional recompute initialises every brick; the wavefront only reaches the bricks a value can be lowered in)
    assert st["region_bricks"] == st["total_bricks"] if esdf_mode == 0 else 0 < st["region_bricks"] <= st["total_bricks"]
    assert g.get_option("esdf_orphans") == 0
    gi, ge = g.export_esdf()
    oi, oe = o.esdf(max_dist=2.0)
    a, b = np.argsort(lin(gi)), np.argsort(lin(oi))
    assert gi.shape[0] == oi.shape[0] > 50000 and np.array_equal(gi[a], oi[b])
    assert np.array_equal(ge[a], oe[b]), f"max diff {np.abs(ge[a] - oe[b]).max()}"
    # sanity against geometry: behind/in front of the surface the magnitude grows away from the zero band
    e = g.export_submap()
    t = e["TSDF"].astype(np.float32)[np.argsort(lin(e["indices"]))]
    esdf = ge[a]
    fixed = np.abs(t) < SMALL["voxel_scale"]
    assert np.array_equal(esdf[fixed], t[fixed])
    assert (np.sign(esdf[~fixed]) == np.sign(t[~fixed])).all() and np.abs(esdf).max() <= 2.0


def test_esdf_analytic_sphere(hip_lib):
    """Sphere SDF: the quasi-Euclidean 26-neighbour distance over-estimates the true distance by at most ~9 %."""
    from taichislam_amd.mapping import DenseTSDF
    g = DenseTSDF(map_scale=[6.4, 6.4], voxel_scale=0.05, num_voxel_per_blk_axis=16)
    g.init_sphere(voxels=60, radius=0.8)
    g.update_esdf(max_dist=3.0)
    idx, esdf = g.export_esdf()
    true = np.linalg.norm(idx.astype(np.float64) * 0.05, axis=1) - 0.8
    assert np.abs(esdf - true).max() < 0.09 * np.abs(true).max() + 0.05
    assert np.abs(esdf - true)[np.abs(true) < 0.05].max() < 1e-3


def _esdf_sorted(m):
    i, e = m.export_esdf()
    o = np.argsort(lin(i))
    return i[o], e[o]


def test_incremental_update_equals_full_recompute_after_every_frame(hip_lib, esdf_mode):
    """20-frame stream, ESDF brought up to date after every frame: the incremental update (dirty bricks dilated by max_dist, re-initialised,
    relaxed from a device-side work queue) must give exactly the map of a full recompute, and exactly the oracle's Dijkstra."""
    from oracle import BATCHED
    from taichislam_amd.mapping import DenseTSDF
    K, frames = small_stream(20)
    inc, o = make_pair(SMALL, K)
    full = DenseTSDF(**SMALL); full.set_dep_camera_intrinsic(K)
    full.set_option("esdf_full", 1); full.set_option("esdf_mode", 0)
    md = 0.5                                       # 12.5 voxels: the influence of a change reaches one brick
    part = []
    for f, (R, T, d) in enumerate(frames):
        for m in (inc, full):
            m.recast_depth_to_map(R, T, d, None)
            m.update_esdf(max_dist=md)
        o.integrate_depth(R, T, d, mode=BATCHED)
        si, sf = inc.esdf_stats(), full.esdf_stats()
        assert sf["incremental"] == 0 and sf["region_bricks"] == sf["total_bricks"]
        assert si["incremental"] == (1 if f else 0) and 0 < si["dirty_bricks"] and si["region_bricks"] <= si["total_bricks"] == sf["total_bricks"]
        # (the wavefront's first, full update starts from the bricks that hold a band voxel and reaches only the bricks a value can be lowered in)
        assert si["changed_bricks"] <= si["region_bricks"] or (esdf_mode == 1 and f == 0)
        assert not f or 0 < si["changed_bricks"] <= si["dirty_bricks"]      # (a full update does not look at what changed)
        part.append(si["region_bricks"] / si["total_bricks"])
        (ii, ie), (fi, fe) = _esdf_sorted(inc), _esdf_sorted(full)
        assert np.array_equal(ii, fi) and np.array_equal(ie, fe), f"frame {f}: incremental != full at {(ie != fe).sum()} voxels"
        if f in (0, 7, 19):
            oi, oe = o.esdf(max_dist=md)
            oo = np.argsort(lin(oi))
            assert np.array_equal(ii, oi[oo]) and np.array_equal(ie, oe[oo]), f"frame {f}: != oracle"
    assert all(0 < x <= 1.0 for x in part)         # (a 16^3-brick map at 4 cm: one brick of dilation reaches everything; the 512^3 test below is partial)
    assert inc.get_option("esdf_orphans") == 0 and inc.get_option("esdf_mode") == esdf_mode
    if esdf_mode == 1:       # the wavefront writes the voxels whose value changes (and a few twice), not every voxel of a dilated region
        assert 0 < si["voxels_raised"] <= si["voxel_pushes"] < 0.8 * sf["voxel_pushes"]
    # nothing integrated since the last update: nothing to do
    assert inc.update_esdf(max_dist=md) == 0 and inc.esdf_stats()["dirty_bricks"] == 0
    # other parameters: everything again
    inc.update_esdf(max_dist=1.0)
    assert inc.esdf_stats()["incremental"] == 0


def test_incremental_update_at_benchmark_size(hip_lib, esdf_mode):
    """BASELINE configs[3] geometry (512^3 / 2 cm): incremental == full after 6 frames, and the update touches fewer bricks."""
    from taichislam_amd.mapping import DenseTSDF
    from taichislam_amd.utils import synthetic as syn
    from util import C2
    frames = list(syn.sphere_room_stream(6))
    inc, full = DenseTSDF(**C2), DenseTSDF(**C2)
    for m in (inc, full):
        m.set_dep_camera_intrinsic(syn.K_DEPTH)
    full.set_option("esdf_full", 1); full.set_option("esdf_mode", 0)
    for R, T, d in frames:
        for m in (inc, full):
            m.recast_depth_to_map(R, T, d, None)
            m.update_esdf(max_dist=0.4)
    (ii, ie), (fi, fe) = _esdf_sorted(inc), _esdf_sorted(full)
    assert ii.shape[0] > 1_000_000 and np.array_equal(ii, fi) and np.array_equal(ie, fe)
    si, sf = inc.esdf_stats(), full.esdf_stats()
    assert si["incremental"] == 1 and si["changed_bricks"] <= si["region_bricks"] <= sf["region_bricks"] == sf["total_bricks"]
    if esdf_mode == 0:
        assert si["brick_relaxations"] <= sf["brick_relaxations"] * 1.05          # (work counters depend on the order in which lanes meet: not exactly reproducible)
    else:                    # more, cheaper visits (a brick is looked at again when a neighbour moved its halo), far fewer voxel writes
        assert si["voxel_pushes"] < 0.7 * sf["voxel_pushes"] and inc.get_option("esdf_orphans") == 0


def test_asynchronous_updates_and_repair_of_a_short_update(hip_lib):
    """The per-frame hook form: update_esdf(wait=False) only enqueues (up to four updates in flight); what export_esdf / esdf_totals
    return is complete.  With the round cap knob an update stops while bricks are still listed: the next reader must get the repaired
    (fully recomputed) map, equal to a synchronous full recompute."""
    from taichislam_amd.mapping import DenseTSDF
    K, frames = small_stream(12)
    a = DenseTSDF(**SMALL); a.set_dep_camera_intrinsic(K)
    short = DenseTSDF(**SMALL); short.set_dep_camera_intrinsic(K)
    short.set_option("esdf_round_cap", 2)
    full = DenseTSDF(**SMALL); full.set_dep_camera_intrinsic(K)
    full.set_option("esdf_full", 1); full.set_option("esdf_mode", 0)
    md = 0.5
    for f, (R, T, d) in enumerate(frames):
        for m in (a, short):
            m.recast_depth_to_map(R, T, d, None)
            assert m.update_esdf(max_dist=md, wait=False) is None
        full.recast_depth_to_map(R, T, d, None)
        full.update_esdf(max_dist=md)
        if f in (5, 11):
            (ai, ae), (si, se), (fi, fe) = _esdf_sorted(a), _esdf_sorted(short), _esdf_sorted(full)
            assert np.array_equal(ai, fi) and np.array_equal(ae, fe), f"frame {f}: asynchronous != full at {(ae != fe).sum()} voxels"
            assert np.array_equal(si, fi) and np.array_equal(se, fe), f"frame {f}: repaired != full at {(se != fe).sum()} voxels"
    ta, ts = a.esdf_totals(), short.esdf_totals()
    assert ta["updates"] == len(frames) and ta["incremental"] == len(frames) - 1
    assert ts["updates"] > len(frames)             # the repairs are updates of their own


def test_updates_beside_the_next_frames_integration_equal_serial_full_recomputes(hip_lib, esdf_mode):
    """The per-frame hook at the benchmark's settings (512^3 / 2 cm, max_dist 1 m, updates only enqueued): with "esdf_overlap" the relaxation rounds
    of update n run on a phase-A stream beside the integration of frame n + 1.  After 8 frames the map must equal that of a handle that recomputes
    everything, waits for every update and keeps it on the handle's stream -- bit for bit."""
    from taichislam_amd.mapping import DenseTSDF
    from taichislam_amd.utils import synthetic as syn
    from util import C2
    frames = list(syn.sphere_room_stream(8))
    inc, ref = DenseTSDF(**C2), DenseTSDF(**C2)
    for m in (inc, ref):
        m.set_dep_camera_intrinsic(syn.K_DEPTH)
    ref.set_option("esdf_full", 1); ref.set_option("esdf_mode", 0); ref.set_option("esdf_overlap", 0)
    for R, T, d in frames:
        inc.recast_depth_to_map(R, T, d, None)
        assert inc.update_esdf(max_dist=1.0, wait=False) is None
        ref.recast_depth_to_map(R, T, d, None)
        ref.update_esdf(max_dist=1.0)
    (ii, ie), (fi, fe) = _esdf_sorted(inc), _esdf_sorted(ref)
    assert ii.shape[0] > 1_000_000 and np.array_equal(ii, fi) and np.array_equal(ie, fe), f"{(ie != fe).sum()} voxels differ"
    ti, tr = inc.esdf_totals(), ref.esdf_totals()
    assert ti["updates"] == tr["updates"] == 8 and ti["incremental"] == 7 and tr["incremental"] == 0
    # (at 1 m the dilated region is the whole map here and both start from the same band bricks: about the same work either way)
    assert ti["brick_relaxations"] <= tr["brick_relaxations"] * 1.05 if esdf_mode == 0 else ti["voxel_pushes"] < 0.75 * tr["voxel_pushes"]
    # and the TSDF itself is untouched by the overlap
    from util import assert_export_equal
    assert_export_equal(inc.export_submap(), ref.export_submap(), "TSDF beside overlapped ESDF updates")


def test_updates_follow_the_active_submap(hip_lib):
    """The ESDF belongs to the active submap: after switch_to_next_submap the next update recomputes everything for the new submap (enqueued-only
    updates of the old one still in flight beside the frames), and equals what a handle that only ever saw the new submap's frames computes."""
    from taichislam_amd.mapping import DenseTSDF
    K, frames = small_stream(6)
    a = DenseTSDF(**SMALL); b = DenseTSDF(**SMALL)
    for m in (a, b):
        m.set_dep_camera_intrinsic(K)
    eye, zero = np.eye(3), np.zeros(3)
    a.set_base_pose_submap(0, eye, zero)
    for R, T, d in frames[:3]:
        a.recast_depth_to_map(R, T, d, None)
        assert a.update_esdf(max_dist=0.5, wait=False) is None
    a.switch_to_next_submap()
    a.set_base_pose_submap(1, eye, zero)
    b.switch_to_next_submap()
    b.set_base_pose_submap(1, eye, zero)
    for R, T, d in frames[3:]:
        a.recast_depth_to_map(R, T, d, None)
        assert a.update_esdf(max_dist=0.5, wait=False) is None
        b.recast_depth_to_map(R, T, d, None)
        b.update_esdf(max_dist=0.5)
    (ai, ae), (bi, be) = _esdf_sorted(a), _esdf_sorted(b)
    assert ai.shape[0] > 10000 and np.array_equal(ai, bi) and np.array_equal(ae, be)
    ta = a.esdf_totals()
    assert ta["updates"] == 6 and ta["incremental"] == 4          # the first update of each submap is a full one


def test_async_update_followed_by_many_batches_on_every_phase_a_stream(hip_lib):
    """an asynchronous update takes a snapshot of the brick pool; phase A of LATER frames (it allocates bricks) must start behind it on
    every one of the three phase-A streams, not only on the first that issues a batch.  One update(wait=False), then 40 frames (five full batches:
    each stream issues at least one) without a sync, repeated; the ESDF at the end must equal the full recompute of the final map, and the slice
    export must be current without an explicit update (the reference updates inside every recast, dense_esdf.py:400-402)."""
    from taichislam_amd.mapping import DenseTSDF
    K, frames = small_stream(44, start_deg=10.0)
    inc, full = DenseTSDF(**SMALL), DenseTSDF(**SMALL)
    for m in (inc, full):
        m.set_dep_camera_intrinsic(K)
    full.set_option("esdf_full", 1); full.set_option("esdf_mode", 0)
    md = 0.5
    for f, (R, T, d) in enumerate(frames):
        inc.recast_depth_to_map(R, T, d, None)
        full.recast_depth_to_map(R, T, d, None)
        if f in (1, 3):
            inc.update_esdf(max_dist=md, wait=False)          # enqueued; the next frames pile up behind it
    z = 0.2
    pi, vi = inc.get_voxels_ESDF_slice(z)                      # brings the ESDF up to date first (the last update saw frame 3)
    full.update_esdf(max_dist=md)
    pf, vf = full.get_voxels_ESDF_slice(z)
    assert pi.shape[0] == pf.shape[0] > 500
    oi, of = np.lexsort(pi.T[::-1]), np.lexsort(pf.T[::-1])
    assert np.array_equal(pi[oi], pf[of]) and np.array_equal(vi[oi], vf[of])
    (ii, ie), (fi, fe) = _esdf_sorted(inc), _esdf_sorted(full)
    assert np.array_equal(ii, fi) and np.array_equal(ie, fe), f"incremental != full at {(ie != fe).sum()} voxels"


def test_a_surface_that_moves_by_many_voxels(hip_lib, esdf_mode):
    """Raise wave under load: the wall jumps from 3.0 m to 2.4 m and back while frames keep coming (the TSDF's weighted mean drags the zero crossing through
    fifteen voxels, band voxels appear and vanish, sides flip): values have to RISE by far more than an ulp.  After every frame the incremental update must
    equal the full recompute, and the oracle's Dijkstra at the end."""
    from oracle import BATCHED
    from taichislam_amd.mapping import DenseTSDF
    from taichislam_amd.utils import synthetic as syn
    h, w = 120, 160
    K = syn.scaled_intrinsics(h, w)
    inc, o = make_pair(SMALL, K)
    full = DenseTSDF(**SMALL); full.set_dep_camera_intrinsic(K)
    full.set_option("esdf_full", 1); full.set_option("esdf_mode", 0)
    md = 0.8
    radii = [3.0, 3.0, 2.4, 2.4, 2.4, 3.0, 2.4, 3.0, 3.0, 2.7, 2.2, 3.0]
    repaired = 0
    for f, rad in enumerate(radii):
        R, T = syn.camera_pose(2 * f)
        d = syn.sphere_room_depth(R, T, h, w, radius=rad, K=K)
        for m in (inc, full):
            m.recast_depth_to_map(R, T, d, None)
            m.update_esdf(max_dist=md)
        o.integrate_depth(R, T, d, mode=BATCHED)
        (ii, ie), (fi, fe) = _esdf_sorted(inc), _esdf_sorted(full)
        assert np.array_equal(ii, fi) and np.array_equal(ie, fe), f"frame {f} (radius {rad}): incremental != full at {(ie != fe).sum()} voxels"
        repaired += inc.esdf_stats()["incremental"] == 0 and f > 0
    oi, oe = o.esdf(max_dist=md)
    oo = np.argsort(lin(oi))
    assert np.array_equal(ii, oi[oo]) and np.array_equal(ie, oe[oo])
    assert inc.get_option("esdf_orphans") == 0
    assert repaired <= 4          # (an update that runs out of rounds is repaired by a full recompute: allowed, but it must not be the rule)
