"""GPU parity against the reference-LITERAL semantics, measured.

The HIP path is bit-exact with the oracle's order-free BATCHED restatement (tests/test_tsdf_parity_gpu.py).  The reference itself
updates TSDF / W per ray step in f16 with an unsynchronised read-modify-write (taichi_slam/mapping/dense_tsdf.py:264-267); its
deterministic reading is the oracle's FAITHFUL mode (sequential f16 replay).  This test integrates the BASELINE configs[1] stream
(640x480 -> 512^3 / 2 cm) for 12 frames with the HIP path, FAITHFUL and IDEAL (FAITHFUL's update sequence with the map state in
float64) and asserts MEASURED bounds -- thresholds are the values observed on MI355X with ~1.5-2x headroom; the full histogram is
written by tools/parity_report.py (profiles/r03_parity_vs_faithful.json).  The same yardstick is applied to the two
consumers of the map: marching cubes (mesh of the HIP map vs mesh of the FAITHFUL map) and submap fusion (HIP fuse_submaps vs the oracle's
sequential FAITHFUL fusion, dense_tsdf.py:272-307).  The GPU can also COMPUTE the literal semantics: tests/test_sequential_gpu.py.

Both yardsticks are restatements; FAITHFUL is pinned to the reference's own source run on tools/ti_seq (tests/test_ref_golden.py), not to Taichi."""
import numpy as np
import pytest

from taichislam_amd.utils import parity
from taichislam_amd.utils import synthetic as syn
from util import C2

pytestmark = pytest.mark.gpu
NFRAMES = 12


@pytest.fixture(scope="module")
def three_maps(hip_lib):
    from oracle import FAITHFUL, IDEAL, OracleTSDF
    from taichislam_amd.mapping import DenseTSDF
    frames = list(syn.sphere_room_stream(NFRAMES))
    g = DenseTSDF(**C2)
    g.set_dep_camera_intrinsic(syn.K_DEPTH)
    of, oi = OracleTSDF(**C2), OracleTSDF(**C2)
    of.set_intrinsics(syn.K_DEPTH); oi.set_intrinsics(syn.K_DEPTH)
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None)
        of.integrate_depth(R, T, d, mode=FAITHFUL)
        oi.integrate_depth(R, T, d, mode=IDEAL)
    rep = parity.deviation_report(g.export_submap(), of.export_sparse(), C2["voxel_scale"], sensor_xyz=frames[-1][1], ideal=oi.export_sparse())
    return rep


def test_sets_are_identical_to_the_literal_replay(three_maps):
    rep = three_maps
    assert rep["voxels_test"] == rep["voxels_ref"] > 1_400_000
    assert rep["index_sets_identical"] and rep["occupancy_identical"]


def test_tsdf_deviation_from_the_literal_replay_is_bounded(three_maps):
    t = three_maps["tsdf"]
    assert t["identical"] > 0.40 and t["within_1ulp"] > 0.75
    assert t["abs"]["p50"] <= 1e-4 and t["abs"]["p99"] <= 1.2e-2 and t["abs"]["p100"] <= 0.25
    band = three_maps["tsdf_surface_band"]                      # |TSDF| < 1.8 voxels: what meshing and planning read
    assert band["n"] > 50_000 and band["identical"] > 0.55
    assert band["abs"]["p99"] <= 1e-4 and band["abs"]["p100"] <= 4e-4
    # the north-star's "1e-4 relative" holds for about half of the voxels against an f16, order-dependent reference: stated, not hidden
    assert three_maps["tsdf_relative_floor_voxel"]["frac_le_1e4"] > 0.40


def test_deviation_shrinks_with_distance_from_the_sensor(three_maps):
    """Where thousands of rays meet in one voxel (next to the sensor) the reference's f16 running average drifts; at the surface it does not."""
    rows = three_maps["by_distance_from_sensor"]
    p99 = [r["test_vs_ref"]["abs"]["p99"] for r in rows]
    assert all(a >= b for a, b in zip(p99[:-1], p99[1:])), p99
    assert rows[-1]["test_vs_ref"]["abs"]["p100"] <= 6e-3


def test_hip_is_closer_to_the_float64_sequence_than_the_literal_replay_is(three_maps):
    rep = three_maps
    a, b = rep["tsdf_test_vs_ideal"], rep["tsdf_ref_vs_ideal"]
    assert a["mean_abs"] < b["mean_abs"] and a["abs"]["p99"] <= b["abs"]["p99"] and a["abs"]["p100"] <= b["abs"]["p100"]
    assert a["within_1ulp"] > 0.90 and a["within_1ulp"] > b["within_1ulp"]
    for row in rep["by_distance_from_sensor"]:
        assert row["test_vs_ideal"]["abs"]["p100"] <= row["ref_vs_ideal"]["abs"]["p100"], row
    wa, wb = rep["w_test_vs_ideal"], rep["w_ref_vs_ideal"]
    assert wa["within_1ulp"] > 0.97 and wa["ulps"]["p100"] <= 6 and wa["mean_abs"] < wb["mean_abs"]


def test_mesh_of_the_hip_map_is_the_mesh_of_the_literal_map_within_1e4_m(hip_lib):
    """marching_cube_mesher.py:44-187 on both maps after 12 frames: same triangle count, every vertex of either mesh has a vertex of the other
    within 3e-4 m (observed: p99 1.2e-5 m, max 9.7e-5 m -- the surface band is where the two maps agree best)."""
    from oracle import FAITHFUL, OracleTSDF
    from taichislam_amd.mapping import DenseTSDF, MarchingCubeMesher
    g = DenseTSDF(**C2); g.set_dep_camera_intrinsic(syn.K_DEPTH)
    of = OracleTSDF(**C2); of.set_intrinsics(syn.K_DEPTH)
    for R, T, d in syn.sphere_room_stream(NFRAMES):
        g.recast_depth_to_map(R, T, d, None)
        of.integrate_depth(R, T, d, mode=FAITHFUL)
    mesher = MarchingCubeMesher(g, max_triangles=2_000_000, tsdf_surface_thres=5 * C2["voxel_scale"])
    mesher.generate_mesh(1)
    rep = parity.mesh_deviation(mesher.get_mesh()[0], of.generate_mesh(1, 5 * C2["voxel_scale"], 2_000_000)[0], C2["voxel_scale"])
    assert rep["triangles_test"] == rep["triangles_ref"] > 50_000
    for side in ("test_to_ref_m", "ref_to_test_m"):
        assert rep[side]["p99"] <= 3e-5 and rep[side]["p100"] <= 3e-4, rep
    assert rep["frac_within_1_percent_of_a_voxel"] == 1.0


def test_fusion_deviation_from_the_sequential_fusion_is_bounded(hip_lib):
    """fuse_submaps: the HIP path forms sum(w t) / sum(w) once per global voxel from exact sums; the reference (and the oracle's FAITHFUL mode)
    applies the contributions one by one in f16.  Three submaps of three frames; measured bounds with ~2x headroom (profiles/r03_parity_vs_faithful.json)."""
    from oracle import BATCHED, FAITHFUL, OracleTSDF
    from taichislam_amd.mapping import DenseTSDF
    NS, PER = 3, 3
    sub_g = DenseTSDF(**C2, max_submap_num=8); sub_g.set_dep_camera_intrinsic(syn.K_DEPTH)
    sub_b, sub_f = OracleTSDF(**C2, max_submap_num=8), OracleTSDF(**C2, max_submap_num=8)
    glob_g = DenseTSDF(**dict(C2, is_global_map=True), max_submap_num=8)
    gb, gf = (OracleTSDF(**dict(C2, is_global_map=True), max_submap_num=8) for _ in range(2))
    for o in (sub_b, sub_f):
        o.set_intrinsics(syn.K_DEPTH)
    for s in range(NS):
        fr = list(syn.sphere_room_stream(PER, start_deg=30.0 * s))
        sub_g.active_submap_id[None] = s
        for m in (sub_g, glob_g):
            m.set_base_pose_submap(s, fr[0][0], fr[0][1])
        for o in (sub_b, sub_f, gb, gf):
            o.set_base_pose_submap(s, fr[0][0], fr[0][1])
        for o in (sub_b, sub_f):
            o.set_active_submap(s)
        for R, T, d in fr:
            sub_g.recast_depth_to_map(R, T, d, None)
            sub_b.integrate_depth(R, T, d, mode=BATCHED)
            sub_f.integrate_depth(R, T, d, mode=FAITHFUL)
    sub_g.active_submap_id[None] = NS
    for o in (sub_b, sub_f):
        o.set_active_submap(NS)
    glob_g.fuse_submaps(sub_g)
    gb.fuse_submaps(sub_b, mode=FAITHFUL)        # the HIP submaps' bits, fused sequentially: the fusion's own semantics
    gf.fuse_submaps(sub_f, mode=FAITHFUL)        # submaps and fusion both literal
    eg = glob_g.export_submap()
    same = parity.deviation_report(eg, gb.export_sparse(), C2["voxel_scale"])
    lit = parity.deviation_report(eg, gf.export_sparse(), C2["voxel_scale"])
    for rep in (same, lit):
        assert rep["index_sets_identical"] and rep["voxels_ref"] > 1_500_000
    assert same["tsdf"]["within_1ulp"] > 0.93 and same["tsdf"]["abs"]["p99"] <= 5e-3
    assert lit["tsdf"]["within_1ulp"] > 0.85 and lit["tsdf"]["abs"]["p99"] <= 8e-3
    assert same["tsdf_surface_band"]["abs"]["p99"] <= 2e-3 and lit["tsdf_surface_band"]["abs"]["p99"] <= 2e-3
