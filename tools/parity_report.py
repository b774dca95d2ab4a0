"""HIP path vs the reference-literal semantics (oracle FAITHFUL) at BASELINE configs[1], on the GPU box:
   integration: deviation histogram after n frames, and its growth with the length of the stream;
   fusion (dense_tsdf.py:272-307): HIP fuse_submaps vs the oracle's sequential FAITHFUL fusion -- of the same submaps (the fusion's own
           semantics) and of FAITHFUL submaps (the whole pipeline literal);
   marching cubes (marching_cube_mesher.py:44-187): mesh of the HIP map vs mesh of the FAITHFUL map, vertex distances.
python tools/parity_report.py [frames] [out.json]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import BATCHED, FAITHFUL, IDEAL, OracleTSDF                      # checker only
from taichislam_amd.mapping import DenseTSDF, MarchingCubeMesher
from taichislam_amd.utils import parity
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "parity_vs_faithful.json")
GROWTH = [int(x) for x in os.environ.get("PARITY_GROWTH", "12,25,50,77").split(",")]
nmax = max([n] + GROWTH)
frames = list(syn.sphere_room_stream(nmax))
g = DenseTSDF(**C2); g.set_dep_camera_intrinsic(syn.K_DEPTH)
of, oi = OracleTSDF(**C2), OracleTSDF(**C2)
of.set_intrinsics(syn.K_DEPTH); oi.set_intrinsics(syn.K_DEPTH)
rep, growth = None, []
for f, (R, T, d) in enumerate(frames):
    g.recast_depth_to_map(R, T, d, None)
    of.integrate_depth(R, T, d, mode=FAITHFUL)
    if f < n:
        oi.integrate_depth(R, T, d, mode=IDEAL)
    if f + 1 == n:
        eg, ef = g.export_submap(), of.export_sparse()
        rep = parity.deviation_report(eg, ef, C2["voxel_scale"], sensor_xyz=T, ideal=oi.export_sparse())
        # marching cubes on the two maps (same mesher settings as scripts/taichislam_node.py:209)
        mesher = MarchingCubeMesher(g, max_triangles=4_000_000, tsdf_surface_thres=5 * C2["voxel_scale"])
        mesher.generate_mesh(1)
        vg = mesher.get_mesh()[0]
        vf = of.generate_mesh(1, 5 * C2["voxel_scale"], 4_000_000)[0]
        rep["mesh_vs_faithful_map"] = parity.mesh_deviation(vg, vf, C2["voxel_scale"])
    if f + 1 in GROWTH:
        r = parity.deviation_report(g.export_submap(), of.export_sparse(), C2["voxel_scale"])
        growth.append({"frames": f + 1, "voxels": r["voxels_ref"], "index_sets_identical": r["index_sets_identical"], "tsdf_bits_identical": r["tsdf"]["identical"],
                       "tsdf_within_1_f16_ulp": r["tsdf"]["within_1ulp"], "tsdf_rel_frac_le_1e-4": r["tsdf_relative_floor_voxel"]["frac_le_1e4"],
                       "tsdf_abs_m": r["tsdf"]["abs"], "surface_band_abs_m": r["tsdf_surface_band"]["abs"], "surface_band_within_1_f16_ulp": r["tsdf_surface_band"]["within_1ulp"]})
rep["growth_with_stream_length"] = growth
rep["config"] = f"BASELINE configs[1], first {n} frames of the synthetic sphere-room stream; test = HIP path, ref = oracle FAITHFUL, ideal = oracle IDEAL"

# ---- fusion: three submaps of four frames each (base pose = first camera pose of the submap), fused into a 512^3 global map ----
NS, PER = 3, 4
sub_g = DenseTSDF(**C2, max_submap_num=8); sub_g.set_dep_camera_intrinsic(syn.K_DEPTH)
sub_b, sub_f = OracleTSDF(**C2, max_submap_num=8), OracleTSDF(**C2, max_submap_num=8)
for o in (sub_b, sub_f):
    o.set_intrinsics(syn.K_DEPTH)
glob_g = DenseTSDF(**dict(C2, is_global_map=True), max_submap_num=8)
glob = {k: OracleTSDF(**dict(C2, is_global_map=True), max_submap_num=8) for k in ("ff", "bf")}
for s in range(NS):
    fr = list(syn.sphere_room_stream(PER, start_deg=30.0 * s))
    sub_g.active_submap_id[None] = s
    sub_g.set_base_pose_submap(s, fr[0][0], fr[0][1]); glob_g.set_base_pose_submap(s, fr[0][0], fr[0][1])
    for o in (sub_b, sub_f):
        o.set_active_submap(s); o.set_base_pose_submap(s, fr[0][0], fr[0][1])
    for o in glob.values():
        o.set_base_pose_submap(s, fr[0][0], fr[0][1])
    for R, T, d in fr:
        sub_g.recast_depth_to_map(R, T, d, None)
        sub_b.integrate_depth(R, T, d, mode=BATCHED)
        sub_f.integrate_depth(R, T, d, mode=FAITHFUL)
sub_g.active_submap_id[None] = NS
for o in (sub_b, sub_f):
    o.set_active_submap(NS)
glob_g.fuse_submaps(sub_g)
glob["bf"].fuse_submaps(sub_b, mode=FAITHFUL)      # the HIP submaps' bits (== BATCHED), fused sequentially as the reference does
glob["ff"].fuse_submaps(sub_f, mode=FAITHFUL)      # everything literal
eg = glob_g.export_submap()
ok = ~np.isnan(np.asarray(eg["TSDF"]).astype(np.float32))


def fusion_rep(ref):
    r = parity.deviation_report(eg, ref, C2["voxel_scale"])
    keep = ["voxels_test", "voxels_ref", "index_sets_identical", "occupancy_identical", "tsdf", "w", "tsdf_surface_band", "tsdf_relative_floor_voxel"]
    return {k: r[k] for k in keep if k in r}


rep["fusion"] = {"config": f"{NS} submaps x {PER} frames (30 degrees apart) fused into a 512^3 global map; test = HIP fuse_submaps of HIP submaps",
                 "vs_sequential_fusion_of_the_same_submaps": fusion_rep(glob["bf"].export_sparse()),
                 "vs_sequential_fusion_of_sequential_submaps": fusion_rep(glob["ff"].export_sparse())}
mg = MarchingCubeMesher(glob_g, max_triangles=4_000_000, tsdf_surface_thres=5 * C2["voxel_scale"]); mg.generate_mesh(1)
rep["fusion"]["mesh_of_the_global_map_vs_all_sequential"] = parity.mesh_deviation(mg.get_mesh()[0], glob["ff"].generate_mesh(1, 5 * C2["voxel_scale"], 4_000_000)[0], C2["voxel_scale"])
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(rep, open(out, "w"), indent=1)
s = parity.short_summary(rep)
s["mesh"] = rep["mesh_vs_faithful_map"]
s["growth"] = [{k: g_[k] for k in ("frames", "tsdf_rel_frac_le_1e-4", "tsdf_within_1_f16_ulp", "surface_band_within_1_f16_ulp")} | {"p99_m": g_["tsdf_abs_m"]["p99"], "max_m": g_["tsdf_abs_m"]["p100"], "band_p99_m": g_["surface_band_abs_m"]["p99"]} for g_ in growth]
fu = rep["fusion"]
s["fusion"] = {k: {"index_sets_identical": v.get("index_sets_identical"), "tsdf_within_1_f16_ulp": v.get("tsdf", {}).get("within_1ulp"), "tsdf_abs_m": v.get("tsdf", {}).get("abs"),
                   "rel_le_1e-4": v.get("tsdf_relative_floor_voxel", {}).get("frac_le_1e4")} for k, v in fu.items() if k.startswith("vs_")}
s["fusion"]["mesh"] = fu["mesh_of_the_global_map_vs_all_sequential"]
print(json.dumps(s))
