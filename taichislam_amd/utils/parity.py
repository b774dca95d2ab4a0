"""Deviation statistics between two sparse map exports (numpy only; no oracle, no GPU).

Used by tests/test_parity_vs_faithful_gpu.py, tools/parity_report.py and bench.py to state, in numbers, how far the HIP path (bit-exact
with the order-free BATCHED restatement) is from the reference-literal sequential f16 replay (FAITHFUL) and from the same update
sequence carried in float64 (IDEAL) -- see DESIGN.md section 2."""
import numpy as np


def _lin(idx):
    i = idx.astype(np.int64)
    return ((i[:, 0] + 32768) << 32) | ((i[:, 1] + 32768) << 16) | (i[:, 2] + 32768)


def _sorted(e):
    o = np.argsort(_lin(e["indices"]), kind="stable")
    return {"indices": e["indices"][o], "TSDF": np.asarray(e["TSDF"])[o].view(np.float16), "W_TSDF": np.asarray(e["W_TSDF"])[o].view(np.float16),
            "occupy": e["occupy"][o]}


def f16_ulp(x):
    """Spacing of float16 at |x| (subnormal spacing below 2^-14)."""
    ax = np.maximum(np.abs(np.asarray(x, dtype=np.float64)), 2.0 ** -14)
    return 2.0 ** (np.floor(np.log2(ax)) - 10)


def _pct(d, qs=(50, 90, 99, 99.9, 100)):
    if d.size == 0:
        return {f"p{q:g}": None for q in qs}
    return {f"p{q:g}": float(v) for q, v in zip(qs, np.percentile(d, qs))}


def field_deviation(a, ref):
    """a, ref: float16 arrays of one field over the same voxels.  NaNs (a fusion's 0 / 0, dense_tsdf.py:275) are counted on each side and
    left out of the distance statistics; `identical` compares bits over everything."""
    a64, r64 = a.astype(np.float64), ref.astype(np.float64)
    fin = np.isfinite(a64) & np.isfinite(r64)
    out = {"n": int(a64.size), "identical": float(np.mean(a.view(np.uint16) == ref.view(np.uint16))) if a64.size else None}
    if not fin.all():
        out["nan_test"], out["nan_ref"], out["nan_both"] = int(np.isnan(a64).sum()), int(np.isnan(r64).sum()), int((np.isnan(a64) & np.isnan(r64)).sum())
    d = np.abs(a64[fin] - r64[fin])
    u = d / f16_ulp(r64[fin])
    out.update({"within_1ulp": float(np.mean(u <= 1.0)) if d.size else None, "abs": _pct(d), "ulps": _pct(u), "mean_abs": float(d.mean()) if d.size else None})
    return out


def deviation_report(test, ref, voxel_scale, sensor_xyz=None, ideal=None, dist_bins=(0.0, 0.3, 1.0, 2.0, 3.0, 1e9)):
    """Compare export `test` with export `ref` (dicts with indices/TSDF/W_TSDF/occupy).  Returns a JSON-able dict; index and
    occupancy set equality are reported, not asserted.  `ideal`: optional third export (float64 yardstick) -- per distance bin the
    report then also says how far each of the two is from it."""
    t, r = _sorted(test), _sorted(ref)
    same_idx = t["indices"].shape == r["indices"].shape and bool(np.array_equal(t["indices"], r["indices"]))
    out = {"voxels_test": int(t["indices"].shape[0]), "voxels_ref": int(r["indices"].shape[0]), "index_sets_identical": same_idx}
    if not same_idx:
        return out
    out["occupancy_identical"] = bool(np.array_equal(t["occupy"], r["occupy"]))
    vs = float(voxel_scale)
    out["tsdf"] = field_deviation(t["TSDF"], r["TSDF"])
    out["w"] = field_deviation(t["W_TSDF"], r["W_TSDF"])
    rt = r["TSDF"].astype(np.float64)
    with np.errstate(invalid="ignore"):
        band = np.abs(rt) < 1.8 * vs                                        # the reference's surface threshold (dense_tsdf.py:39)
    out["tsdf_surface_band"] = field_deviation(t["TSDF"][band], r["TSDF"][band])
    rel = np.abs(t["TSDF"].astype(np.float64) - rt) / np.maximum(np.abs(rt), vs)
    rel = rel[np.isfinite(rel)]
    out["tsdf_relative_floor_voxel"] = dict(_pct(rel), frac_le_1e4=float(np.mean(rel <= 1e-4)) if rel.size else None)
    i = None
    if ideal is not None:
        i = _sorted(ideal)
        if not np.array_equal(i["indices"], r["indices"]):
            i = None
    if i is not None:
        out["tsdf_test_vs_ideal"] = field_deviation(t["TSDF"], i["TSDF"])
        out["tsdf_ref_vs_ideal"] = field_deviation(r["TSDF"], i["TSDF"])
        out["w_test_vs_ideal"] = field_deviation(t["W_TSDF"], i["W_TSDF"])
        out["w_ref_vs_ideal"] = field_deviation(r["W_TSDF"], i["W_TSDF"])
    if sensor_xyz is not None:
        p = r["indices"].astype(np.float64) * vs
        dist = np.linalg.norm(p - np.asarray(sensor_xyz, dtype=np.float64).reshape(1, 3), axis=1)
        rows = []
        for lo, hi in zip(dist_bins[:-1], dist_bins[1:]):
            s = (dist >= lo) & (dist < hi)
            if not s.any():
                continue
            row = {"from_m": float(lo), "to_m": float(min(hi, 1e9)), "n": int(s.sum()), "test_vs_ref": field_deviation(t["TSDF"][s], r["TSDF"][s])}
            if i is not None:
                row["test_vs_ideal"] = field_deviation(t["TSDF"][s], i["TSDF"][s])
                row["ref_vs_ideal"] = field_deviation(r["TSDF"][s], i["TSDF"][s])
            rows.append(row)
        out["by_distance_from_sensor"] = rows
    return out


def mesh_deviation(v_test, v_ref, voxel_scale):
    """Two triangle soups (vertices [3n, 3] in metres): nearest-vertex distances in both directions (scipy k-d tree).  The meshes come from
    maps that differ in the last f16 bits, so their triangulations differ where a cube's case flips; distance between the vertex sets is the
    geometric statement (marching_cube_mesher.py:44-60 interpolates vertices from TSDF values)."""
    from scipy.spatial import cKDTree
    a, b = np.asarray(v_test, dtype=np.float64).reshape(-1, 3), np.asarray(v_ref, dtype=np.float64).reshape(-1, 3)
    out = {"triangles_test": int(a.shape[0] // 3), "triangles_ref": int(b.shape[0] // 3)}
    # vertexInterp divides by the difference of two corner values (marching_cube_mesher.py:44-60): equal corners give non-finite vertices,
    # in the reference as in both implementations -- counted, not compared
    fa, fb = np.isfinite(a).all(1), np.isfinite(b).all(1)
    out["nonfinite_vertices_test"], out["nonfinite_vertices_ref"] = int((~fa).sum()), int((~fb).sum())
    a, b = a[fa], b[fb]
    if a.size == 0 or b.size == 0:
        return out
    d_ab = cKDTree(b).query(a)[0]
    d_ba = cKDTree(a).query(b)[0]
    vs = float(voxel_scale)
    out["test_to_ref_m"] = _pct(d_ab); out["ref_to_test_m"] = _pct(d_ba)
    out["frac_within_1e-4_of_a_voxel"] = float(np.mean(np.concatenate([d_ab, d_ba]) <= 1e-4 * vs))
    out["frac_within_1_percent_of_a_voxel"] = float(np.mean(np.concatenate([d_ab, d_ba]) <= 1e-2 * vs))
    out["identical_vertex_fraction"] = float(np.mean(np.concatenate([d_ab, d_ba]) == 0.0))
    return out


def short_summary(rep):
    """The handful of numbers quoted in BASELINE.md / the bench line."""
    if not rep.get("index_sets_identical"):
        return {"index_sets_identical": False}
    s = {"voxels": rep["voxels_ref"], "index_sets_identical": True, "occupancy_identical": rep["occupancy_identical"],
         "tsdf_bits_identical": rep["tsdf"]["identical"], "tsdf_within_1_f16_ulp": rep["tsdf"]["within_1ulp"],
         "tsdf_abs_m": rep["tsdf"]["abs"], "tsdf_surface_band_abs_m": rep["tsdf_surface_band"]["abs"],
         "tsdf_rel_frac_le_1e-4": rep["tsdf_relative_floor_voxel"]["frac_le_1e4"], "w_within_1_f16_ulp": rep["w"]["within_1ulp"]}
    if "tsdf_test_vs_ideal" in rep:
        s["mean_abs_m_vs_float64_sequence"] = {"hip": rep["tsdf_test_vs_ideal"]["mean_abs"], "faithful": rep["tsdf_ref_vs_ideal"]["mean_abs"]}
        s["max_abs_m_vs_float64_sequence"] = {"hip": rep["tsdf_test_vs_ideal"]["abs"]["p100"], "faithful": rep["tsdf_ref_vs_ideal"]["abs"]["p100"]}
    return s
