"""Shared helpers for the parity tests (HIP path vs CPU oracle on identical inputs)."""
import numpy as np

from taichislam_amd.utils import synthetic as syn

C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0,
          min_ray_length=0.3, internal_voxels=10, recast_step=2, texture_enabled=False)
SMALL = dict(map_scale=[10.24, 10.24], voxel_scale=0.04, num_voxel_per_blk_axis=16, max_ray_length=5.0,
             min_ray_length=0.3, internal_voxels=10, recast_step=2, texture_enabled=False)


def lin(idx):
    i = idx.astype(np.int64)
    return ((i[:, 0] + 32768) << 32) | ((i[:, 1] + 32768) << 16) | (i[:, 2] + 32768)


def sort_export(e):
    o = np.argsort(lin(e["indices"]), kind="stable")
    out = {"indices": e["indices"][o], "TSDF": np.asarray(e["TSDF"])[o].view(np.uint16),
           "W_TSDF": np.asarray(e["W_TSDF"])[o].view(np.uint16), "occupy": e["occupy"][o]}
    if getattr(e.get("color", None), "size", 0):
        out["color"] = np.asarray(e["color"])[o].view(np.uint16)
    return out


def assert_export_equal(a, b, what=""):
    a, b = sort_export(a), sort_export(b)
    assert a["indices"].shape == b["indices"].shape, f"{what}: voxel count {a['indices'].shape[0]} != {b['indices'].shape[0]}"
    assert np.array_equal(a["indices"], b["indices"]), f"{what}: voxel index sets differ"
    for k in ("TSDF", "W_TSDF", "occupy"):
        bad = np.nonzero(a[k] != b[k])[0]
        assert bad.size == 0, f"{what}: {k} differs at {bad.size} voxels, first {a['indices'][bad[0]]}: {a[k][bad[0]]} vs {b[k][bad[0]]}"
    if "color" in a or "color" in b:
        assert np.array_equal(a["color"], b["color"]), f"{what}: color differs"


def sorted_rows(x, decimals=None):
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None]
    o = np.lexsort(x.T[::-1])
    return x[o]


def small_stream(n, h=120, w=160, **kw):
    K = syn.scaled_intrinsics(h, w)
    frames = []
    for f in range(n):
        R, T = syn.camera_pose(f, **{k: v for k, v in kw.items() if k in ("orbit", "start_deg")})
        frames.append((R, T, syn.sphere_room_depth(R, T, h, w, radius=kw.get("radius", 3.0), K=K)))
    return K, frames


def make_pair(cfg, K, **gpu_kw):
    """(HIP DenseTSDF, OracleTSDF) with identical configuration."""
    from oracle import OracleTSDF
    from taichislam_amd.mapping import DenseTSDF
    g = DenseTSDF(**cfg, **gpu_kw)
    g.set_dep_camera_intrinsic(K)
    g.set_color_camera_intrinsic(K)
    ocfg = {k: v for k, v in cfg.items() if k not in ("max_disp_particles",)}
    o = OracleTSDF(**ocfg)
    o.set_intrinsics(K, K)
    return g, o
