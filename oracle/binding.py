"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference; test infrastructure)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FAITHFUL = 0
BATCHED = 1
IDEAL = 2        # FAITHFUL sequence with the map state in float64 (measurement aid, see tsl_oracle.h)


class TsdfCfg(C.Structure):
    _fields_ = [("map_size_xy", C.c_double), ("map_size_z", C.c_double), ("voxel_scale", C.c_double),
                ("num_voxel_per_blk_axis", C.c_int), ("max_ray_length", C.c_double), ("min_ray_length", C.c_double),
                ("internal_voxels", C.c_int), ("max_submap_num", C.c_int), ("is_global_map", C.c_int),
                ("texture_enabled", C.c_int), ("disp_ceiling", C.c_double), ("disp_floor", C.c_double),
                ("recast_step", C.c_int), ("color_same_proj", C.c_int)]


class OctoCfg(C.Structure):
    _fields_ = [("map_size_xy", C.c_double), ("map_size_z", C.c_double), ("voxel_scale", C.c_double),
                ("min_occupy_thres", C.c_double), ("texture_enabled", C.c_int),
                ("min_ray_length", C.c_double), ("max_ray_length", C.c_double), ("K", C.c_int),
                ("max_submap_num", C.c_int), ("disp_ceiling", C.c_double), ("disp_floor", C.c_double),
                ("is_global_map", C.c_int), ("recast_step", C.c_int), ("color_same_proj", C.c_int)]


class FrameStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("p_used", "p_valid", "p_oob", "v_pcl", "v_skipped", "steps",
                                          "steps_oob", "unique", "bricks")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("tsl_oracle.c", "tsl_oracle_octo.c", "tsl_oracle.h", "mc_tables.h",
                                              "mc_tables_data.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i64, dp, fp = C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_float)
        L.ora_tsdf_create.restype = vp
        L.ora_tsdf_create.argtypes = [C.POINTER(TsdfCfg)]
        L.ora_tsdf_destroy.argtypes = [vp]
        L.ora_tsdf_get_dims.argtypes = [vp] + [C.POINTER(C.c_int)] * 4
        L.ora_tsdf_set_intrinsics.argtypes = [vp, dp, dp]
        L.ora_tsdf_set_base_pose.argtypes = [vp, dp, dp]
        L.ora_tsdf_set_base_pose_submap.argtypes = [vp, C.c_int, dp, dp]
        L.ora_tsdf_get_active_submap.argtypes = [vp]
        L.ora_tsdf_set_active_submap.argtypes = [vp, C.c_int]
        L.ora_tsdf_reset.argtypes = [vp]
        L.ora_tsdf_integrate_depth.argtypes = [vp, C.c_int, dp, dp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int,
                                               C.POINTER(FrameStats)]
        L.ora_tsdf_integrate_points.argtypes = [vp, C.c_int, dp, dp, vp, vp, i64, C.POINTER(FrameStats)]
        L.ora_tsdf_set_schedule.argtypes = [vp, C.c_int, C.c_int, C.c_uint64]
        L.ora_tsdf_integrate_depth_mt.argtypes = [vp, dp, dp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(FrameStats)]
        L.ora_tsdf_count_active.restype = i64
        L.ora_tsdf_count_active.argtypes = [vp]
        L.ora_tsdf_export_sparse.restype = i64
        L.ora_tsdf_export_sparse.argtypes = [vp, vp, vp, vp, vp, vp, i64]
        L.ora_tsdf_export_occupied.restype = i64
        L.ora_tsdf_export_occupied.argtypes = [vp, vp, vp, i64]
        L.ora_tsdf_import_sparse.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, i64]
        L.ora_tsdf_surface_voxels.restype = i64
        L.ora_tsdf_surface_voxels.argtypes = [vp, vp, vp, i64]
        L.ora_tsdf_slice_voxels.restype = i64
        L.ora_tsdf_slice_voxels.argtypes = [vp, C.c_float, C.c_float, vp, vp, vp, i64]
        L.ora_tsdf_fuse_submaps.argtypes = [vp, vp, C.c_int]
        L.ora_tsdf_fuse_accumulate_dense.argtypes = [vp, vp, vp, vp]
        L.ora_tsdf_fuse_finalize_dense.argtypes = [vp, vp, vp]
        L.ora_tsdf_query_points.argtypes = [vp, C.c_int, C.c_int, vp, i64, vp]
        L.ora_tsdf_query_raycast.argtypes = [vp, vp, vp, C.c_float, i64, vp, vp, vp]
        L.ora_mesh_generate.restype = i64
        L.ora_mesh_generate.argtypes = [vp, C.c_int, C.c_float, i64, vp, vp, vp]
        L.ora_esdf_compute.restype = i64
        L.ora_esdf_compute.argtypes = [vp, C.c_float, C.c_float, vp, vp, i64]
        L.ora_f32_to_f16.restype = C.c_uint16
        L.ora_f32_to_f16.argtypes = [C.c_float]
        L.ora_f16_to_f32.restype = C.c_float
        L.ora_f16_to_f32.argtypes = [C.c_uint16]
        L.ora_octo_create.restype = vp
        L.ora_octo_create.argtypes = [C.POINTER(OctoCfg)]
        L.ora_octo_destroy.argtypes = [vp]
        L.ora_octo_get_dims.argtypes = [vp] + [C.POINTER(C.c_int)] * 4 + [dp]
        L.ora_octo_set_intrinsics.argtypes = [vp, dp, dp]
        L.ora_octo_set_base_pose_submap.argtypes = [vp, C.c_int, dp, dp]
        L.ora_octo_set_active_submap.argtypes = [vp, C.c_int]
        L.ora_octo_reset.argtypes = [vp]
        L.ora_octo_integrate_depth.argtypes = [vp, dp, dp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int,
                                               C.POINTER(FrameStats)]
        L.ora_octo_integrate_points.argtypes = [vp, dp, dp, vp, vp, i64, C.POINTER(FrameStats)]
        L.ora_octo_export_leaves.restype = i64
        L.ora_octo_export_leaves.argtypes = [vp, vp, vp, vp, i64]
        L.ora_octo_occupied_voxels.restype = i64
        L.ora_octo_occupied_voxels.argtypes = [vp, C.c_int, vp, vp, i64]
        L.ora_octo_fuse_submaps.argtypes = [vp, vp]
        _LIB = L
    return _LIB


def _d(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    assert a.size == n
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleTSDF:
    """CPU oracle of DenseTSDF (reference: taichi_slam/mapping/dense_tsdf.py)."""

    def __init__(self, map_scale=(10, 10), voxel_scale=0.05, texture_enabled=False, num_voxel_per_blk_axis=16,
                 max_ray_length=10, min_ray_length=0.3, internal_voxels=10, max_submap_num=1024,
                 is_global_map=False, disp_ceiling=1.8, disp_floor=-0.3, recast_step=2, color_same_proj=True):
        self.cfg = TsdfCfg(float(map_scale[0]), float(map_scale[1]), float(voxel_scale), int(num_voxel_per_blk_axis),
                           float(max_ray_length), float(min_ray_length), int(internal_voxels), int(max_submap_num),
                           int(is_global_map), int(texture_enabled), float(disp_ceiling), float(disp_floor),
                           int(recast_step), int(color_same_proj))
        self.L = lib()
        self.h = self.L.ora_tsdf_create(C.byref(self.cfg))
        self.texture = bool(texture_enabled)
        d = [C.c_int() for _ in range(4)]
        self.L.ora_tsdf_get_dims(self.h, *[C.byref(x) for x in d])
        self.N, self.Nz, self.pcl_lo, self.pcl_hi = [x.value for x in d]
        self.voxel_scale = voxel_scale

    def __del__(self):
        try:
            self.L.ora_tsdf_destroy(self.h)
        except Exception:
            pass

    def set_intrinsics(self, Kdep, Kcol=None):
        _, kd = _d(Kdep, 9)
        _, kc = _d(Kcol if Kcol is not None else Kdep, 9)
        self.L.ora_tsdf_set_intrinsics(self.h, kd, kc)

    def set_base_pose_submap(self, sid, R, T):
        _, r = _d(R, 9)
        _, t = _d(T, 3)
        self.L.ora_tsdf_set_base_pose_submap(self.h, sid, r, t)

    def set_schedule(self, kind=0, param=0, seed=0):
        """Which legal serialisation of the reference's racy ray loop FAITHFUL / IDEAL replay: 0 struct-for order (default), 1 random ray
        order (seed), 2 `param` threads over contiguous shares of the struct-for order, one step per turn."""
        self.L.ora_tsdf_set_schedule(self.h, int(kind), int(param), int(seed))

    def set_active_submap(self, sid):
        self.L.ora_tsdf_set_active_submap(self.h, sid)

    def get_active_submap(self):
        return self.L.ora_tsdf_get_active_submap(self.h)

    def reset(self):
        self.L.ora_tsdf_reset(self.h)

    def integrate_depth(self, R, T, depth, texture=None, mode=BATCHED):
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        _, r = _d(R, 9)
        _, t = _d(T, 3)
        st = FrameStats()
        th = tw = 0
        if texture is not None and getattr(texture, "size", 0):
            texture = np.ascontiguousarray(texture, dtype=np.uint8)
            th, tw = texture.shape[:2]
        else:
            texture = None
        self.L.ora_tsdf_integrate_depth(self.h, mode, r, t, _p(depth), depth.shape[0], depth.shape[1],
                                        _p(texture), th, tw, C.byref(st))
        return st.as_dict()

    def integrate_depth_mt(self, R, T, depth, nthreads):
        """All-core port of the BATCHED semantics (tsl_oracle.c: brick-binned, OpenMP); bit-identical to integrate_depth(mode=BATCHED)."""
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        _, r = _d(R, 9)
        _, t = _d(T, 3)
        st = FrameStats()
        rc = self.L.ora_tsdf_integrate_depth_mt(self.h, r, t, _p(depth), depth.shape[0], depth.shape[1], int(nthreads), C.byref(st))
        assert rc == 0
        return st.as_dict()

    def integrate_points(self, R, T, xyz, rgb=None, mode=BATCHED):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        _, r = _d(R, 9)
        _, t = _d(T, 3)
        if rgb is not None and getattr(rgb, "size", 0):
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        else:
            rgb = None
        st = FrameStats()
        self.L.ora_tsdf_integrate_points(self.h, mode, r, t, _p(xyz), _p(rgb), xyz.shape[0], C.byref(st))
        return st.as_dict()

    def count_active(self):
        return int(self.L.ora_tsdf_count_active(self.h))

    def export_sparse(self):
        n = self.count_active()
        idx = np.zeros((n, 3), np.int16)
        t = np.zeros(n, np.uint16)
        w = np.zeros(n, np.uint16)
        occ = np.zeros(n, np.int8)
        col = np.zeros((n, 3), np.uint16) if self.texture else None
        self.L.ora_tsdf_export_sparse(self.h, _p(idx), _p(t), _p(w), _p(occ), _p(col), n)
        return {"indices": idx, "TSDF": t.view(np.float16), "W_TSDF": w.view(np.float16), "occupy": occ,
                "color": col.view(np.float16) if col is not None else np.array([])}

    def export_occupied(self):
        n = int(self.L.ora_tsdf_export_occupied(self.h, None, None, 0))
        idx = np.zeros((n, 3), np.int16)
        occ = np.zeros(n, np.int8)
        self.L.ora_tsdf_export_occupied(self.h, _p(idx), _p(occ), n)
        return idx, occ

    def import_sparse(self, sid, indices, tsdf, w, occ, color=None):
        indices = np.ascontiguousarray(indices, dtype=np.int16)
        tsdf = np.ascontiguousarray(tsdf, dtype=np.float16)
        w = np.ascontiguousarray(w, dtype=np.float16)
        occ = np.ascontiguousarray(occ, dtype=np.int8)
        col = np.ascontiguousarray(color, dtype=np.float16) if color is not None and getattr(color, "size", 0) else None
        self.L.ora_tsdf_import_sparse(self.h, sid, _p(indices), _p(tsdf), _p(w), _p(occ), _p(col), indices.shape[0])

    def surface_voxels(self, cap=1 << 22):
        xyz = np.zeros((cap, 3), np.float32)
        rgb = np.zeros((cap, 3), np.float32)
        n = int(self.L.ora_tsdf_surface_voxels(self.h, _p(xyz), _p(rgb), cap))
        return xyz[:min(n, cap)], rgb[:min(n, cap)], n

    def slice_voxels(self, z, dz=0.5, cap=1 << 22):
        xyz = np.zeros((cap, 3), np.float32)
        val = np.zeros(cap, np.float32)
        rgb = np.zeros((cap, 3), np.float32)
        n = int(self.L.ora_tsdf_slice_voxels(self.h, z, dz, _p(xyz), _p(val), _p(rgb), cap))
        return xyz[:min(n, cap)], val[:min(n, cap)], rgb[:min(n, cap)], n

    def fuse_submaps(self, sub, mode=BATCHED):
        self.L.ora_tsdf_fuse_submaps(self.h, sub.h, mode)

    # dense accumulate / finalize (multi-rank merge); same call shape as DenseTSDF.fuse_accumulate / fuse_finalize
    def fuse_accumulate(self, sub, acc, cnt):
        self.L.ora_tsdf_fuse_accumulate_dense(self.h, sub.h, _p(acc), _p(cnt))

    def fuse_finalize(self, acc, cnt):
        self.L.ora_tsdf_fuse_finalize_dense(self.h, _p(acc), _p(cnt))

    def generate_mesh(self, step=1, surface_thres=0.1, max_tri=1000000):
        v = np.zeros((max_tri * 3, 3), np.float32)
        nrm = np.zeros((max_tri * 3, 3), np.float32)
        col = np.zeros((max_tri * 3, 3), np.float32) if self.texture else None
        n = int(self.L.ora_mesh_generate(self.h, step, surface_thres, max_tri, _p(v), _p(nrm), _p(col)))
        k = min(n, max_tri) * 3
        return v[:k], nrm[:k], (col[:k] if col is not None else None), n

    def query_points(self, mode, xyz, param=0):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(xyz.shape[0], np.uint8)
        self.L.ora_tsdf_query_points(self.h, mode, param, _p(xyz), xyz.shape[0], _p(out))
        return out.astype(bool)

    def raycast(self, pos, dir, max_dist):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        dir = np.ascontiguousarray(dir, dtype=np.float32).reshape(-1, 3)
        n = pos.shape[0]
        hit = np.zeros(n, np.uint8); end = np.zeros((n, 3), np.float32); ln = np.zeros(n, np.float32)
        self.L.ora_tsdf_query_raycast(self.h, _p(pos), _p(dir), max_dist, n, _p(hit), _p(end), _p(ln))
        return hit.astype(bool), end, ln

    def esdf(self, gamma=None, max_dist=None):
        n = self.count_active()
        idx = np.zeros((n, 3), np.int16)
        e = np.zeros(n, np.float32)
        g = self.cfg.voxel_scale if gamma is None else gamma
        md = self.cfg.max_ray_length if max_dist is None else max_dist
        self.L.ora_esdf_compute(self.h, g, md, _p(idx), _p(e), n)
        return idx, e


class OracleOctomap:
    """CPU oracle of Octomap (reference: taichi_slam/mapping/taichi_octomap.py)."""

    def __init__(self, map_scale=(10, 10), voxel_scale=0.05, min_occupy_thres=3, texture_enabled=False,
                 min_ray_length=0.3, max_ray_length=3.0, K=2, max_submap_num=1024, disp_ceiling=10.0,
                 disp_floor=-10.0, is_global_map=False, recast_step=2, color_same_proj=True):
        self.cfg = OctoCfg(float(map_scale[0]), float(map_scale[1]), float(voxel_scale), float(min_occupy_thres),
                           int(texture_enabled), float(min_ray_length), float(max_ray_length), int(K),
                           int(max_submap_num), float(disp_ceiling), float(disp_floor), int(is_global_map),
                           int(recast_step), int(color_same_proj))
        self.L = lib()
        self.h = self.L.ora_octo_create(C.byref(self.cfg))
        d = [C.c_int() for _ in range(4)]
        vs = C.c_double()
        self.L.ora_octo_get_dims(self.h, *[C.byref(x) for x in d], C.byref(vs))
        self.N, self.Nz, self.Rxy, self.Rz = [x.value for x in d]
        self.voxel_scale = vs.value

    def __del__(self):
        try:
            self.L.ora_octo_destroy(self.h)
        except Exception:
            pass

    def set_intrinsics(self, Kdep, Kcol=None):
        _, kd = _d(Kdep, 9)
        _, kc = _d(Kdep if Kcol is None else Kcol, 9)
        self.L.ora_octo_set_intrinsics(self.h, kd, kc)

    def set_base_pose_submap(self, sid, R, T):
        _, r = _d(R, 9)
        _, t = _d(T, 3)
        self.L.ora_octo_set_base_pose_submap(self.h, sid, r, t)

    def set_active_submap(self, sid):
        self.L.ora_octo_set_active_submap(self.h, sid)

    def reset(self):
        self.L.ora_octo_reset(self.h)

    def integrate_depth(self, R, T, depth, texture=None):
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        _, r = _d(R, 9)
        _, t = _d(T, 3)
        st = FrameStats()
        if texture is not None and np.size(texture):
            tex = np.ascontiguousarray(texture, dtype=np.uint8)
            self.L.ora_octo_integrate_depth(self.h, r, t, _p(depth), depth.shape[0], depth.shape[1], _p(tex), tex.shape[0], tex.shape[1],
                                            C.byref(st))
        else:
            self.L.ora_octo_integrate_depth(self.h, r, t, _p(depth), depth.shape[0], depth.shape[1], None, 0, 0, C.byref(st))
        return st.as_dict()

    def integrate_points(self, R, T, xyz, rgb=None):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        _, r = _d(R, 9)
        _, t = _d(T, 3)
        st = FrameStats()
        c = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, 3) if rgb is not None and np.size(rgb) else None
        self.L.ora_octo_integrate_points(self.h, r, t, _p(xyz), _p(c) if c is not None else None, xyz.shape[0], C.byref(st))
        return st.as_dict()

    def export_leaves(self, with_color=False):
        n = int(self.L.ora_octo_export_leaves(self.h, None, None, None, 0))
        idx = np.zeros((n, 3), np.int32)
        cnt = np.zeros(n, np.float32)
        rgb = np.zeros((n, 3), np.float32)
        self.L.ora_octo_export_leaves(self.h, _p(idx), _p(cnt), _p(rgb), n)
        return (idx, cnt, rgb) if with_color else (idx, cnt)

    def occupied_voxels(self, level=0, cap=1 << 22, with_color=False):
        xyz = np.zeros((cap, 3), np.float32)
        rgb = np.zeros((cap, 3), np.float32) if with_color else None
        n = int(self.L.ora_octo_occupied_voxels(self.h, level, _p(xyz), _p(rgb) if with_color else None, cap))
        return (xyz[:min(n, cap)], rgb[:min(n, cap)], n) if with_color else (xyz[:min(n, cap)], n)

    def fuse_submaps(self, sub):
        self.L.ora_octo_fuse_submaps(self.h, sub.h)


def f32_to_f16_bits(x):
    return int(lib().ora_f32_to_f16(float(x)))


def f16_bits_to_f32(b):
    return float(lib().ora_f16_to_f32(int(b)))
