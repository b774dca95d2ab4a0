"""ctypes loader of libtaichislam_hip.so (include/taichislam_hip.h).  There is NO CPU fallback: if the HIP
library cannot be loaded, importing the mapping classes raises."""
import ctypes as C
import os

from . import build as _build

TSL_OK = 0
K_VOXELIZE, K_SORT, K_RAYS, K_INTEGRATE, K_FINALIZE, K_MESH, K_SEGMENTS, K_BIN, K_ESDF, K_FUSE = range(10)
KERNEL_NAMES = {K_VOXELIZE: "voxelize", K_SORT: "sort", K_RAYS: "build_rays", K_INTEGRATE: "integrate",
                K_FINALIZE: "finalize", K_MESH: "marching_cubes", K_SEGMENTS: "segments", K_BIN: "bin", K_ESDF: "esdf", K_FUSE: "fuse"}


class TsdfCfg(C.Structure):
    _fields_ = [("map_size_xy", C.c_double), ("map_size_z", C.c_double), ("voxel_scale", C.c_double),
                ("num_voxel_per_blk_axis", C.c_int32), ("max_ray_length", C.c_double), ("min_ray_length", C.c_double),
                ("internal_voxels", C.c_int32), ("max_submap_num", C.c_int32), ("is_global_map", C.c_int32),
                ("texture_enabled", C.c_int32), ("disp_ceiling", C.c_double), ("disp_floor", C.c_double),
                ("recast_step", C.c_int32), ("color_same_proj", C.c_int32), ("max_disp_particles", C.c_int64),
                ("max_bricks", C.c_int32), ("max_frame_bricks", C.c_int32), ("max_points", C.c_int32)]


class OctoCfg(C.Structure):
    _fields_ = [("map_size_xy", C.c_double), ("map_size_z", C.c_double), ("voxel_scale", C.c_double),
                ("min_occupy_thres", C.c_double), ("texture_enabled", C.c_int32),
                ("min_ray_length", C.c_double), ("max_ray_length", C.c_double), ("K", C.c_int32),
                ("max_submap_num", C.c_int32), ("disp_ceiling", C.c_double), ("disp_floor", C.c_double),
                ("is_global_map", C.c_int32), ("recast_step", C.c_int32), ("color_same_proj", C.c_int32),
                ("max_disp_particles", C.c_int64), ("max_bricks", C.c_int32), ("max_points", C.c_int32)]


class FrameStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("p_used", "p_valid", "p_oob", "v_pcl", "v_skipped", "steps",
                                          "steps_oob", "unique", "bricks")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class EsdfStats(C.Structure):
    _fields_ = [("incremental", C.c_int32), ("dirty_bricks", C.c_int32), ("changed_bricks", C.c_int32), ("region_bricks", C.c_int32), ("total_bricks", C.c_int32),
                ("brick_relaxations", C.c_int64), ("voxel_pushes", C.c_int64), ("rounds", C.c_int32), ("max_passes", C.c_int32), ("raise_sets", C.c_int32), ("passes", C.c_int64), ("voxels_raised", C.c_int64), ("max_raise_sets", C.c_int32), ("reserved_", C.c_int32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class EsdfTotals(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("updates", "incremental", "dirty_bricks", "region_bricks", "brick_relaxations", "voxel_pushes", "passes")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class TslError(RuntimeError):
    pass


_LIB = None

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
dp = C.POINTER(C.c_double)
pi32, pi64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)

# name -> (restype, argtypes); every symbol declared in include/taichislam_hip.h
SIGNATURES = {
    "tsl_version": (C.c_char_p, []),
    "tsl_last_error": (C.c_char_p, []),
    "tsl_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "tsl_selftest": (C.c_int, [C.c_int, C.POINTER(C.c_int64)]),
    "tsl_tsdf_create": (C.c_int, [C.POINTER(TsdfCfg), C.c_int, C.POINTER(vp)]),
    "tsl_tsdf_destroy": (None, [vp]),
    "tsl_tsdf_get_dims": (C.c_int, [vp, pi32, pi32, pi32, pi32]),
    "tsl_tsdf_sync": (C.c_int, [vp]),
    "tsl_tsdf_reset": (C.c_int, [vp]),
    "tsl_tsdf_memory_bytes": (C.c_int, [vp, pi64]),
    "tsl_tsdf_bricks_in_use": (C.c_int, [vp, pi32]),
    "tsl_tsdf_set_intrinsics": (C.c_int, [vp, dp, dp]),
    "tsl_tsdf_set_base_pose": (C.c_int, [vp, dp, dp]),
    "tsl_tsdf_set_base_pose_submap": (C.c_int, [vp, C.c_int, dp, dp]),
    "tsl_tsdf_get_active_submap": (C.c_int, [vp, pi32]),
    "tsl_tsdf_set_active_submap": (C.c_int, [vp, i32]),
    "tsl_tsdf_set_colormap": (C.c_int, [vp, vp]),
    "tsl_tsdf_integrate_depth": (C.c_int, [vp, dp, dp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]),
    "tsl_tsdf_integrate_depth_dev": (C.c_int, [vp, dp, dp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]),
    "tsl_tsdf_integrate_points": (C.c_int, [vp, dp, dp, vp, vp, i64]),
    "tsl_tsdf_integrate_points_dev": (C.c_int, [vp, dp, dp, vp, vp, i64]),
    "tsl_tsdf_input_stream": (C.c_int, [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]),
    "tsl_tsdf_integrate_depth_stream": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tsl_tsdf_queued_frames": (C.c_int, [vp, pi32]),
    "tsl_tsdf_frames_consumed": (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tsl_tsdf_last_frame_stats": (C.c_int, [vp, C.POINTER(FrameStats)]),
    "tsl_tsdf_count_active": (C.c_int, [vp, pi64]),
    "tsl_tsdf_export_sparse": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, pi64]),
    "tsl_tsdf_import_sparse": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, i64]),
    "tsl_tsdf_export_occupied": (C.c_int, [vp, vp, vp, i64, pi64]),
    "tsl_tsdf_surface_voxels": (C.c_int, [vp, vp, C.c_int, pi32]),
    "tsl_tsdf_slice_voxels": (C.c_int, [vp, f32, f32, C.c_int, pi32]),
    "tsl_tsdf_read_exports": (C.c_int, [vp, vp, vp, vp, i64]),
    "tsl_tsdf_exports_dev": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), pi32]),
    "tsl_tsdf_set_export_row": (C.c_int, [vp, C.c_int, i64, vp]),
    "tsl_tsdf_pack_pointcloud2": (C.c_int, [vp, C.c_int, i64, vp]),
    "tsl_tsdf_num_particles": (C.c_int, [vp, pi32]),
    "tsl_tsdf_set_num_particles": (C.c_int, [vp, i32]),
    "tsl_tsdf_fuse_submaps": (C.c_int, [vp, vp]),
    "tsl_comm_unique_id": (C.c_int, [C.c_char_p]),
    "tsl_comm_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "tsl_comm_destroy": (None, [vp]),
    "tsl_comm_handle": (vp, [vp]),
    "tsl_tsdf_allreduce_merge": (C.c_int, [vp, vp, vp, pi64]),
    "tsl_tsdf_merge_mask_bytes": (C.c_int, [vp, pi64]),
    "tsl_tsdf_merge_begin": (C.c_int, [vp, vp, vp, i64]),
    "tsl_tsdf_merge_union": (C.c_int, [vp, vp, pi32]),
    "tsl_tsdf_merge_pack": (C.c_int, [vp, vp, vp]),
    "tsl_tsdf_merge_finish": (C.c_int, [vp, vp, vp]),
    "tsl_tsdf_merge_record_bytes": (C.c_int, [pi64]),
    "tsl_tsdf_merge_finalize_slice": (C.c_int, [vp, vp, vp, i32, vp]),
    "tsl_tsdf_merge_finish_records": (C.c_int, [vp, vp]),
    "tsl_mesh_generate": (C.c_int, [vp, C.c_int, f32, i64, pi32]),
    "tsl_mesh_read": (C.c_int, [vp, vp, vp, vp, i64]),
    "tsl_mesh_buffers_dev": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), pi32]),
    "tsl_tsdf_query_points": (C.c_int, [vp, C.c_int, C.c_int, vp, i64, vp]),
    "tsl_tsdf_query_raycast": (C.c_int, [vp, vp, vp, f32, i64, vp, vp, vp]),
    "tsl_tsdf_query_points_dev": (C.c_int, [vp, C.c_int, C.c_int, vp, i64, vp, vp]),
    "tsl_tsdf_query_raycast_dev": (C.c_int, [vp, vp, vp, f32, i64, vp, vp, vp, vp]),
    "tsl_esdf_update": (C.c_int, [vp, f32, f32, pi32]),
    "tsl_esdf_last_stats": (C.c_int, [vp, C.POINTER(EsdfStats)]),
    "tsl_esdf_totals": (C.c_int, [vp, C.POINTER(EsdfTotals)]),
    "tsl_esdf_export": (C.c_int, [vp, vp, vp, i64, pi64]),
    "tsl_esdf_export_dev": (C.c_int, [vp, i64, C.POINTER(vp), C.POINTER(vp), pi64]),
    "tsl_esdf_slice": (C.c_int, [vp, f32, pi32]),
    "tsl_esdf_read_slice": (C.c_int, [vp, vp, vp, i64]),
    "tsl_esdf_slice_dev": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), pi32]),
    "tsl_tsdf_set_option": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "tsl_tsdf_get_option": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_int)]),
    "tsl_tsdf_prof_enable": (C.c_int, [vp, C.c_int]),
    "tsl_tsdf_prof_query": (C.c_int, [vp, C.c_int, dp, pi64]),
    "tsl_octo_create": (C.c_int, [C.POINTER(OctoCfg), C.c_int, C.POINTER(vp)]),
    "tsl_octo_destroy": (None, [vp]),
    "tsl_octo_get_dims": (C.c_int, [vp, pi32, pi32, pi32, pi32, dp]),
    "tsl_octo_sync": (C.c_int, [vp]),
    "tsl_octo_reset": (C.c_int, [vp]),
    "tsl_octo_set_intrinsics": (C.c_int, [vp, dp, dp]),
    "tsl_octo_set_base_pose_submap": (C.c_int, [vp, C.c_int, dp, dp]),
    "tsl_octo_get_active_submap": (C.c_int, [vp, pi32]),
    "tsl_octo_set_active_submap": (C.c_int, [vp, i32]),
    "tsl_octo_integrate_depth": (C.c_int, [vp, dp, dp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]),
    "tsl_octo_integrate_depth_dev": (C.c_int, [vp, dp, dp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]),
    "tsl_octo_integrate_points": (C.c_int, [vp, dp, dp, vp, vp, i64]),
    "tsl_octo_integrate_points_dev": (C.c_int, [vp, dp, dp, vp, vp, i64]),
    "tsl_octo_last_frame_stats": (C.c_int, [vp, C.POINTER(FrameStats)]),
    "tsl_octo_export_leaves": (C.c_int, [vp, vp, vp, vp, i64, pi64]),
    "tsl_octo_occupied_voxels": (C.c_int, [vp, vp, C.c_int, C.c_int, pi32]),
    "tsl_octo_read_exports": (C.c_int, [vp, vp, vp, i64]),
    "tsl_octo_exports_dev": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), pi32]),
    "tsl_octo_pack_pointcloud2": (C.c_int, [vp, C.c_int, i64, vp]),
    "tsl_octo_num_particles": (C.c_int, [vp, pi32]),
    "tsl_octo_fuse_submaps": (C.c_int, [vp, vp]),
}


def library_path():
    return _build.LIB


def lib():
    """Load (building first if the sources are newer and hipcc exists) the HIP library.  Raises if unavailable."""
    global _LIB
    if _LIB is None:
        # TSL_LIB: developer aid, an alternative build of the same library (e.g. a -DTSL_TIMING build for tools/timing_probe*.py)
        path = os.environ.get("TSL_LIB") or _build.build_library()
        if not os.path.exists(path):
            raise TslError("libtaichislam_hip.so is missing; run `python -m taichislam_amd.build`")
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != TSL_OK:
        raise TslError(f"taichislam_hip error {rc}: {lib().tsl_last_error().decode()}")


def device_count():
    n = C.c_int(0)
    lib().tsl_device_count(C.byref(n))
    return n.value


_HOST_FN = None


_OCTO_DEV_FN = None


def octo_integrate_depth_dev_fn():
    """tsl_octo_integrate_depth_dev with integer pointer arguments (no ctypes pointer objects per call): the per-frame path of a device-resident stream into the
    Octomap, which the library only queues -- the host side of a call is what a frame costs."""
    global _OCTO_DEV_FN
    if _OCTO_DEV_FN is None:
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int)
        _OCTO_DEV_FN = proto(("tsl_octo_integrate_depth_dev", lib()))
    return _OCTO_DEV_FN


def integrate_depth_host_fn():
    """tsl_tsdf_integrate_depth bound a second time with integer pointer arguments (no ctypes pointer objects per call): the per-frame path of
    a host-image stream, see DenseTSDF.recast_depth_to_map."""
    global _HOST_FN
    if _HOST_FN is None:
        lib()
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int)
        _HOST_FN = proto(("tsl_tsdf_integrate_depth", lib()))
    return _HOST_FN
