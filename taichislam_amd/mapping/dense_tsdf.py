"""DenseTSDF: drop-in for taichi_slam.mapping.DenseTSDF (reference taichi_slam/mapping/dense_tsdf.py:12-516)
running on hand-written HIP kernels for MI355X through include/taichislam_hip.h.

Same constructor keywords, methods and attribute names as the reference; numpy in / numpy out.  Depth images
and point clouds may also be torch CUDA tensors (their device pointer is handed to the *_dev entry points)."""
import ctypes as C
import os
import math
import time

import numpy as np

from .. import _lib
from .fields import DeviceArrayField, MapFieldRef, ScalarField
from .mapping_common import BaseMap, _dptr, jet_colormap

Wmax = 1000


def _is_device_tensor(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda") and x.is_cuda


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_TORCH = None


def _torch():
    global _TORCH
    if _TORCH is None:
        import torch
        _TORCH = torch
    return _TORCH


def _DEPTH_DTYPES(torch, _cache=[]):
    if not _cache:
        _cache.append((torch.int16, getattr(torch, "uint16", torch.int16)))
    return _cache[0]


def _f64(a, n):
    """`a` as a C-contiguous float64 array of n values; the array itself when it already is one (the common case)."""
    if type(a) is np.ndarray and a.dtype == np.float64 and a.size == n and a.flags.c_contiguous:
        return a
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    if a.size != n:
        raise ValueError(f"expected {n} values, got {a.size}")
    return a


class DenseTSDF(BaseMap):
    _prefix = "tsl_tsdf"

    def __init__(self, map_scale=[10, 10], voxel_scale=0.05, texture_enabled=False,
                 max_disp_particles=1024 * 1024, num_voxel_per_blk_axis=16, max_ray_length=10, min_ray_length=0.3,
                 internal_voxels=10, max_submap_num=1024, is_global_map=False,
                 disp_ceiling=1.8, disp_floor=-0.3, recast_step=2, color_same_proj=True,
                 # legacy keywords still passed by tests/marching_cube_test.py:12-16 and TaichiSLAM_demo.py:147-152
                 min_occupy_thres=None, enable_esdf=None,
                 # backend knobs (not in the reference)
                 device=0, max_bricks=0, max_frame_bricks=0, max_points=0):
        super(DenseTSDF, self).__init__(voxel_scale)
        self.num_voxel_per_blk_axis = num_voxel_per_blk_axis
        self.voxel_scale = voxel_scale
        self.N = math.ceil(map_scale[0] / voxel_scale / num_voxel_per_blk_axis) * num_voxel_per_blk_axis
        self.Nz = math.ceil(map_scale[1] / voxel_scale / num_voxel_per_blk_axis) * num_voxel_per_blk_axis
        self.block_num_xy = math.ceil(map_scale[0] / voxel_scale / num_voxel_per_blk_axis)
        self.block_num_z = math.ceil(map_scale[1] / voxel_scale / num_voxel_per_blk_axis)
        self.map_size_xy = voxel_scale * self.N
        self.map_size_z = voxel_scale * self.Nz
        self.max_disp_particles = max_disp_particles
        self.enable_texture = texture_enabled
        self.max_ray_length = max_ray_length
        self.min_ray_length = min_ray_length
        self.tsdf_surface_thres = self.voxel_scale * 1.8
        self.internal_voxels = internal_voxels
        self.max_submap_num = max_submap_num
        self.is_global_map = is_global_map
        self.disp_ceiling = disp_ceiling
        self.disp_floor = disp_floor
        self.recast_step = recast_step
        self.color_same_proj = color_same_proj
        self.clear_last_TSDF_exporting = False
        self.device = device
        self._held, self._pending_inputs = [], None
        self._c_total, self._c_done, self._c_stream = C.c_int64(), C.c_int64(), C.c_void_p()
        self._ref_total, self._ref_done = C.byref(self._c_total), C.byref(self._c_done)
        self._integrate_depth_stream = self.L.tsl_tsdf_integrate_depth_stream
        self._integrate_depth_host = _lib.integrate_depth_host_fn()
        self.mem_per_voxel = 2 + 2 + 1 + 1 + (6 if texture_enabled else 0)

        cfg = _lib.TsdfCfg(float(map_scale[0]), float(map_scale[1]), float(voxel_scale), int(num_voxel_per_blk_axis),
                           float(max_ray_length), float(min_ray_length), int(internal_voxels), int(max_submap_num),
                           int(bool(is_global_map)), int(bool(texture_enabled)), float(disp_ceiling), float(disp_floor),
                           int(recast_step), int(bool(color_same_proj)), int(max_disp_particles),
                           int(max_bricks), int(max_frame_bricks), int(max_points))
        h = C.c_void_p()
        _lib.check(self.L.tsl_tsdf_create(C.byref(cfg), int(device), C.byref(h)))
        self.h = h
        n, nz = C.c_int32(), C.c_int32()
        self._call("get_dims", C.byref(n), C.byref(nz), None, None)
        assert (n.value, nz.value) == (self.N, self.Nz), "host/device extent mismatch"
        self.initialize_fields()
        if os.environ.get("TSL_ESDF_MODE"):          # developer / test aid: the form of the incremental ESDF update for handles made from here on (set_option("esdf_mode", ...))
            self.set_option("esdf_mode", int(os.environ["TSL_ESDF_MODE"]))
        print(f"TSDF map initialized blocks {self.block_num_xy}x{self.block_num_xy}x{self.block_num_z}")

    # ---- fields (dense_tsdf.py:52-106) -------------------------------------------------------------------
    def initialize_fields(self):
        self.num_TSDF_particles = ScalarField(self._get_num_particles, self._set_num_particles, "num_TSDF_particles")
        self.num_export_particles = ScalarField(lambda: 0, None, "num_export_particles")
        md = self.max_disp_particles
        self.num_export_ESDF_particles = ScalarField(lambda: self._esdf_slice_n, None, "num_export_ESDF_particles")
        self._esdf_slice_n = 0
        self.export_TSDF_xyz = DeviceArrayField(self, lambda n: self._read_exports(n)[0], md, 3, "export_TSDF_xyz",
                                                writer=lambda row, v: self._call("set_export_row", 0, row, _vp(np.ascontiguousarray(v[:3], np.float32))),
                                                dev=lambda: self._exports_dev(0))
        self.export_color = DeviceArrayField(self, lambda n: self._read_exports(n)[1], md, 3, "export_color",
                                             writer=lambda row, v: self._call("set_export_row", 1, row, _vp(np.ascontiguousarray(v[:3], np.float32))),
                                             dev=lambda: self._exports_dev(1))
        self.export_TSDF = DeviceArrayField(self, lambda n: self._read_exports(n)[2], md, 1, "export_TSDF", dev=lambda: self._exports_dev(2))
        # dense_esdf.py:112-113 (legacy module): the ESDF slice exports
        self.export_ESDF_xyz = DeviceArrayField(self, lambda n: self._read_esdf_slice(n)[0], md, 3, "export_ESDF_xyz", dev=lambda: self._esdf_slice_dev(0))
        self.export_ESDF = DeviceArrayField(self, lambda n: self._read_esdf_slice(n)[1], md, 1, "export_ESDF", dev=lambda: self._esdf_slice_dev(1))
        self.export_x = self.export_TSDF_xyz
        self.TSDF = MapFieldRef(self, "TSDF")
        self.W_TSDF = MapFieldRef(self, "W_TSDF")
        self.TSDF_observed = MapFieldRef(self, "TSDF_observed")
        self.occupy = MapFieldRef(self, "occupy")
        self.color = MapFieldRef(self, "color") if self.enable_texture else None
        self.colormap = jet_colormap()
        self._call("set_colormap", _vp(self.colormap))
        self.initialize_submap_fields(self.max_submap_num)

    def _get_num_particles(self):
        v = C.c_int32()
        self._call("num_particles", C.byref(v))
        return v.value

    def _set_num_particles(self, n):
        self._call("set_num_particles", int(n))

    def _read_exports(self, n):
        n = int(max(0, min(n, self.max_disp_particles)))
        xyz = np.empty((n, 3), np.float32)
        rgb = np.empty((n, 3), np.float32)
        val = np.empty(n, np.float32)
        self._call("read_exports", _vp(xyz), _vp(rgb), _vp(val), n)
        return xyz, rgb, val

    def _exports_dev(self, which):
        """(device pointer of export_TSDF_xyz / export_color / export_TSDF, particles of the last cvt_* call)"""
        p = [C.c_void_p(), C.c_void_p(), C.c_void_p()]
        n = C.c_int32()
        self._call("exports_dev", C.byref(p[0]), C.byref(p[1]), C.byref(p[2]), C.byref(n))
        return p[which].value, max(0, min(n.value, self.max_disp_particles))

    def _read_esdf_slice(self, n):
        n = int(max(0, min(n, self.max_disp_particles, self._esdf_slice_n)))
        xyz = np.empty((n, 3), np.float32)
        val = np.empty(n, np.float32)
        _lib.check(self.L.tsl_esdf_read_slice(self.h, _vp(xyz), _vp(val), n))
        return xyz, val

    def _esdf_slice_dev(self, which):
        p = [C.c_void_p(), C.c_void_p()]
        n = C.c_int32()
        _lib.check(self.L.tsl_esdf_slice_dev(self.h, C.byref(p[0]), C.byref(p[1]), C.byref(n)))
        return p[which].value, max(0, min(n.value, self.max_disp_particles))

    # ---- backend knobs ---------------------------------------------------------------------------------
    def set_option(self, name, value):
        self._call("set_option", name.encode(), int(value))

    def get_option(self, name):
        v = C.c_int()
        self._call("get_option", name.encode(), C.byref(v))
        return v.value

    def enable_profiling(self, on=True, only=None):
        """HIP-event timing of the per-frame kernels; `only` = iterable of kernel ids restricts it (less host overhead)."""
        if on and only is not None:
            mask = 0
            for k in only:
                mask |= 1 << int(k)
            self._call("prof_enable", 2 * mask)
        else:
            self._call("prof_enable", int(bool(on)))

    def kernel_time(self, kernel_id):
        """(total_ms, launches) recorded by HIP events since the last query; synchronises."""
        ms, n = C.c_double(), C.c_int64()
        self._call("prof_query", int(kernel_id), C.byref(ms), C.byref(n))
        return ms.value, n.value

    def last_frame_stats(self):
        st = _lib.FrameStats()
        self._call("last_frame_stats", C.byref(st))
        return st.as_dict()

    def memory_bytes(self):
        v = C.c_int64()
        self._call("memory_bytes", C.byref(v))
        return v.value

    def bricks_in_use(self):
        v = C.c_int32()
        self._call("bricks_in_use", C.byref(v))
        return v.value

    # ---- integration (dense_tsdf.py:157-165) -----------------------------------------------------------------
    def _adopt_device_inputs(self, points, *tensors):
        """The reference's recast_* calls are synchronous; here a frame is only queued and its kernels are enqueued later (when
        its batch is full, or when anything else needs the map) on the library's own streams.  For torch tensors the
        shim therefore (1) has the stream that will read the frame wait for the work already queued on torch's current
        stream (the tensor may still be being produced; tsl_tsdf_input_stream) and (2) keeps the tensors referenced until the
        device has read them (tsl_tsdf_frames_consumed; the library never lets the host run more than eight batches ahead,
        so at most ~72 frames are held), so a tensor the caller drops right after the call is not recycled by torch's
        caching allocator under a queued frame.  No torch stream / event queries: they cost ~100 us each while the GPU is busy."""
        import torch
        cur = torch.cuda.current_stream(tensors[0].device)
        self._call("input_stream", int(points), 1, C.c_void_p(cur.cuda_stream), C.byref(self._c_stream))
        self._pending_inputs = tensors

    def _release_device_inputs(self, force=False):
        """Drop the references to input tensors whose frames have been read (called right after the frame was queued)."""
        if force:
            self._held.clear(); self._pending_inputs = None
            return
        total, done = self._c_total, self._c_done                             # preallocated: no per-frame garbage for the cyclic GC
        self._call("frames_consumed", C.byref(total), C.byref(done))          # host-side counters only: no device query
        if self._pending_inputs is not None:
            self._held.append((total.value - 1, self._pending_inputs))       # index of the frame just queued
            self._pending_inputs = None
        while self._held and self._held[0][0] < done.value:
            self._held.pop(0)

    def sync(self):
        super().sync()
        self._release_device_inputs(force=True)

    def __del__(self):
        try:
            if self.h is not None and (self._held or self._pending_inputs is not None):
                super().sync()                                  # nothing may still read the tensors when they are released
            self._held.clear(); self._pending_inputs = None
        except Exception:
            pass
        super().__del__()

    def recast_pcl_to_map(self, R, T, xyz_array, rgb_array=None, n=None):
        """recast_pcl_to_map(R, T, xyz_array, rgb_array); the third positional `n` of the stale demo
        (TaichiSLAM_demo.py:52) is tolerated and ignored."""
        r, t = _dptr(R, 9)[1], _dptr(T, 3)[1]
        if _is_device_tensor(xyz_array):
            import torch
            x = xyz_array.reshape(-1, 3).contiguous().float()
            ins, rgb_ptr = [x], None
            if self.enable_texture and rgb_array is not None and _is_device_tensor(rgb_array) and rgb_array.numel():
                c = rgb_array.reshape(-1, 3).contiguous().to(torch.uint8)
                assert c.shape[0] == x.shape[0], "rgb_array must hold one colour per point"
                ins.append(c)
                rgb_ptr = C.c_void_p(c.data_ptr())
            self._adopt_device_inputs(1, *ins)
            self._call("integrate_points_dev", r, t, C.c_void_p(x.data_ptr()), rgb_ptr, int(x.shape[0]))
            self._release_device_inputs()
            return
        xyz = np.ascontiguousarray(np.asarray(xyz_array, dtype=np.float32).reshape(-1, 3))
        rgb = None
        if self.enable_texture and rgb_array is not None and getattr(rgb_array, "size", 0):
            rgb = np.ascontiguousarray(np.asarray(rgb_array, dtype=np.uint8).reshape(-1, 3))
        self._call("integrate_points", r, t, _vp(xyz), _vp(rgb), int(xyz.shape[0]))

    def recast_depth_to_map(self, R, T, depthmap, texture=None):
        if _is_device_tensor(depthmap):
            # the per-frame path of a device-resident stream: ONE crossing into the library (input ordering against torch's current
            # stream + queueing + the consumed-frames count), no numpy / ctypes temporaries for poses that already are float64 arrays
            torch = _torch()
            assert depthmap.dim() == 2 and depthmap.is_contiguous() and depthmap.dtype in _DEPTH_DTYPES(torch), \
                "device depth must be a contiguous [h,w] uint16 tensor of millimetres (int16 = the same bits)"
            ins, tex_ptr, th, tw = (depthmap,), None, 0, 0
            if self.enable_texture and texture is not None and _is_device_tensor(texture):
                assert texture.dtype == torch.uint8 and texture.is_contiguous() and texture.dim() == 3 and texture.shape[2] == 3
                ins = (depthmap, texture)
                tex_ptr, th, tw = texture.data_ptr(), int(texture.shape[0]), int(texture.shape[1])
            Ra, Ta = _f64(R, 9), _f64(T, 3)
            total, done = self._c_total, self._c_done                         # preallocated: no per-frame garbage for the cyclic GC
            shape = depthmap.shape
            _lib.check(self._integrate_depth_stream(self.h, Ra.ctypes.data, Ta.ctypes.data, depthmap.data_ptr(), shape[0], shape[1],
                                                    tex_ptr, th, tw, torch.cuda.current_stream(depthmap.device).cuda_stream,
                                                    self._ref_total, self._ref_done))
            # keep the tensors referenced until the device has read them (see _adopt_device_inputs)
            held = self._held
            held.append((total.value - 1, ins))
            d = done.value
            while held and held[0][0] < d:
                held.pop(0)
            return
        if type(depthmap) is np.ndarray and depthmap.dtype == np.uint16 and depthmap.ndim == 2 and depthmap.flags.c_contiguous and (texture is None or not self.enable_texture):
            # the reference node's calling convention (taichislam_node.py:381-382: a uint16 numpy image per frame) without numpy / ctypes temporaries:
            # pointers as plain integers, one crossing into the library (the host side of a 25 k frames/s stream has ~40 us per call)
            Ra, Ta = _f64(R, 9), _f64(T, 3)
            rc = self._integrate_depth_host(self.h, Ra.ctypes.data, Ta.ctypes.data, depthmap.ctypes.data, depthmap.shape[0], depthmap.shape[1], None, 0, 0)
            if rc:
                _lib.check(rc)
            return
        r, t = _dptr(R, 9)[1], _dptr(T, 3)[1]
        depth = np.ascontiguousarray(np.asarray(depthmap, dtype=np.uint16))
        if depth.ndim != 2:
            raise ValueError("depthmap must be a 2-d uint16 array")
        tex, th, tw = None, 0, 0
        if self.enable_texture and texture is not None and getattr(texture, "size", 0):
            tex = np.ascontiguousarray(np.asarray(texture, dtype=np.uint8))
            th, tw = tex.shape[:2]
        self._call("integrate_depth", r, t, _vp(depth), depth.shape[0], depth.shape[1], _vp(tex), th, tw)

    # north-star alias (BASELINE.json says "integrate"; the reference method is recast_depth_to_map)
    integrate = recast_depth_to_map

    # ---- fusion (dense_tsdf.py:309-318) -------------------------------------------------------------------------
    def reset(self):
        self._call("reset")

    def fuse_submaps(self, submaps):
        t = time.time()
        # the kernel reads the global map's pose table (dense_tsdf.py:286-293): make sure the handle has it
        _lib.check(self.L.tsl_tsdf_fuse_submaps(self.h, submaps.h))
        print(f"[DenseTSDF] Fuse submaps {(time.time() - t) * 1000:.1f}ms, active local: "
              f"{submaps.active_submap_id[None]} remote: {submaps.remote_submap_num[None]}")

    # multi-GPU merge (include/taichislam_hip.h; taichislam_amd.distributed.allreduce_merge drives these)
    def allreduce_merge(self, submaps, comm=None):
        """One native call: splat `submaps`, all-reduce over `comm` (taichislam_amd.distributed.Communicator, None = this rank
        alone), finalise.  Returns the bytes all-reduced."""
        n = C.c_int64()
        _lib.check(self.L.tsl_tsdf_allreduce_merge(self.h, submaps.h, comm.handle if comm is not None else None, C.byref(n)))
        return n.value

    def merge_begin(self, submaps):
        """Step 1: reset this global map, splat `submaps` into the per-brick sums; returns the touched-brick byte mask (torch CUDA)."""
        import torch
        nb = C.c_int64()
        self._call("merge_mask_bytes", C.byref(nb))
        mask = torch.zeros(nb.value, dtype=torch.uint8, device=f"cuda:{self.device}")
        torch.cuda.current_stream(mask.device).synchronize()
        _lib.check(self.L.tsl_tsdf_merge_begin(self.h, submaps.h, C.c_void_p(mask.data_ptr()), nb.value))
        return mask

    def empty_merge_mask(self):
        """An all-zero brick mask: what a rank that could not splat its submaps contributes to the exchange (distributed.allreduce_merge)."""
        import torch
        nb = C.c_int64()
        self._call("merge_mask_bytes", C.byref(nb))
        return torch.zeros(nb.value, dtype=torch.uint8, device=f"cuda:{self.device}")

    def merge_pack(self, mask):
        """Step 2 (after the MAX all-reduce of the mask): packed sums of the union bricks, (int64 [n,4096,2], int32 [n,4096])."""
        import torch
        n = C.c_int32()
        self._call("merge_union", C.c_void_p(mask.data_ptr()), C.byref(n))
        acc = torch.empty((n.value, 4096, 2), dtype=torch.int64, device=mask.device)
        cnt = torch.empty((n.value, 4096), dtype=torch.int32, device=mask.device)
        torch.cuda.current_stream(mask.device).synchronize()
        self._call("merge_pack", C.c_void_p(acc.data_ptr()) if n.value else None, C.c_void_p(cnt.data_ptr()) if n.value else None)
        return acc, cnt

    def merge_finish(self, acc, cnt):
        """Step 3 (after the SUM all-reduce of both): write the global map."""
        self._call("merge_finish", C.c_void_p(acc.data_ptr()) if len(acc) else None, C.c_void_p(cnt.data_ptr()) if len(cnt) else None)

    MERGE_RECORD_BYTES = 4096 * 4 + 4096 + 4096 // 8          # tsl_tsdf_merge_record_bytes

    def merge_finalize_slice(self, acc, cnt):
        """Reduce-scatter form, step 3: the union bricks whose reduced sums this rank holds (int64 [k,4096,2], int32 [k,4096]) as finalised records, uint8 [k, 20992]."""
        import torch
        k = int(acc.shape[0])
        rec = torch.empty((k, self.MERGE_RECORD_BYTES), dtype=torch.uint8, device=acc.device)
        torch.cuda.current_stream(acc.device).synchronize()
        _lib.check(self.L.tsl_tsdf_merge_finalize_slice(self.h, C.c_void_p(acc.data_ptr()) if k else None, C.c_void_p(cnt.data_ptr()) if k else None, k,
                                                        C.c_void_p(rec.data_ptr()) if k else None))
        return rec

    def merge_finish_records(self, rec):
        """Reduce-scatter form, step 4 (after the all-gather): write the global map from the records of all union bricks (those behind the union are padding)."""
        import torch
        torch.cuda.current_stream(rec.device).synchronize()
        _lib.check(self.L.tsl_tsdf_merge_finish_records(self.h, C.c_void_p(rec.data_ptr()) if len(rec) else None))

    # ---- visualisation exports (dense_tsdf.py:320-404) --------------------------------------------------------------
    def cvt_occupy_to_voxels(self):
        self.cvt_TSDF_surface_to_voxels()

    def cvt_TSDF_surface_to_voxels(self):
        n = C.c_int32()
        self._call("surface_voxels", None, 0, C.byref(n))

    def cvt_TSDF_surface_to_voxels_to(self, num_TSDF_particles, max_disp_particles, export_TSDF_xyz, export_color):
        """Append this map's surface voxels to another map's export buffers (submap_mapping.py:212-213)."""
        dst = export_TSDF_xyz._owner
        n = C.c_int32()
        self._call("surface_voxels", dst.h, 1, C.byref(n))

    def cvt_TSDF_to_voxels_slice(self, z, dz=0.5, clear_last=True):
        n = C.c_int32()
        self._call("slice_voxels", float(z), float(dz), int(bool(clear_last)), C.byref(n))

    def pointcloud2(self, n=None, has_rgb=None):
        """The first `n` exported particles (default: all of the last cvt_* call) as a sensor_msgs/PointCloud2 payload -- what
        scripts/taichislam_node.py:420-425 + utils/ros_pcl_transfer.py:96-136 assemble from numpy copies -- interleaved on the device.
        See taichislam_amd.utils.ros_adapters for the message fields."""
        from ..utils import ros_adapters
        n = self.num_TSDF_particles[None] if n is None else int(n)
        n = max(0, min(n, self.max_disp_particles))
        has_rgb = self.enable_texture if has_rgb is None else bool(has_rgb)
        data = np.empty((n, 6 if has_rgb else 3), np.float32)
        self._call("pack_pointcloud2", int(has_rgb), n, _vp(data))
        return ros_adapters.pointcloud2_payload(data, has_rgb)

    def get_voxels_TSDF_surface(self):
        self.cvt_TSDF_surface_to_voxels()
        n = self.num_TSDF_particles[None]
        xyz, rgb, val = self._read_exports(n)
        return xyz, val, (rgb if self.enable_texture else None)

    def get_voxels_TSDF_slice(self, z):
        self.cvt_TSDF_to_voxels_slice(z)
        xyz, _, val = self._read_exports(self.num_TSDF_particles[None])
        return xyz, val

    def get_voxels_occupy(self):
        self.cvt_occupy_to_voxels()
        xyz, rgb, _ = self._read_exports(self.num_TSDF_particles[None])
        return xyz, rgb

    # ---- sparse export / import (dense_tsdf.py:412-515) ----------------------------------------------------------------
    def count_active(self):
        v = C.c_int64()
        self._call("count_active", C.byref(v))
        return v.value

    def to_numpy(self, data_indices, data_tsdf, data_wtsdf, data_occ, data_color):
        n = C.c_int64()
        col = data_color if (self.enable_texture and getattr(data_color, "size", 0)) else None
        self._call("export_sparse", _vp(data_indices), _vp(data_tsdf), _vp(data_wtsdf), _vp(data_occ), _vp(col),
                   int(data_tsdf.shape[0]), C.byref(n))
        return n.value

    def load_numpy(self, submap_id, data_indices, data_tsdf, data_wtsdf, data_occ, data_color):
        idx = np.ascontiguousarray(data_indices, dtype=np.int16)
        t = np.ascontiguousarray(data_tsdf, dtype=np.float16)
        w = np.ascontiguousarray(data_wtsdf, dtype=np.float16)
        occ = np.ascontiguousarray(data_occ, dtype=np.int8)
        col = None
        if self.enable_texture and data_color is not None and getattr(data_color, "size", 0):
            col = np.ascontiguousarray(data_color, dtype=np.float16)
        self._call("import_sparse", int(submap_id), _vp(idx), _vp(t), _vp(w), _vp(occ), _vp(col), int(t.shape[0]))

    def export_submap(self):
        s = time.time()
        num = self.count_active()
        indices = np.zeros((num, 3), np.int16)
        tsdf = np.zeros((num), np.float16)
        w_tsdf = np.zeros((num), np.float16)
        occupy = np.zeros((num), np.int8)
        color = np.zeros((num, 3), np.float16) if self.enable_texture else np.array([])
        self.to_numpy(indices, tsdf, w_tsdf, occupy, color)
        obj = {
            'indices': indices, 'TSDF': tsdf, 'W_TSDF': w_tsdf, 'color': color, 'occupy': occupy,
            "map_scale": [self.map_size_xy, self.map_size_z], "voxel_scale": self.voxel_scale,
            "texture_enabled": self.enable_texture, "num_voxel_per_blk_axis": self.num_voxel_per_blk_axis,
        }
        print(f"Export submap {self.active_submap_id[None]} to numpy, voxels {num / 1024:.1f}k, "
              f"time: {1000 * (time.time() - s):.1f}ms")
        return obj

    def export_occupied(self):
        """(indices int16[n,3], occupy int8[n]) of voxels with occupy != 0 (backend extra, used by parity tests)."""
        n = C.c_int64()
        self._call("export_occupied", None, None, 0, C.byref(n))
        idx = np.zeros((n.value, 3), np.int16)
        occ = np.zeros(n.value, np.int8)
        self._call("export_occupied", _vp(idx), _vp(occ), n.value, C.byref(n))
        return idx, occ

    def saveMap(self, filename):
        np.save(filename, self.export_submap())

    @staticmethod
    def loadMap(filename, **backend_opts):
        obj = np.load(filename, allow_pickle=True).item()
        mapping = DenseTSDF(map_scale=obj['map_scale'], voxel_scale=obj['voxel_scale'],
                            texture_enabled=obj['texture_enabled'],
                            num_voxel_per_blk_axis=obj['num_voxel_per_blk_axis'], is_global_map=True, **backend_opts)
        mapping.load_numpy(0, obj['indices'], obj['TSDF'], obj['W_TSDF'], obj['occupy'], obj['color'])
        print(f"[SubmapMapping] Loaded {obj['TSDF'].shape[0]} voxels from {filename}")
        return mapping

    def input_remote_submap(self, submap):
        # remote submaps are stored from the top of the submap axis downwards (dense_tsdf.py:500-515)
        self.remote_submap_num[None] = self.remote_submap_num[None] + 1
        idx = self.max_submap_num - self.remote_submap_num[None]
        R, T = submap['pose']
        color = submap['color'] if self.enable_texture else np.array([])
        self.load_numpy(idx, submap['indices'], submap['TSDF'], submap['W_TSDF'], submap['occupy'], color)
        self.set_base_pose_submap(idx, R, T)
        return idx

    # ---- batched map queries (mapping_common.py:165-204; the reference exposes them as ti.funcs for TopoGraphGen) ----
    def _query_points(self, mode, xyz, param=0):
        if _is_device_tensor(xyz):
            # device tensors in, device tensor out, nothing waited for: the query runs on the map's stream behind every queued frame and
            # torch's current stream is ordered after it (tsl_tsdf_query_points_dev)
            torch = _torch()
            x = xyz.reshape(-1, 3).contiguous().float()
            out = torch.empty(x.shape[0], dtype=torch.uint8, device=x.device)
            _lib.check(self.L.tsl_tsdf_query_points_dev(self.h, mode, int(param), x.data_ptr(), x.shape[0], out.data_ptr(),
                                                        torch.cuda.current_stream(x.device).cuda_stream))
            return out.bool()
        xyz = np.ascontiguousarray(np.asarray(xyz, dtype=np.float32).reshape(-1, 3))
        out = np.zeros(xyz.shape[0], np.uint8)
        _lib.check(self.L.tsl_tsdf_query_points(self.h, mode, int(param), _vp(xyz), xyz.shape[0], _vp(out)))
        return out.astype(bool)

    def is_pos_occupy(self, xyz):
        return self._query_points(0, xyz)

    def is_pos_unobserved(self, xyz):
        return self._query_points(1, xyz)

    def is_near_pos_occupy(self, xyz, voxel):
        return self._query_points(2, xyz, voxel)

    def raycast(self, pos, dir, max_dist):
        """Batched BaseMap.raycast: returns (hit bool[n], end xyz f32[n,3], length f32[n]); torch CUDA tensors in -> torch CUDA tensors
        out, asynchronously (the planner's 64-128 rays per node expansion, topo_graph.py:444-507, without a host round trip)."""
        if _is_device_tensor(pos):
            torch = _torch()
            p = pos.reshape(-1, 3).contiguous().float(); d = dir.reshape(-1, 3).contiguous().float()
            assert p.shape == d.shape and d.is_cuda
            n = p.shape[0]
            hit = torch.empty(n, dtype=torch.uint8, device=p.device); end = torch.empty((n, 3), dtype=torch.float32, device=p.device)
            ln = torch.empty(n, dtype=torch.float32, device=p.device)
            _lib.check(self.L.tsl_tsdf_query_raycast_dev(self.h, p.data_ptr(), d.data_ptr(), float(max_dist), n, hit.data_ptr(), end.data_ptr(),
                                                         ln.data_ptr(), torch.cuda.current_stream(p.device).cuda_stream))
            return hit.bool(), end, ln
        pos = np.ascontiguousarray(np.asarray(pos, dtype=np.float32).reshape(-1, 3))
        dir = np.ascontiguousarray(np.asarray(dir, dtype=np.float32).reshape(-1, 3))
        assert pos.shape == dir.shape
        n = pos.shape[0]
        hit = np.zeros(n, np.uint8); end = np.zeros((n, 3), np.float32); ln = np.zeros(n, np.float32)
        _lib.check(self.L.tsl_tsdf_query_raycast(self.h, _vp(pos), _vp(dir), float(max_dist), n, _vp(hit), _vp(end), _vp(ln)))
        return hit.astype(bool), end, ln

    # ---- ESDF (definition from the legacy dense_esdf.py:228-333) -----------------------------------
    def update_esdf(self, gamma=None, max_dist=None, wait=True):
        """Bring the ESDF of the active submap up to date with its TSDF -- incrementally: only the bricks integrated into since
        the previous call (dilated by max_dist) are recomputed, with the result of a full recompute.  Returns the number of brick
        relaxations; `esdf_stats()` has the breakdown.  The reference's hook ran after every frame (dense_esdf.py:400-402); for that use
        pass wait=False: the update is only enqueued behind the frame (returns None) and `export_esdf` / `esdf_stats` / `esdf_totals`
        wait for whatever is still in flight."""
        g = self.voxel_scale if gamma is None else gamma                 # dense_esdf.py:40 gamma = voxel_scale
        md = self.max_ray_length if max_dist is None else max_dist       # dense_esdf.py:265 sign * max_ray_length
        self._esdf_ever = True
        self._esdf_gamma, self._esdf_max_dist = g, md
        if not wait:
            _lib.check(self.L.tsl_esdf_update(self.h, float(g), float(md), None))
            return None
        it = C.c_int32()
        _lib.check(self.L.tsl_esdf_update(self.h, float(g), float(md), C.byref(it)))
        return it.value

    def esdf_totals(self):
        """Sums over the completed ESDF updates of this map (waits for the ones in flight)."""
        st = _lib.EsdfTotals()
        _lib.check(self.L.tsl_esdf_totals(self.h, C.byref(st)))
        return st.as_dict()

    def esdf_stats(self):
        st = _lib.EsdfStats()
        _lib.check(self.L.tsl_esdf_last_stats(self.h, C.byref(st)))
        return st.as_dict()

    def cvt_ESDF_to_voxels_slice(self, z):
        """dense_esdf.py:498-509: ESDF values of the voxel layer at height z -> export_ESDF / export_ESDF_xyz (device-resident;
        `.to_numpy()` / `.to_torch()`), count in num_export_ESDF_particles[None].  The ESDF is brought up to date first -- the reference updates it
        inside every recast (dense_esdf.py:400-402), so its slice is always current: an incremental update is queued here (it finds nothing to do when
        no frame was integrated since the last one; tsl_esdf_slice waits for it), with the parameters of the last update."""
        self.update_esdf(gamma=getattr(self, "_esdf_gamma", None), max_dist=getattr(self, "_esdf_max_dist", None), wait=False)
        n = C.c_int32()
        _lib.check(self.L.tsl_esdf_slice(self.h, float(z), C.byref(n)))
        self._esdf_slice_n = n.value

    def get_voxels_ESDF_slice(self, z):
        self.cvt_ESDF_to_voxels_slice(z)
        return self._read_esdf_slice(self._esdf_slice_n)

    def export_esdf_torch(self):
        """(indices int16[n,3], esdf f32[n]) as torch tensors on the device: zero-copy views of the map's staging buffer, valid until the
        next exporting call on this map (clone them to keep them)."""
        from .fields import device_view
        n = self.count_active()
        pi, pv, cnt = C.c_void_p(), C.c_void_p(), C.c_int64()
        _lib.check(self.L.tsl_esdf_export_dev(self.h, n, C.byref(pi), C.byref(pv), C.byref(cnt)))
        k = min(n, cnt.value)
        return device_view(pi.value, (k, 3), "<i2", self, self.device), device_view(pv.value, (k,), "<f4", self, self.device)

    def export_esdf(self):
        """(indices int16[n,3], esdf f32[n]) for every observed voxel of the active submap."""
        n = self.count_active()
        idx = np.zeros((n, 3), np.int16)
        val = np.zeros(n, np.float32)
        cnt = C.c_int64()
        _lib.check(self.L.tsl_esdf_export(self.h, _vp(idx), _vp(val), n, C.byref(cnt)))
        return idx[:cnt.value], val[:cnt.value]

    def init_sphere(self, voxels=30, radius=None):
        """dense_tsdf.py:136-146 as intended by tests/marching_cube_test.py: an analytic sphere SDF of
        `voxels`^3 cells centred on the map origin, every cell observed (the reference body is broken at HEAD:
        3-index access on 4-d fields)."""
        radius = self.voxel_scale * 3 if radius is None else radius
        r = np.arange(-(voxels // 2), voxels - voxels // 2, dtype=np.int16)
        ii, jj, kk = np.meshgrid(r, r, r, indexing="ij")
        idx = np.stack([ii, jj, kk], -1).reshape(-1, 3)
        p = idx.astype(np.float32) * np.float32(self.voxel_scale)
        tsdf = (np.sqrt((p * p).sum(1)) - np.float32(radius)).astype(np.float16)
        n = idx.shape[0]
        col = np.zeros((n, 3), np.float16) if self.enable_texture else np.array([])
        self.load_numpy(self.get_active_submap_id(), idx, tsdf, np.ones(n, np.float16), np.zeros(n, np.int8), col)
