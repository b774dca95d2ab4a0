"""Octomap: drop-in for taichi_slam.mapping.Octomap (reference taichi_slam/mapping/taichi_octomap.py:12-211): an f32
hit counter per leaf voxel with a `> min_occupy_thres` occupancy test, on HIP kernels through the C-ABI."""
import ctypes as C
import math
import time

import numpy as np

from .. import _lib
from .fields import DeviceArrayField, MapFieldRef, ScalarField
from .mapping_common import BaseMap, _dptr


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Octomap(BaseMap):
    _prefix = "tsl_octo"

    def __init__(self, map_scale=[10, 10], voxel_scale=0.05, min_occupy_thres=3, texture_enabled=False,
                 min_ray_length=0.3, max_ray_length=3.0, max_disp_particles=1000000, K=2,
                 max_submap_num=1024, disp_ceiling=10.0, disp_floor=-10.0,
                 is_global_map=False, recast_step=2, color_same_proj=True, device=0, max_bricks=0, max_points=0):
        super(Octomap, self).__init__(voxel_scale)
        Rxy = math.ceil(math.log2(map_scale[0] / voxel_scale) / math.log2(K))
        Rz = math.ceil(math.log2(map_scale[1] / voxel_scale) / math.log2(K))
        self.Rxy, self.Rz = Rxy, Rz
        self.map_size_xy, self.map_size_z = map_scale[0], map_scale[1]
        self.K = K
        self.N = self.K ** self.Rxy
        self.Nz = self.K ** self.Rz
        self.voxel_scale = self.map_size_xy / self.N          # recomputed as in the reference (:28); index math keeps voxel_scale_ (Q15)
        self.max_disp_particles = max_disp_particles
        self.min_occupy_thres = min_occupy_thres
        self.max_ray_length, self.min_ray_length = max_ray_length, min_ray_length
        self.enable_texture = texture_enabled
        self.max_submap_num = max_submap_num
        self.disp_ceiling, self.disp_floor = disp_ceiling, disp_floor
        self.is_global_map = is_global_map
        self.recast_step = recast_step
        self.color_same_proj = color_same_proj
        cfg = _lib.OctoCfg(float(map_scale[0]), float(map_scale[1]), float(voxel_scale), float(min_occupy_thres),
                           int(bool(texture_enabled)), float(min_ray_length), float(max_ray_length), int(K),
                           int(max_submap_num), float(disp_ceiling), float(disp_floor), int(bool(is_global_map)),
                           int(recast_step), int(bool(color_same_proj)), int(max_disp_particles), int(max_bricks), int(max_points))
        h = C.c_void_p()
        _lib.check(self.L.tsl_octo_create(C.byref(cfg), int(device), C.byref(h)))
        self.h = h
        self._queued = []
        self._integrate_depth_dev = _lib.octo_integrate_depth_dev_fn()
        self.num_export_particles = ScalarField(self._get_num, None, "num_export_particles")
        self.device = device
        self.export_x = DeviceArrayField(self, lambda n: self._read(n)[0], max_disp_particles, 3, "export_x", dev=lambda: self._exports_dev(0))
        self.export_color = DeviceArrayField(self, lambda n: self._read(n)[1], max_disp_particles, 3, "export_color", dev=lambda: self._exports_dev(1))
        self.occupy = MapFieldRef(self, "occupy")
        self.color = MapFieldRef(self, "color") if texture_enabled else None
        self.initialize_submap_fields(self.max_submap_num)
        print(f'The map voxel is:[{self.max_submap_num}x{self.N}x{self.N}x{self.Nz}] voxel scale {self.voxel_scale:3.3f}^3 '
              f'map scale:[{self.map_size_xy}mx{self.map_size_xy}mx{self.map_size_z}m] tree depth [{self.Rxy}, {self.Rz}]')

    def _exports_dev(self, which):
        p = [C.c_void_p(), C.c_void_p()]
        n = C.c_int32()
        self._call("exports_dev", C.byref(p[0]), C.byref(p[1]), C.byref(n))
        return p[which].value, max(0, min(n.value, self.max_disp_particles))

    def _get_num(self):
        v = C.c_int32()
        self._call("num_particles", C.byref(v))
        return v.value

    def _read(self, n):
        n = int(max(0, min(n, self.max_disp_particles)))
        xyz = np.empty((n, 3), np.float32)
        if self.enable_texture:
            rgb = np.empty((n, 3), np.float32)
            self._call("read_exports", _vp(xyz), _vp(rgb), n)
            return xyz, rgb
        self._call("read_exports", _vp(xyz), None, n)
        return xyz, np.full((n, 3), 0.5, np.float32)

    def last_frame_stats(self):
        st = _lib.FrameStats()
        self._call("last_frame_stats", C.byref(st))
        return st.as_dict()

    def is_occupy_count(self, count):
        return count > self.min_occupy_thres

    def recast_pcl_to_map(self, R, T, xyz_array, rgb_array=None, n=None):
        if hasattr(xyz_array, "data_ptr") and getattr(xyz_array, "is_cuda", False):
            import torch
            x = xyz_array.reshape(-1, 3).contiguous().float()
            if n is not None:
                x = x[:int(n)].contiguous()
            c = None
            if self.enable_texture and rgb_array is not None and hasattr(rgb_array, "data_ptr") and rgb_array.numel() >= 3 * x.shape[0] > 0:
                c = rgb_array.reshape(-1, 3)[:x.shape[0]].contiguous().to(torch.uint8)
            torch.cuda.current_stream(x.device).synchronize()          # the tensors are complete; the map's stream reads them
            self._call("integrate_points_dev", _dptr(R, 9)[1], _dptr(T, 3)[1], C.c_void_p(x.data_ptr()), C.c_void_p(c.data_ptr()) if c is not None else None, int(x.shape[0]))
            self._call("sync")                                         # ... and has read them before they can be freed
            return
        xyz = np.ascontiguousarray(np.asarray(xyz_array, dtype=np.float32).reshape(-1, 3))
        if n is not None:
            xyz = xyz[:int(n)]
        rgb = None
        if self.enable_texture and rgb_array is not None and np.size(rgb_array) >= 3 * xyz.shape[0] > 0:
            rgb = np.ascontiguousarray(np.asarray(rgb_array, dtype=np.uint8).reshape(-1, 3)[:xyz.shape[0]])     # BGR as delivered by OpenCV (:120)
        self._call("integrate_points", _dptr(R, 9)[1], _dptr(T, 3)[1], _vp(xyz), _vp(rgb) if rgb is not None else None, int(xyz.shape[0]))

    def recast_depth_to_map(self, R, T, depthmap, texture=None):
        use_tex = self.enable_texture and texture is not None and np.ndim(texture) == 3
        if not use_tex and hasattr(depthmap, "data_ptr") and depthmap.is_cuda:
            # the per-frame path of a device-resident stream: the library only QUEUES the frame (up to eight per launch), so the host side of this call is
            # the frame's cost -- pointers as plain integers, no numpy / ctypes temporaries for poses that already are float64 arrays
            Ra = R if (type(R) is np.ndarray and R.dtype == np.float64 and R.size == 9 and R.flags.c_contiguous) else _dptr(R, 9)[0]
            Ta = T if (type(T) is np.ndarray and T.dtype == np.float64 and T.size == 3 and T.flags.c_contiguous) else _dptr(T, 3)[0]
            q = self._queued
            q.append(depthmap)                                  # the tensor has to outlive the queue
            if len(q) > 16:
                del q[:8]
            shape = depthmap.shape
            rc = self._integrate_depth_dev(self.h, Ra.ctypes.data, Ta.ctypes.data, depthmap.data_ptr(), shape[0], shape[1], None, 0, 0)
            if rc:
                _lib.check(rc)
            return
        r, t = _dptr(R, 9)[1], _dptr(T, 3)[1]
        if hasattr(depthmap, "data_ptr") and getattr(depthmap, "is_cuda", False):
            if hasattr(texture, "data_ptr") and getattr(texture, "is_cuda", False):
                self._call("integrate_depth_dev", r, t, C.c_void_p(depthmap.data_ptr()), int(depthmap.shape[0]), int(depthmap.shape[1]),
                           C.c_void_p(texture.data_ptr()), int(texture.shape[0]), int(texture.shape[1]))
            else:
                self._queued.append(depthmap)
                self._call("integrate_depth_dev", r, t, C.c_void_p(depthmap.data_ptr()), int(depthmap.shape[0]), int(depthmap.shape[1]), None, 0, 0)
            return
        depth = np.ascontiguousarray(np.asarray(depthmap, dtype=np.uint16))
        if use_tex:
            tex = np.ascontiguousarray(np.asarray(texture, dtype=np.uint8))
            self._call("integrate_depth", r, t, _vp(depth), depth.shape[0], depth.shape[1], _vp(tex), tex.shape[0], tex.shape[1])
        else:
            self._call("integrate_depth", r, t, _vp(depth), depth.shape[0], depth.shape[1], None, 0, 0)

    def pointcloud2(self, n=None, has_rgb=None):
        """The first `n` occupied voxels of the last cvt_occupy_to_voxels call as a sensor_msgs/PointCloud2 payload (scripts/taichislam_node.py:330-333
        + utils/ros_pcl_transfer.py:96-136), interleaved on the device; see taichislam_amd.utils.ros_adapters."""
        from ..utils import ros_adapters
        n = self.num_export_particles[None] if n is None else int(n)
        n = max(0, min(n, self.max_disp_particles))
        has_rgb = self.enable_texture if has_rgb is None else bool(has_rgb)
        data = np.empty((n, 6 if has_rgb else 3), np.float32)
        self._call("pack_pointcloud2", int(has_rgb), n, _vp(data))
        return ros_adapters.pointcloud2_payload(data, has_rgb)

    def cvt_occupy_to_voxels(self, level=0):
        n = C.c_int32()
        self._call("occupied_voxels", None, int(level), 0, C.byref(n))

    def cvt_occupy_voxels_to(self, level, cur_num, max_disp_particles, x, color):
        n = C.c_int32()
        self._call("occupied_voxels", x._owner.h, int(level), 1, C.byref(n))

    def get_occupy_voxels(self, l):
        self.cvt_occupy_to_voxels(l)
        n = self.num_export_particles[None]
        return self._read(n)

    def export_leaves(self, with_color=False):
        """(indices int32[n,3], counts f32[n][, colours f32[n,3]]) of every touched leaf of the active submap (backend extra for tests)."""
        n = C.c_int64()
        self._call("export_leaves", None, None, None, 0, C.byref(n))
        idx = np.zeros((n.value, 3), np.int32)
        cnt = np.zeros(n.value, np.float32)
        rgb = np.zeros((n.value, 3), np.float32)
        self._call("export_leaves", _vp(idx), _vp(cnt), _vp(rgb), n.value, C.byref(n))
        return (idx, cnt, rgb) if with_color else (idx, cnt)

    def random_init_octo(self, pts, seed=0):
        """mapping_common.py:67-73 (demo helper): `pts` random leaves of the active submap receive a random hit count in 0..9."""
        rng = np.random.default_rng(seed)
        ijk = np.stack([rng.integers(0, self.N, pts), rng.integers(0, self.N, pts), rng.integers(0, self.Nz, pts)], 1)
        ijk -= np.array([self.N // 2, self.N // 2, self.Nz // 2])
        xyz = np.repeat(ijk, rng.integers(0, 10, pts), axis=0).astype(np.float32) * np.float32(self.voxel_scale_)
        sid = self.active_submap_id[None]
        self.recast_pcl_to_map(self.submaps_base_R_np[sid], self.submaps_base_T_np[sid], xyz, None)

    def fuse_submaps(self, submaps):
        t = time.time()
        _lib.check(self.L.tsl_octo_fuse_submaps(self.h, submaps.h))
        print(f"[OctoMap] Fuse submaps {(time.time() - t) * 1000:.1f}ms, active local: {submaps.active_submap_id[None]} "
              f"remote: {submaps.remote_submap_num[None]}")

    def reset(self):
        self._call("reset")

    def saveMap(self, path):
        pass

    def export_submap(self):
        return {}
