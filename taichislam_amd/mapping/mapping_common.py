"""BaseMap: pose bookkeeping shared by DenseTSDF and Octomap.

Mirrors taichi_slam/mapping/mapping_common.py:10-156 of the reference (same attribute and method names) on
top of the C-ABI handle; all per-voxel math runs in the HIP kernels."""
import ctypes as C

import numpy as np

from .. import _lib
from .fields import ScalarField


def _dptr(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    if a.size != n:
        raise ValueError(f"expected {n} values, got {a.size}")
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def jet_colormap():
    """float32[1024,3] equal to [matplotlib.cm.jet(i/1024.0)[:3] for i in range(1024)] (mapping_common.py:158-163):
    a 256-entry piecewise-linear LUT indexed by int(x*256)."""
    r = np.array([[0, 0], [0.35, 0], [0.66, 1], [0.89, 1], [1, 0.5]])
    g = np.array([[0, 0], [0.125, 0], [0.375, 1], [0.64, 1], [0.91, 0], [1, 0]])
    b = np.array([[0, 0.5], [0.11, 1], [0.34, 1], [0.65, 0], [1, 0]])
    k = np.minimum((np.arange(1024) / 1024.0 * 256.0).astype(np.int64), 255)
    x = k / 255.0
    return np.stack([np.interp(x, s[:, 0], s[:, 1]) for s in (r, g, b)], axis=1).astype(np.float32)


class BaseMap:
    _prefix = "tsl_tsdf"

    def __init__(self, voxel_scale):
        self.L = _lib.lib()
        self.h = None
        self.base_T_np = np.zeros(3)
        self.base_R_np = np.eye(3)
        self.frame_id = 0
        self.submap_enabled = False
        self.voxel_scale = voxel_scale
        self.voxel_scale_ = np.array([voxel_scale] * 3, dtype=np.float32)
        self.K_cam_dep = None
        self.K_cam_color = None

    # ---- C-ABI plumbing -------------------------------------------------------------------------
    def _fn(self, name):
        return getattr(self.L, f"{self._prefix}_{name}")

    def _call(self, name, *args):
        _lib.check(self._fn(name)(self.h, *args))

    def __del__(self):
        try:
            if self.h is not None:
                self._fn("destroy")(self.h)
                self.h = None
        except Exception:
            pass

    def sync(self):
        self._call("sync")

    # ---- camera (mapping_common.py:25-29) -----------------------------------------------------------
    def _push_intrinsics(self):
        kd = _dptr(self.K_cam_dep, 9)[1] if self.K_cam_dep is not None else None
        kc = _dptr(self.K_cam_color, 9)[1] if self.K_cam_color is not None else None
        self._call("set_intrinsics", kd, kc)

    def set_dep_camera_intrinsic(self, K):
        self.K_cam_dep = np.asarray(K, dtype=np.float64).reshape(-1)
        self._push_intrinsics()

    def set_color_camera_intrinsic(self, K):
        self.K_cam_color = np.asarray(K, dtype=np.float64).reshape(-1)
        self._push_intrinsics()

    # ---- submap pose tables (mapping_common.py:102-147) ------------------------------------------------
    def initialize_submap_fields(self, max_submap_num):
        self.submap_enabled = True
        # DEVIATION: identity instead of the reference's all-zero matrices (mapping_common.py:106)
        self.submaps_base_R_np = np.tile(np.eye(3), (max_submap_num, 1, 1))
        self.submaps_base_T_np = np.zeros((max_submap_num, 3))
        self._remote_submap_num = 0
        self.active_submap_id = ScalarField(self.get_active_submap_id, self._set_active_submap, "active_submap_id")
        self.remote_submap_num = ScalarField(lambda: self._remote_submap_num, self._set_remote_num, "remote_submap_num")

    def _set_remote_num(self, v):
        self._remote_submap_num = int(v)

    def _set_active_submap(self, v):
        self._call("set_active_submap", int(v))

    def get_active_submap_id(self):
        v = C.c_int32()
        self._call("get_active_submap", C.byref(v))
        return v.value

    def finalization_current_submap(self):
        pass

    def switch_to_next_submap(self):
        self.finalization_current_submap()
        nxt = self.get_active_submap_id() + 1
        self._set_active_submap(nxt)
        return nxt

    def set_base_pose_submap(self, submap_id, _R, _T):
        self.submaps_base_T_np[submap_id] = _T
        self.submaps_base_R_np[submap_id] = _R
        self._call("set_base_pose_submap", int(submap_id), _dptr(_R, 9)[1], _dptr(_T, 3)[1])

    def set_base_pose(self, _R, _T):
        self.base_T_np = np.asarray(_T, dtype=np.float64)
        self.base_R_np = np.asarray(_R, dtype=np.float64)
        if hasattr(self.L, f"{self._prefix}_set_base_pose"):
            self._call("set_base_pose", _dptr(_R, 9)[1], _dptr(_T, 3)[1])

    def convert_by_base(self, R, T):
        """mapping_common.py:91-100 (host-side copy for callers; the kernels get the same conversion from the C-ABI)."""
        if self.submap_enabled:
            sid = self.get_active_submap_id()
            base_R_inv = self.submaps_base_R_np[sid].T
            return base_R_inv @ R, base_R_inv @ (T - self.submaps_base_T_np[sid])
        base_R_inv = self.base_R_np.T
        return base_R_inv @ R, base_R_inv @ (T - self.base_T_np)

    def set_pose(self, _R, _T):
        """mapping_common.py:149-156.  Kept for API parity: stores the pose used by the next recast call."""
        self._pending_pose = (np.asarray(_R, dtype=np.float64), np.asarray(_T, dtype=np.float64))
