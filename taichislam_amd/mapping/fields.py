"""Field-like adapters: the reference exposes Taichi fields (`x[None]`, `x[None] = v`, `.to_numpy()`,
`.from_numpy()`) on its mapping classes and callers rely on them (scripts/taichislam_node.py:307,331,
342-351; taichi_slam/mapping/submap_mapping.py:100-107).  These thin objects keep that surface while the
data lives in device buffers owned by the C-ABI handle."""
import numpy as np


class ScalarField:
    """0-d field backed by getter/setter callables (`active_submap_id[None]`, `num_TSDF_particles[None]`)."""

    def __init__(self, getter, setter=None, name="scalar"):
        self._get, self._set, self._name = getter, setter, name

    def __getitem__(self, key):
        return self._get()

    def __setitem__(self, key, value):
        if self._set is None:
            raise TypeError(f"{self._name} is read-only")
        self._set(int(value))

    def to_numpy(self):
        return np.asarray(self._get())

    def __repr__(self):
        return f"<{self._name}={self._get()}>"


class _DevicePointer:
    """A device buffer of the C-ABI handle presented through `__cuda_array_interface__` (version 2), which torch-ROCm (and cupy /
    numba) accept without copying.  `keep` pins the owning map for as long as a view made from this object lives."""

    def __init__(self, ptr, shape, typestr, keep):
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}
        self._keep = keep


def device_view(ptr, shape, typestr="<f4", keep=None, device=0):
    """torch tensor over `shape` elements at device pointer `ptr` (zero-copy; the buffer belongs to the handle `keep`)."""
    import torch
    empty = any(int(x) == 0 for x in shape)
    if not ptr and not empty:       # e.g. mesh_colors / export_color of an untextured map: there is no such buffer (an uninitialised tensor would pass for one)
        raise ValueError("device_view: the handle has no buffer for this field (texture disabled?)")
    if empty:
        dt = {"<f4": torch.float32, "<i2": torch.int16, "|u1": torch.uint8}[typestr]
        return torch.empty(tuple(int(x) for x in shape), dtype=dt, device=f"cuda:{device}")
    return torch.as_tensor(_DevicePointer(ptr, shape, typestr, keep), device=f"cuda:{device}")


class DeviceArrayField:
    """1-d vector field living in a device export buffer; `.to_numpy()` copies `rows` rows back, `.to_torch()` is a zero-copy
    view of the device buffer (include/taichislam_hip.h: the *_dev entry points)."""

    def __init__(self, owner, reader, rows, width, name, writer=None, dev=None):
        self._owner, self._reader, self._rows, self._width, self._name, self._writer = owner, reader, rows, width, name, writer
        self._dev = dev                      # () -> (device pointer, rows valid) of the buffer

    def to_torch(self, n=None):
        """The first `n` rows (default: the rows the last export / mesh call produced) as a torch tensor ON THE DEVICE, without a copy.
        The view stays valid as long as the owning map lives; a later export overwrites its contents."""
        if self._dev is None:
            raise TypeError(f"{self._name} has no device view")
        ptr, valid = self._dev()
        n = int(valid if n is None else min(int(n), self._rows))
        shape = (n,) if self._width == 1 else (n, self._width)
        return device_view(ptr, shape, "<f4", keep=self._owner, device=getattr(self._owner, "device", 0))

    @property
    def shape(self):
        return (self._rows,) if self._width == 1 else (self._rows, self._width)

    def to_numpy(self, n=None):
        return self._reader(self._rows if n is None else int(n))

    def __setitem__(self, row, value):
        """`field[i] = vector` (tests/gen_topo_graph.py:64-65 appends a marker point to export_TSDF_xyz this way)."""
        if self._writer is None:
            raise TypeError(f"{self._name} cannot be written")
        self._writer(int(row), np.asarray(value, dtype=np.float32).reshape(-1))

    def __repr__(self):
        return f"<device field {self._name} {self.shape}>"


class MapFieldRef:
    """Opaque reference to a per-voxel map field (TSDF, W_TSDF, TSDF_observed, occupy, color).  The
    reference hands these between maps (dense_tsdf.py:315-317) and to the mesher
    (marching_cube_mesher.py:193); here they only identify the owning map."""

    def __init__(self, owner, name):
        self.owner, self.name = owner, name

    def __repr__(self):
        return f"<map field {self.name} of {type(self.owner).__name__}>"
