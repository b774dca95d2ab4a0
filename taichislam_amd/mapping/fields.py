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


class DeviceArrayField:
    """1-d vector field living in a device export buffer; `.to_numpy()` copies `rows` rows back."""

    def __init__(self, owner, reader, rows, width, name, writer=None):
        self._owner, self._reader, self._rows, self._width, self._name, self._writer = owner, reader, rows, width, name, writer

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
