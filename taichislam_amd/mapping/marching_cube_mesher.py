"""MarchingCubeMesher: drop-in for taichi_slam.mapping.MarchingCubeMesher (reference
taichi_slam/mapping/marching_cube_mesher.py:13-193).  The mesh stays in device buffers owned by the map handle;
`mesh_vertices/mesh_normals/mesh_colors` are field-likes with `.to_numpy()`."""
import ctypes as C

import numpy as np

from .. import _lib
from .fields import DeviceArrayField, ScalarField

EPS = 1e-6


class MarchingCubeMesher:
    def __init__(self, mapping, max_triangles=1000000, tsdf_surface_thres=0.1):
        self.max_triangles = int(max_triangles)
        self.mapping = mapping
        self.enable_texture = mapping.enable_texture
        self.tsdf_surface_thres = tsdf_surface_thres
        self._n_tri = 0
        self.device = getattr(mapping, "device", 0)
        self.mesh_vertices = DeviceArrayField(self, lambda n: self._read(n)[0], self.max_triangles * 3, 3, "mesh_vertices", dev=lambda: self._dev(0))
        self.mesh_normals = DeviceArrayField(self, lambda n: self._read(n)[1], self.max_triangles * 3, 3, "mesh_normals", dev=lambda: self._dev(1))
        self.mesh_colors = DeviceArrayField(self, lambda n: self._read(n)[2], self.max_triangles * 3, 3, "mesh_colors", dev=lambda: self._dev(2))
        self.mesh_indices = None
        self.num_facelets = ScalarField(lambda: self._n_tri, None, "num_facelets")
        # the reference never updates num_vertices although the node uses it as the mesh size
        # (marching_cube_mesher.py:22, scripts/taichislam_node.py:342, Q14); here it is kept equal to 3*num_facelets
        self.num_vertices = ScalarField(lambda: 3 * min(self._n_tri, self.max_triangles), None, "num_vertices")

    def _read(self, n):
        n = int(max(0, min(n, 3 * min(self._n_tri, self.max_triangles))))
        v = np.empty((n, 3), np.float32)
        nr = np.empty((n, 3), np.float32)
        col = np.full((n, 3), 0.5, np.float32)
        _lib.check(_lib.lib().tsl_mesh_read(self.mapping.h, v.ctypes.data_as(C.c_void_p), nr.ctypes.data_as(C.c_void_p),
                                            col.ctypes.data_as(C.c_void_p) if self.enable_texture else None, n))
        return v, nr, col

    def _dev(self, which):
        """(device pointer of mesh_vertices / mesh_normals / mesh_colors, vertices of the last generate_mesh): `.to_torch()` of the fields"""
        p = [C.c_void_p(), C.c_void_p(), C.c_void_p()]
        n = C.c_int32()
        _lib.check(_lib.lib().tsl_mesh_buffers_dev(self.mapping.h, C.byref(p[0]), C.byref(p[1]), C.byref(p[2]), C.byref(n)))
        return p[which].value, 3 * min(n.value, self.max_triangles)

    def vertice_num(self):
        return self.num_facelets[None] * 3

    def generate_mesh(self, step=1):
        n = C.c_int32()
        _lib.check(_lib.lib().tsl_mesh_generate(self.mapping.h, int(step), float(self.tsdf_surface_thres), self.max_triangles, C.byref(n)))
        self._n_tri = n.value
        print("Total triangles", self._n_tri)

    def get_mesh(self):
        """(vertices[3n,3], normals[3n,3], colors|None) of the last generate_mesh (north-star alias)."""
        v, nr, c = self._read(3 * self._n_tri)
        return v, nr, (c if self.enable_texture else None)
