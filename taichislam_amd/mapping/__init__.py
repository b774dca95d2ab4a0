"""taichislam_amd.mapping -- drop-in for taichi_slam.mapping (reference taichi_slam/mapping/__init__.py:1-6)."""
from .mapping_common import BaseMap  # noqa: F401
from .dense_tsdf import DenseTSDF  # noqa: F401
from .taichi_octomap import Octomap  # noqa: F401
from .marching_cube_mesher import MarchingCubeMesher  # noqa: F401
from .submap_mapping import SubmapMapping  # noqa: F401
