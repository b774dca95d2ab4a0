"""taichislam_amd -- MI355X-native dense-mapping backend with TaichiSLAM's mapping API.

`taichislam_amd.mapping` mirrors `taichi_slam.mapping` (DenseTSDF, Octomap, MarchingCubeMesher, SubmapMapping);
the compute is hand-written HIP for gfx950 behind the C-ABI declared in include/taichislam_hip.h."""
__version__ = "0.1.0"
