// tsl_todo.hip -- entry points declared in include/taichislam_hip.h that are not implemented yet.
// They fail loudly (TSL_ERR_ARG + message); nothing here computes on the CPU.
#include "tsl_common.hpp"
#define TSL_TODO(name) { tsl::set_error(name ": not implemented yet"); return TSL_ERR_ARG; }
extern "C" {
int  tsl_esdf_update(tsl_tsdf*, float, float, int32_t*) TSL_TODO("tsl_esdf_update")
int  tsl_esdf_export(tsl_tsdf*, int16_t*, float*, int64_t, int64_t*) TSL_TODO("tsl_esdf_export")
int  tsl_octo_create(const tsl_octo_cfg*, int, tsl_octo**) TSL_TODO("tsl_octo_create")
void tsl_octo_destroy(tsl_octo*) {}
int  tsl_octo_get_dims(const tsl_octo*, int32_t*, int32_t*, int32_t*, int32_t*, double*) TSL_TODO("tsl_octo_get_dims")
int  tsl_octo_sync(tsl_octo*) TSL_TODO("tsl_octo_sync")
int  tsl_octo_reset(tsl_octo*) TSL_TODO("tsl_octo_reset")
int  tsl_octo_set_intrinsics(tsl_octo*, const double*, const double*) TSL_TODO("tsl_octo_set_intrinsics")
int  tsl_octo_set_base_pose_submap(tsl_octo*, int, const double*, const double*) TSL_TODO("tsl_octo_set_base_pose_submap")
int  tsl_octo_get_active_submap(const tsl_octo*, int32_t*) TSL_TODO("tsl_octo_get_active_submap")
int  tsl_octo_set_active_submap(tsl_octo*, int32_t) TSL_TODO("tsl_octo_set_active_submap")
int  tsl_octo_integrate_depth(tsl_octo*, const double*, const double*, const uint16_t*, int, int, const uint8_t*, int, int) TSL_TODO("tsl_octo_integrate_depth")
int  tsl_octo_integrate_depth_dev(tsl_octo*, const double*, const double*, const void*, int, int, const void*, int, int) TSL_TODO("tsl_octo_integrate_depth_dev")
int  tsl_octo_integrate_points(tsl_octo*, const double*, const double*, const float*, const uint8_t*, int64_t) TSL_TODO("tsl_octo_integrate_points")
int  tsl_octo_last_frame_stats(tsl_octo*, tsl_frame_stats*) TSL_TODO("tsl_octo_last_frame_stats")
int  tsl_octo_export_leaves(tsl_octo*, int32_t*, float*, int64_t, int64_t*) TSL_TODO("tsl_octo_export_leaves")
int  tsl_octo_occupied_voxels(tsl_octo*, tsl_octo*, int, int, int32_t*) TSL_TODO("tsl_octo_occupied_voxels")
int  tsl_octo_read_exports(tsl_octo*, float*, float*, int64_t) TSL_TODO("tsl_octo_read_exports")
int  tsl_octo_num_particles(tsl_octo*, int32_t*) TSL_TODO("tsl_octo_num_particles")
int  tsl_octo_fuse_submaps(tsl_octo*, tsl_octo*) TSL_TODO("tsl_octo_fuse_submaps")
}
