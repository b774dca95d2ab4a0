// tsl_todo.hip -- entry points declared in include/taichislam_hip.h that are not implemented yet.
// They fail loudly (TSL_ERR_ARG + message); nothing here computes on the CPU.
#include "tsl_common.hpp"
#define TSL_TODO(name) { tsl::set_error(name ": not implemented yet"); return TSL_ERR_ARG; }
extern "C" {
int  tsl_esdf_update(tsl_tsdf*, float, float, int32_t*) TSL_TODO("tsl_esdf_update")
int  tsl_esdf_export(tsl_tsdf*, int16_t*, float*, int64_t, int64_t*) TSL_TODO("tsl_esdf_export")
}
