// tsl_query.hip -- batched map queries used by planners on top of the TSDF map.  Replaces BaseMap.raycast /
// is_pos_occupy / is_near_pos_occupy / is_pos_unobserved (taichi_slam/mapping/mapping_common.py:165-204) and
// DenseTSDF.is_occupy / is_unobserved (dense_tsdf.py:148-155), reference root; TopoGraphGen calls them 64-128 rays at a
// time (topo_graph.py:444-507).  One thread per query, read-only gathers through the brick table.
#include "tsl_tsdf.hpp"

namespace tsl {

__device__ __forceinline__ void q_read(const MapDev& M, int s, int i, int j, int k, float* tsdf, int* obs)
{
    *tsdf = 0.0f; *obs = 0;                                   // outside the volume / inactive cell reads as 0 (A7)
    if (!in_volume(M, i, j, k)) return;
    int l; const int b = brick_of(M, i, j, k, &l);
    const int p = pool_lookup_ro(M, s, b);
    if (p < 0) return;
    const size_t v = (size_t)p * TSL_BRK3 + l;
    *tsdf = h2f((h16)(M.tw[v] & 0xffffu)); *obs = M.obs[v];
}
// is_occupy  dense_tsdf.py:153-155 (an unobserved voxel reads TSDF = 0 and therefore counts as occupied)
__device__ __forceinline__ bool q_occupied(const MapDev& M, int s, int i, int j, int k, float thres)
{ float t; int o; q_read(M, s, i, j, k, &t, &o); return t < thres; }

// mode 0: is_pos_occupy  1: is_pos_unobserved  2: is_near_pos_occupy(param)
__global__ void __launch_bounds__(256) k_query_points(MapDev M, int s, float vs, float thres, int mode, int param, const float* __restrict__ xyz, long long n, uint8_t* out)
{
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const int i = rnd_i(xyz[q * 3] / vs), j = rnd_i(xyz[q * 3 + 1] / vs), k = rnd_i(xyz[q * 3 + 2] / vs);       // mapping_common.py:258-266
    bool r = false;
    if (mode == 0) r = q_occupied(M, s, i, j, k, thres);                                                     // :187-191
    else if (mode == 1) { float t; int o; q_read(M, s, i, j, k, &t, &o); r = o == 0; }                       // :181-185
    else { for (int a = -param; a < param; ++a) for (int b = -param; b < param; ++b) for (int c = -param; c < param; ++c) r |= q_occupied(M, s, i + a, j + b, k + c, thres); }   // :193-204
    out[q] = r ? 1 : 0;
}

// raycast  mapping_common.py:165-178
__global__ void __launch_bounds__(256) k_query_raycast(MapDev M, int s, float vs_f, float vs_len, float thres, float max_dist, const float* __restrict__ pos,
                                                       const float* __restrict__ dir, long long n, uint8_t* hit, float* end_xyz, float* len)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int steps = (int)(max_dist / vs_f);                                                                // :167, range() truncates
    float x[3] = { 0.0f, 0.0f, 0.0f }, l = 0.0f; bool succ = false;
    for (int jj = 0; jj < steps; ++jj) {
        l = (float)jj * vs_len;                                                                              // :173
        for (int a = 0; a < 3; ++a) x[a] = dir[q * 3 + a] * l + pos[q * 3 + a];                              // :174
        if (q_occupied(M, s, rnd_i(x[0] / vs_f), rnd_i(x[1] / vs_f), rnd_i(x[2] / vs_f), thres)) { succ = true; break; }   // :175-177
    }
    hit[q] = succ ? 1 : 0;
    for (int a = 0; a < 3; ++a) end_xyz[q * 3 + a] = x[a];
    len[q] = l;
}

}  // namespace tsl

using namespace tsl;

extern "C" {

int tsl_tsdf_query_points(tsl_tsdf* m, int mode, int param, const float* xyz, int64_t n, uint8_t* out)
{
    TSL_REQUIRE(m && n >= 0 && (n == 0 || (xyz && out)), "query_points: bad argument"); TSL_REQUIRE(mode >= 0 && mode <= 2 && param >= 0 && param <= 16, "query_points: bad mode");
    if (n == 0) return TSL_OK;
    TSL_HIP(hipSetDevice(m->device));
    int rc = grow(&m->xbuf, &m->xbuf_bytes, (size_t)n * 13 + 64); if (rc) return rc;
    float* dx = (float*)m->xbuf; uint8_t* dout = (uint8_t*)m->xbuf + (size_t)n * 12;
    TSL_HIP(hipMemcpy(dx, xyz, (size_t)n * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_query_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ms(m), m->M, m->cfg.is_global_map ? 0 : m->active, m->P.vs, m->surf_thres, mode, param,
                       (const float*)dx, (long long)n, dout);
    TSL_HIP(hipGetLastError());
    TSL_HIP(hipStreamSynchronize(ms(m)));
    TSL_HIP(hipMemcpy(out, dout, (size_t)n, hipMemcpyDeviceToHost));
    return TSL_OK;
}

int tsl_tsdf_query_raycast(tsl_tsdf* m, const float* pos, const float* dir, float max_dist, int64_t n, uint8_t* hit, float* end_xyz, float* len)
{
    TSL_REQUIRE(m && n >= 0 && (n == 0 || (pos && dir && hit && end_xyz && len)), "query_raycast: bad argument");
    if (n == 0) return TSL_OK;
    TSL_HIP(hipSetDevice(m->device));
    const size_t c = (size_t)n;
    int rc = grow(&m->xbuf, &m->xbuf_bytes, c * 44 + 64); if (rc) return rc;
    float* dpos = (float*)m->xbuf; float* ddir = dpos + c * 3; float* dend = ddir + c * 3; float* dlen = dend + c * 3; uint8_t* dhit = (uint8_t*)(dlen + c);
    TSL_HIP(hipMemcpy(dpos, pos, c * 12, hipMemcpyHostToDevice));
    TSL_HIP(hipMemcpy(ddir, dir, c * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_query_raycast, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ms(m), m->M, m->cfg.is_global_map ? 0 : m->active, m->P.vs, (float)m->cfg.voxel_scale,
                       m->surf_thres, max_dist, (const float*)dpos, (const float*)ddir, (long long)n, dhit, dend, dlen);
    TSL_HIP(hipGetLastError());
    TSL_HIP(hipStreamSynchronize(ms(m)));
    TSL_HIP(hipMemcpy(hit, dhit, c, hipMemcpyDeviceToHost));
    TSL_HIP(hipMemcpy(end_xyz, dend, c * 12, hipMemcpyDeviceToHost));
    TSL_HIP(hipMemcpy(len, dlen, c * 4, hipMemcpyDeviceToHost));
    return TSL_OK;
}

// ---- device-buffer forms: nothing is staged, nothing is waited for.  The queries are launched on the handle's stream -- behind every frame
// queued so far -- after what `user_stream` (a hipStream_t, e.g. torch's current stream; NULL = the legacy default stream, which is what
// torch uses unless told otherwise) has queued, and `user_stream` is made to wait for them: a planner that expands a node with 64-128
// rays (topo_graph.py:444-507) pays two event operations and one launch, no host round trip.
static int order_before(tsl_tsdf* m, hipStream_t user, hipStream_t q)
{
    if (user == q) return TSL_OK;
    if (!m->in_ev[0]) for (auto& e : m->in_ev) TSL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t e = m->in_ev[m->in_ev_next]; m->in_ev_next = (m->in_ev_next + 1) % 8;
    TSL_HIP(hipEventRecord(e, user));
    TSL_HIP(hipStreamWaitEvent(q, e, 0));
    return TSL_OK;
}
static int order_after(tsl_tsdf* m, hipStream_t user, hipStream_t q)
{
    if (user == q) return TSL_OK;
    hipEvent_t e = m->in_ev[m->in_ev_next]; m->in_ev_next = (m->in_ev_next + 1) % 8;
    TSL_HIP(hipEventRecord(e, q));
    TSL_HIP(hipStreamWaitEvent(user, e, 0));
    return TSL_OK;
}

int tsl_tsdf_query_points_dev(tsl_tsdf* m, int mode, int param, const void* xyz_dev, int64_t n, void* out_dev, void* user_stream)
{
    TSL_REQUIRE(m && n >= 0 && (n == 0 || (xyz_dev && out_dev)), "query_points_dev: bad argument"); TSL_REQUIRE(mode >= 0 && mode <= 2 && param >= 0 && param <= 16, "query_points_dev: bad mode");
    if (n == 0) return TSL_OK;
    TSL_HIP(hipSetDevice(m->device));
    hipStream_t q = ms(m);
    int rc = order_before(m, (hipStream_t)user_stream, q); if (rc) return rc;
    hipLaunchKernelGGL(k_query_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, q, m->M, m->cfg.is_global_map ? 0 : m->active, m->P.vs, m->surf_thres, mode, param,
                       (const float*)xyz_dev, (long long)n, (uint8_t*)out_dev);
    TSL_HIP(hipGetLastError());
    return order_after(m, (hipStream_t)user_stream, q);
}

int tsl_tsdf_query_raycast_dev(tsl_tsdf* m, const void* pos_dev, const void* dir_dev, float max_dist, int64_t n, void* hit_dev, void* end_xyz_dev, void* len_dev, void* user_stream)
{
    TSL_REQUIRE(m && n >= 0 && (n == 0 || (pos_dev && dir_dev && hit_dev && end_xyz_dev && len_dev)), "query_raycast_dev: bad argument");
    if (n == 0) return TSL_OK;
    TSL_HIP(hipSetDevice(m->device));
    hipStream_t q = ms(m);
    int rc = order_before(m, (hipStream_t)user_stream, q); if (rc) return rc;
    // rays are short serial chains (max_dist / voxel steps of dependent gathers): 64 threads per workgroup spread a 128-ray batch over two CUs
    hipLaunchKernelGGL(k_query_raycast, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, q, m->M, m->cfg.is_global_map ? 0 : m->active, m->P.vs, (float)m->cfg.voxel_scale,
                       m->surf_thres, max_dist, (const float*)pos_dev, (const float*)dir_dev, (long long)n, (uint8_t*)hit_dev, (float*)end_xyz_dev, (float*)len_dev);
    TSL_HIP(hipGetLastError());
    return order_after(m, (hipStream_t)user_stream, q);
}

}  // extern "C"
