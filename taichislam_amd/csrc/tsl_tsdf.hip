// tsl_tsdf.hip -- DenseTSDF on MI355X (gfx950): depth/point-cloud integration, sparse export/import,
// surface/slice compaction.  Replaces taichi_slam/mapping/dense_tsdf.py:157-270,309-454 (reference root).
//
// Per-frame pipeline (all on the handle's stream, no host round trip between kernels):
//   K1 k_voxelize_depth / k_voxelize_points : pixel -> sensor-centred voxel (Morton key), f16 payload
//   K2 rocprim radix sort (stable)           : groups pixels per sensor voxel, raster order kept
//   K3 k_build_rays                           : per sensor voxel: raster-order f16 sums -> one ray record
//   K4 k_integrate<variant>                   : ray march, exact int64 fixed-point atomics into per-frame
//                                               brick scratch; bricks are allocated on first touch
//   K5 k_finalize                             : one weighted running-average update per touched voxel
#include "tsl_tsdf.hpp"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace tsl {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }

// ------------------------------------------------------------------------------------------------------
// Morton keys (3 x <=10 bits in 32, 3 x <=21 bits in 64): consecutive keys are spatially compact, so the 64
// rays of a wave stay close to each other along the whole march (coherent atomics, cheap de-duplication)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t part1by2_32(uint32_t x)
{
    x &= 0x3ffu;
    x = (x ^ (x << 16)) & 0xff0000ffu;
    x = (x ^ (x << 8)) & 0x0300f00fu;
    x = (x ^ (x << 4)) & 0x030c30c3u;
    x = (x ^ (x << 2)) & 0x09249249u;
    return x;
}
__device__ __forceinline__ uint64_t part1by2_64(uint64_t x)
{
    x &= 0x1fffffull;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
template <typename K> struct KeyOps;
template <> struct KeyOps<uint32_t> {
    static __device__ __forceinline__ uint32_t make(int x, int y, int z) { return (part1by2_32(x) << 2) | (part1by2_32(y) << 1) | part1by2_32(z); }
    static __device__ __host__ __forceinline__ uint32_t invalid(int bits) { return 1u << (3 * bits); }
};
template <> struct KeyOps<uint64_t> {
    static __device__ __forceinline__ uint64_t make(int x, int y, int z) { return (part1by2_64(x) << 2) | (part1by2_64(y) << 1) | part1by2_64(z); }
    static __device__ __host__ __forceinline__ uint64_t invalid(int bits) { return 1ull << (3 * bits); }
};

// ------------------------------------------------------------------------------------------------------
// K1: depth image -> sensor-centred voxel key + f16 payload       dense_tsdf.py:188-213, process_point :227-229
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int group_insert(const FrameDev& F, unsigned long long vkey, bool inside, uint32_t pid, int log2n, bool* opened, int* xslot);

template <typename K>
__global__ void __launch_bounds__(256) k_voxelize_depth(BatchDev B)
{
    const int fq = blockIdx.y;                      // frame of the batch
    if (fq >= B.n) return;
    const FrameDev& F = B.f[fq];
    const FrameParams& P = *B.p[fq];
    if (P.points) return;
    K* __restrict__ keys = reinterpret_cast<K*>(F.keys);
    const uint16_t* __restrict__ depth = static_cast<const uint16_t*>(P.input);
    // a workgroup visits a 16x16 tile of the sampled pixels: the voxels it opens are listed together (block_reserve), so
    // neighbouring rays -- which cross the same bricks -- are walked and binned by the same workgroup later on
    const int tiles_x = (P.ww + 15) >> 4;
    if ((int)blockIdx.x >= tiles_x * ((P.hh + 15) >> 4)) return;      // the grid covers the largest image of the batch
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int jj = ty * 16 + ((int)threadIdx.x >> 4), ii = tx * 16 + ((int)threadIdx.x & 15);
    const int p = jj * P.ww + ii;                                                    // pixel id = raster order
    bool gate = false, inside = false, opened = false;
    int slot = -1, xslot = -1;
    unsigned long long vkey = 0ull;
    if (jj < P.hh && ii < P.ww) {
        const int j = jj * P.step, i = ii * P.step;
        const uint16_t d = depth[(size_t)jj * P.rstride + (size_t)ii * P.cstride];
        K key = KeyOps<K>::invalid(P.pcl_bits);
        uint2 payload = make_uint2(0u, 0u);
        const float df = (float)d;
        if (d != 0 && !(df > P.thr_max) && !(df < P.thr_min)) {                     // :196-199
            gate = true;
            const float dep = df / 1000.0f;                                           // :201
            const float px = ((float)i - P.cx) * dep / P.fx;                          // mapping_common.py:37-40
            const float py = ((float)j - P.cy) * dep / P.fy;
            const float pz = dep;
            const float mx = (P.R[0] * px + P.R[1] * py) + P.R[2] * pz;               // :203 (rotation only)
            const float my = (P.R[3] * px + P.R[4] * py) + P.R[5] * pz;
            const float mz = (P.R[6] * px + P.R[7] * py) + P.R[8] * pz;
            const int cx = rnd_i(div_vs(mx, P.vs, P.rvs, P.fastdiv)) - P.pcl_lo, cy = rnd_i(div_vs(my, P.vs, P.rvs, P.fastdiv)) - P.pcl_lo, cz = rnd_i(div_vs(mz, P.vs, P.rvs, P.fastdiv)) - P.pcl_lo;   // :229
            if (cx >= 0 && cx < P.pcl_ext && cy >= 0 && cy < P.pcl_ext && cz >= 0 && cz < P.pcl_ext) {
                inside = true;
                key = KeyOps<K>::make(cx, cy, cz);
                payload.x = (uint32_t)f2h(mx) | ((uint32_t)f2h(my) << 16);
                payload.y = (uint32_t)f2h(mz) | ((uint32_t)f2h(dep) << 16);
            }
        }
        F.pix[p] = payload;
        if (!P.group) { keys[p] = key; F.vals[p] = (uint32_t)p; }
        vkey = (unsigned long long)key;
    }
    if (P.group) {
        slot = group_insert(F, vkey, inside, (uint32_t)p, P.hlog2, &opened, &xslot);      // wave-cooperative: reached by every lane
        const int q = block_reserve(&F.counters[6], opened); if (opened) F.act[q] = slot;          // position in the list = ray id
        const int x = wave_reserve(&F.counters[7], xslot >= 0); if (xslot >= 0) F.actx[x] = xslot;   // (rare) overflow slots: listed to be cleared
    }
    block_count_add(&F.stats->p_valid, inside);
    block_count_add(&F.stats->p_oob, gate && !inside);
}

// recast_pcl_to_map_kernel  dense_tsdf.py:167-186 (z := range, gate on the range)
template <typename K>
__global__ void __launch_bounds__(256) k_voxelize_points(BatchDev B)
{
    const int fq = blockIdx.y;                      // frame of the batch
    if (fq >= B.n) return;
    const FrameDev& F = B.f[fq];
    const FrameParams& P = *B.p[fq];
    if (!P.points || (int)blockIdx.x * 256 >= P.total) return;
    K* __restrict__ keys = reinterpret_cast<K*>(F.keys);
    const float* __restrict__ xyz = static_cast<const float*>(P.input);
    const int n = P.total;
    const int p = blockIdx.x * 256 + threadIdx.x;
    bool gate = false, inside = false, opened = false;
    int slot = -1, xslot = -1;
    unsigned long long vkey = 0ull;
    if (p < n) {
        const float px = xyz[(size_t)p * 3], py = xyz[(size_t)p * 3 + 1], pz = xyz[(size_t)p * 3 + 2];
        const float mx = (P.R[0] * px + P.R[1] * py) + P.R[2] * pz;                   // :175
        const float my = (P.R[3] * px + P.R[4] * py) + P.R[5] * pz;
        const float mz = (P.R[6] * px + P.R[7] * py) + P.R[8] * pz;
        const float len = sqrt_rn((mx * mx + my * my) + mz * mz);                  // :176
        K key = KeyOps<K>::invalid(P.pcl_bits);
        uint2 payload = make_uint2(0u, 0u);
        if (len < P.max_ray_f) {                                                      // :177
            gate = true;
            const int cx = rnd_i(div_vs(mx, P.vs, P.rvs, P.fastdiv)) - P.pcl_lo, cy = rnd_i(div_vs(my, P.vs, P.rvs, P.fastdiv)) - P.pcl_lo, cz = rnd_i(div_vs(mz, P.vs, P.rvs, P.fastdiv)) - P.pcl_lo;
            if (cx >= 0 && cx < P.pcl_ext && cy >= 0 && cy < P.pcl_ext && cz >= 0 && cz < P.pcl_ext) {
                inside = true;
                key = KeyOps<K>::make(cx, cy, cz);
                payload.x = (uint32_t)f2h(mx) | ((uint32_t)f2h(my) << 16);
                payload.y = (uint32_t)f2h(mz) | ((uint32_t)f2h(len) << 16);
            }
        }
        F.pix[p] = payload;
        if (!P.group) { keys[p] = key; F.vals[p] = (uint32_t)p; }
        vkey = (unsigned long long)key;
    }
    if (P.group) {
        slot = group_insert(F, vkey, inside, (uint32_t)p, P.hlog2, &opened, &xslot);      // wave-cooperative: reached by every lane
        const int q = block_reserve(&F.counters[6], opened); if (opened) F.act[q] = slot;          // position in the list = ray id
        const int x = wave_reserve(&F.counters[7], xslot >= 0); if (xslot >= 0) F.actx[x] = xslot;   // (rare) overflow slots: listed to be cleared
    }
    block_count_add(&F.stats->p_valid, inside);
    block_count_add(&F.stats->p_oob, gate && !inside);
}

// ------------------------------------------------------------------------------------------------------
// K3: one ray per sensor voxel.  The voxel's pixels are replayed in raster order with per-add f16 rounding
// (process_point :230-234) and turned into a ray record (process_new_pcl :242-249).
// ------------------------------------------------------------------------------------------------------
// (a) pixels grouped by a stable radix sort of the sensor-voxel keys: one thread per sorted entry, segment heads work
template <typename K>
__global__ void __launch_bounds__(256) k_build_rays(const FrameParams* __restrict__ Pp, FrameDev F, const K* __restrict__ keys_s)
{
    const FrameParams& P = *Pp;
    const int total = P.total;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const K bad = KeyOps<K>::invalid(P.pcl_bits);
    bool head = false, ok = false;
    uint4 rec = make_uint4(0, 0, 0, 0);
    int nsteps = 0;
    uint32_t first = 0;
    if (i < total) {
        const K k = keys_s[i];
        head = (k != bad) && (i == 0 || keys_s[i - 1] != k);
        if (head) {
            PixAcc A = {};
            first = F.vals_s[i];                                                     // stable sort: the run starts with its lowest pixel id
            for (int q = i; q < total && keys_s[q] == k; ++q) acc_pixel(P, F, F.vals_s[q], A);
            ok = finish_ray(P, F, A, first, &rec, &nsteps);
        }
    }
    const int r = block_reserve(F.nrays, ok);
    if (ok) { F.rayA[r] = rec; F.rayN[r] = nsteps; F.rayFirst[r] = first; }
    block_count_add(&F.stats->v_pcl, head);
    block_count_add(&F.stats->v_skipped, head && !ok);
}

// (b) pixels grouped through a hash table of sensor voxels (no sort, brick-binned path): k_voxelize_* insert every pixel into its
// voxel's slot and k_segments reads a voxel's pixels back with one 64-byte load and replays them in ASCENDING pixel id, i.e. raster
// order (HSlot, tsl_tsdf.hpp).  Neighbouring pixels fall into the same voxel (2-3 per voxel and more), and the L2 executes ~23 G
// scattered atomics per second chip-wide -- one find-or-insert CAS + one returning add PER PIXEL made this kernel the atomics' speed.
// So a wave first groups its lanes by key (pure lane arithmetic: one iteration per distinct key), the first lane of every group does the
// CAS and reserves the whole group's ranks with ONE add, and the slot / first rank travel back to the group's lanes.
// Must be reached by every lane of the wave.  Returns the slot of the voxel (-1 for lanes without a pixel);
// *opened = this lane opened the voxel in this frame; *xslot = overflow slot this lane opened (or -1).
__device__ __forceinline__ int group_insert(const FrameDev& F, unsigned long long vkey, bool inside, uint32_t pid, int log2n, bool* opened, int* xslot)
{
    HSlot* const tab = F.htab;
    const uint32_t mask = (1u << log2n) - 1u;
    const int lane = lane_id();
    const uint32_t klo = (uint32_t)vkey, khi = (uint32_t)(vkey >> 32);
    int leader = lane, grank = 0, gsize = 1;
    for (unsigned long long todo = __ballot(inside); todo; ) {
        const int l0 = (int)__builtin_ctzll(todo);                                   // (uniform)
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)klo, l0), hi = (uint32_t)__builtin_amdgcn_readlane((int)khi, l0);
        const bool mine = inside && klo == lo && khi == hi;
        const unsigned long long grp = __ballot(mine);
        if (mine) { leader = l0; grank = rank_below(grp); gsize = popc64(grp); }
        todo &= ~grp;
    }
    int h = -1, r0 = 0;
    if (inside && lane == leader) {
        const unsigned long long k0 = h_key(vkey, 0);
        uint32_t hh = h_hash64(k0, log2n);
        for (;;) {
            const unsigned long long cur = atomicCAS(&tab[hh].key, H_EMPTY, k0);
            if (cur == H_EMPTY) { *opened = true; break; }
            if (cur == k0) break;
            hh = (hh + 1u) & mask;
        }
        h = (int)hh;
        r0 = atomicAdd(&tab[hh].cnt, gsize);
    }
    h = __shfl(h, leader); r0 = __shfl(r0, leader);
    if (!inside) return -1;
    const int r = r0 + grank;
    if (r < H_INL) tab[h].pix[r] = pid;
    else {      // crowded voxel: pixels H_INL.. live in slots keyed (voxel, block)
        const unsigned long long k2 = h_key(vkey, r / H_INL);
        uint32_t h2 = h_hash64(k2, log2n);
        for (;;) {
            const unsigned long long cur = atomicCAS(&tab[h2].key, H_EMPTY, k2);
            if (cur == H_EMPTY) { *xslot = (int)h2; break; }
            if (cur == k2) break;
            h2 = (h2 + 1u) & mask;
        }
        tab[h2].pix[r % H_INL] = pid;
    }
    return h;
}

// ------------------------------------------------------------------------------------------------------
// sparse export / import / compaction kernels (struct-for over active cells in the reference)
// ------------------------------------------------------------------------------------------------------
// count_active  dense_tsdf.py:412-423
__global__ void __launch_bounds__(256) k_count_active(MapDev M, int s, long long* out)
{
    long long c = 0;
    for (int b = blockIdx.x; b < M.nb3; b += gridDim.x) {
        const int p = pool_lookup_ro(M, s, b);
        if (p < 0) continue;
        const int8_t* obs = M.obs + (size_t)p * TSL_BRK3;
        for (int l = threadIdx.x; l < TSL_BRK3; l += 256) c += obs[l] > 0;
    }
    c = wave_sum_ll(c);
    if (lane_id() == 0 && c) __hip_atomic_fetch_add(out, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void voxel_ijk(const MapDev& M, int b, int l, int* i, int* j, int* k)
{
    const int bk = b % M.nbz, bj = (b / M.nbz) % M.nbx, bi = b / (M.nbz * M.nbx);
    *i = bi * 16 + (l >> 8) - M.hN; *j = bj * 16 + ((l >> 4) & 15) - M.hN; *k = bk * 16 + (l & 15) - M.hNz;
}

// to_numpy  dense_tsdf.py:425-440 (mode 0: observed voxels; mode 1: occupy != 0)
__global__ void __launch_bounds__(256) k_export_sparse(MapDev M, int s, int mode, int16_t* idx, uint16_t* t, uint16_t* w, int8_t* occ, uint16_t* col,
                                                       long long cap, int* counter)
{
    for (int b = blockIdx.x; b < M.nb3; b += gridDim.x) {
        const int p = pool_lookup_ro(M, s, b);
        if (p < 0) continue;
        for (int l0 = 0; l0 < TSL_BRK3; l0 += 256) {
            const int l = l0 + threadIdx.x;
            const size_t v = (size_t)p * TSL_BRK3 + l;
            const bool pred = mode == 0 ? (M.obs[v] > 0) : (M.occ[v] != 0);
            const int o = wave_reserve(counter, pred);
            if (pred && o < cap) {
                int i, j, k; voxel_ijk(M, b, l, &i, &j, &k);
                idx[(size_t)o * 3] = (int16_t)i; idx[(size_t)o * 3 + 1] = (int16_t)j; idx[(size_t)o * 3 + 2] = (int16_t)k;
                const uint32_t tv = M.tw[v];
                if (t) t[o] = (uint16_t)(tv & 0xffffu);
                if (w) w[o] = (uint16_t)(tv >> 16);
                if (occ) occ[o] = M.occ[v];
                if (col && M.col) for (int a = 0; a < 3; ++a) col[(size_t)o * 3 + a] = M.col[v * 4 + a];
            }
        }
    }
}

// load_numpy  dense_tsdf.py:442-454
__global__ void __launch_bounds__(256) k_import_sparse(MapDev M, int s, const int16_t* idx, const uint16_t* t, const uint16_t* w, const int8_t* occ,
                                                       const uint16_t* col, long long n)
{
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const int i = idx[q * 3], j = idx[q * 3 + 1], k = idx[q * 3 + 2];
    if (!in_volume(M, i, j, k)) return;
    int l; const int b = brick_of(M, i, j, k, &l);
    const int p = pool_claim(M, s, b);
    if (p < 0) return;
    const size_t v = (size_t)p * TSL_BRK3 + l;
    M.tw[v] = (uint32_t)t[q] | ((uint32_t)w[q] << 16);
    M.occ[v] = occ ? occ[q] : (int8_t)0;
    if (col && M.col) for (int a = 0; a < 3; ++a) M.col[v * 4 + a] = col[q * 3 + a];
    M.obs[v] = 1;
}

struct PoseF { float R[9], T[3]; };
// color_from_colomap  mapping_common.py:216-219
__device__ __forceinline__ int colormap_index(float z, float lo, float hi)
{
    float t = ((z - lo) / (hi - lo)) * 1023.0f;
    if (t > 1023.0f) t = 1023.0f;
    if (!(t > 0.0f)) t = 0.0f;
    return (int)t;
}
__device__ __forceinline__ void voxel_xyz(const PoseF& B, int is_global, float vs, int i, int j, int k, float* o)
{
    const float p0 = (float)i * vs, p1 = (float)j * vs, p2 = (float)k * vs;                    // mapping_common.py:221-227
    if (is_global) { o[0] = p0; o[1] = p1; o[2] = p2; return; }
    for (int a = 0; a < 3; ++a) o[a] = ((B.R[a * 3] * p0 + B.R[a * 3 + 1] * p1) + B.R[a * 3 + 2] * p2) + B.T[a];   // :229-232
}

// cvt_TSDF_surface_to_voxels_kernel :339-365 (mode 0) and cvt_TSDF_to_voxels_slice_kernel :367-385 (mode 1)
__global__ void __launch_bounds__(256) k_export_particles(MapDev M, int s, int mode, PoseF B, int is_global, float vs, float thres, float zfloor, float zceil,
                                                          int slice_index, float slice_dz, const float* __restrict__ cmap,
                                                          float* xyz, float* rgb, float* val, long long cap, int* counter)
{
    for (int b = blockIdx.x; b < M.nb3; b += gridDim.x) {
        const int p = pool_lookup_ro(M, s, b);
        if (p < 0) continue;
        for (int l0 = 0; l0 < TSL_BRK3; l0 += 256) {
            const int l = l0 + threadIdx.x;
            const size_t v = (size_t)p * TSL_BRK3 + l;
            const int8_t ob = M.obs[v];
            const float tv = h2f((h16)(M.tw[v] & 0xffffu));
            int i, j, k; voxel_ijk(M, b, l, &i, &j, &k);
            float o[3] = {0, 0, 0};
            bool pred;
            if (mode == 0) {
                pred = (ob == 1) && (fabsf(tv) < thres);                                      // :349-350
                if (pred) { voxel_xyz(B, is_global, vs, i, j, k, o); pred = !(o[2] > zceil || o[2] < zfloor); }   // :356
            } else {
                pred = (ob > 0) && ((float)slice_index - slice_dz < (float)k) && ((float)k < (float)slice_index + slice_dz);   // :376-377
                if (pred) voxel_xyz(B, is_global, vs, i, j, k, o);
            }
            const int idx = wave_reserve(counter, pred);                                      // :358 (Q10: clamp by the returned index)
            if (pred && idx < cap) {
                for (int a = 0; a < 3; ++a) xyz[(size_t)idx * 3 + a] = o[a];
                if (mode == 1 && val) val[idx] = tv;                                          // :380
                if (rgb) {
                    if (mode == 0 && M.col) { for (int a = 0; a < 3; ++a) rgb[(size_t)idx * 3 + a] = h2f(M.col[v * 4 + a]); }    // :361
                    else {
                        const int ci = mode == 0 ? colormap_index(o[2], zfloor, zceil) : colormap_index(tv, -0.5f, 0.5f);      // :364,:385
                        for (int a = 0; a < 3; ++a) rgb[(size_t)idx * 3 + a] = cmap[ci * 3 + a];
                    }
                }
            }
        }
    }
}

// PointCloud2 data block: interleaved float32 rows [x y z] or [x y z r g b]  (taichi_slam/utils/ros_pcl_transfer.py:96-136, scripts/taichislam_node.py:420-425)
__global__ void __launch_bounds__(256) k_pack_pointcloud2(const float* __restrict__ xyz, const float* __restrict__ rgb, float* __restrict__ out, long long n, int stride)
{
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= n * stride) return;
    const long long row = q / stride; const int c = (int)(q - row * stride);
    out[q] = c < 3 ? xyz[row * 3 + c] : rgb[row * 3 + c - 3];
}

// reset(): hand every brick back (dense_tsdf.py:309-310 deactivates the whole tree)
__global__ void __launch_bounds__(256) k_reset_bricks(MapDev M, int nused)
{
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        uint32_t* tw = M.tw + (size_t)p * TSL_BRK3;
        uint32_t* ob = reinterpret_cast<uint32_t*>(M.obs + (size_t)p * TSL_BRK3);
        uint32_t* oc = reinterpret_cast<uint32_t*>(M.occ + (size_t)p * TSL_BRK3);
        for (int l = threadIdx.x; l < TSL_BRK3; l += 256) tw[l] = 0u;
        for (int l = threadIdx.x; l < TSL_BRK3 / 4; l += 256) { ob[l] = 0u; oc[l] = 0u; }
        if (M.col) { uint2* c = reinterpret_cast<uint2*>(M.col + (size_t)p * TSL_BRK3 * 4); for (int l = threadIdx.x; l < TSL_BRK3; l += 256) c[l] = make_uint2(0u, 0u); }
        if (threadIdx.x == 0) M.table[M.owner[p]] = TSL_EMPTY;
    }
}

// Verify on the device that the fma-refined reciprocal division equals IEEE division bit for bit for every float x whose
// quotient can reach an integer conversion (2^-100 <= |x|, |x/vs| < 2^31) -- 2^32 candidates, ~1 ms.  *bad counts mismatches.
__global__ void __launch_bounds__(256) k_verify_div(float vs, float rvs, unsigned long long* bad)
{
    unsigned long long nbad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * 256) {
        const float x = __uint_as_float((uint32_t)i);
        const float ax = fabsf(x);
        if (!(ax >= 7.8886090522101181e-31f) || !(ax < 2147483648.0f * vs)) continue;
        const float q = x / vs;
        const float f = div_vs(x, vs, rvs, 1);
        if (__float_as_uint(q) != __float_as_uint(f)) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

// Exhaustive checks of the two arithmetic shortcuts that do not depend on any parameter (tsl_selftest):
// which 0: rnd_i == round-half-away for every float; which 1: sqrt_rn_norm == sqrtf for every float in [2^-96, inf).
__global__ void __launch_bounds__(256) k_selftest(int which, unsigned long long* bad)
{
    unsigned long long nbad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * 256) {
        const float x = __uint_as_float((uint32_t)i);
        if (which == 0) {
            if (!(fabsf(x) < 2147483648.0f)) continue;
            if (rnd_i(x) != rnd_i_ref(x)) ++nbad;
        } else {
            if (!(x >= 1.2621774483536189e-29f) || !(x < INFINITY)) continue;
            if (__float_as_uint(sqrt_rn_norm(x)) != __float_as_uint(sqrt_rn(x))) ++nbad;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
int grow(void** p, size_t* have, size_t need)
{
    if (*have >= need) return TSL_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *have = 0;
    size_t want = need + need / 4 + 4096;
    TSL_HIP(hipMalloc(p, want));
    *have = want;
    return TSL_OK;
}

int dev_alloc(tsl_tsdf* m, void** p, size_t bytes, int fill)
{
    TSL_HIP(hipMalloc(p, bytes));
    TSL_HIP(hipMemsetAsync(*p, fill, bytes, m->stream_));
    m->bytes += (int64_t)bytes;
    return TSL_OK;
}

void prof_begin(tsl_tsdf* m, int kid, hipStream_t st, int count)
{
    if (m->prof_group) return;                     // inside a bracket that covers several launches
    m->prof_open = false;
    if (!m->prof_on || !((m->prof_mask >> kid) & 1)) return;
    ProfSlot s; s.kid = kid; s.count = count;
    if (m->prof_free.size() >= 2) { s.a = m->prof_free.back(); m->prof_free.pop_back(); s.b = m->prof_free.back(); m->prof_free.pop_back(); }
    else if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return;
    (void)hipEventRecord(s.a, st ? st : m->stream_);
    m->prof.push_back(s);
    m->prof_open = true;
}
// a timing slot whose events are attached to ONE kernel dispatch (hipExtLaunchKernelGGL records start / stop in the dispatch itself: no
// marker packets on the stream); false when this kernel is not being profiled
bool prof_slot(tsl_tsdf* m, int kid, int count, hipEvent_t* a, hipEvent_t* b)
{
    if (m->prof_group || !m->prof_on || !((m->prof_mask >> kid) & 1)) return false;
    ProfSlot s; s.kid = kid; s.count = count;
    if (m->prof_free.size() >= 2) { s.a = m->prof_free.back(); m->prof_free.pop_back(); s.b = m->prof_free.back(); m->prof_free.pop_back(); }
    else if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return false;
    m->prof.push_back(s);
    *a = s.a; *b = s.b;
    return true;
}
void prof_end(tsl_tsdf* m, hipStream_t st)
{
    if (m->prof_group || !m->prof_open) return;
    (void)hipEventRecord(m->prof.back().b, st ? st : m->stream_);
    m->prof_open = false;
}

// set_pose + convert_by_base  mapping_common.py:91-100,149-156 (float64, fixed summation order, then f32)
void convert_pose(const double* Rb, const double* Tb, const double* R, const double* T, float* outR, float* outT)
{
    const double d[3] = { T[0] - Tb[0], T[1] - Tb[1], T[2] - Tb[2] };
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += Rb[k * 3 + i] * R[k * 3 + j];
            outR[i * 3 + j] = (float)acc;
        }
        double acc = 0.0;
        for (int k = 0; k < 3; ++k) acc += Rb[k * 3 + i] * d[k];
        outT[i] = (float)acc;
    }
}

static int map_slot(const tsl_tsdf* m, int s) { return m->cfg.is_global_map ? 0 : s; }

static PoseF pose_of(const tsl_tsdf* m, int s)
{
    PoseF B;
    for (int a = 0; a < 9; ++a) B.R[a] = m->baseRf[(size_t)s * 9 + a];
    for (int a = 0; a < 3; ++a) B.T[a] = m->baseTf[(size_t)s * 3 + a];
    return B;
}

// Host inputs of a batch: pinned (device-mapped) buffer -> device staging buffer, 16 bytes per lane, at the head of the batch's phase A.  Round 4 let
// k_voxelize_depth read the pinned buffer in place: 2 bytes per lane in 16 x 16 pixel tiles, i.e. 32-byte reads across the host link -- 117 us per
// batch against 31 us from device memory.  One launch per batch, no copy call, no event on the way in.
__global__ void __launch_bounds__(256) k_stage_host(StageCopy C)
{
    const int k = blockIdx.y;
    const int n = C.n16[k];
    const uint4* __restrict__ src = C.src[k];
    uint4* __restrict__ dst = C.dst[k];
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < n; i += gridDim.x * 1024) {      // four independent 16-byte loads per lane in flight
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int j = i + u * 256; v[u] = j < n ? src[j] : make_uint4(0u, 0u, 0u, 0u); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int j = i + u * 256; if (j < n) dst[j] = v[u]; }
    }
}

// per-frame prologue of a working set: publish the frame's parameters, clear stats | nrays | counters (256 bytes)
__global__ void k_set_params(ParamPack PP, SetPtrs S)
{
    if (threadIdx.x == 0) *S.p[blockIdx.x] = PP.p[blockIdx.x];
    S.header[blockIdx.x][threadIdx.x] = 0;                                    // stats | nrays | counters: the set's 256-byte header
}

// phase A of a batch of n frames on stream `sa`: depth -> rays -> brick-sorted segments, every kernel once with grid.y = frame.
// The radix-sort grouping and the global-atomics variants only exist per frame (batches of one).
template <typename K>
static int enqueue_phase_a(tsl_tsdf* m, const BatchDev& B, const FrameParams* hp, hipStream_t sa)
{
    const int n = B.n;
    int total = 0, tiles = 0, any_points = 0, any_depth = 0;
    for (int q = 0; q < n; ++q) {
        total = hp[q].total > total ? hp[q].total : total;
        const int t = ((hp[q].ww + 15) / 16) * ((hp[q].hh + 15) / 16);
        if (hp[q].points) any_points = 1; else { any_depth = 1; tiles = t > tiles ? t : tiles; }
    }
    if (total <= 0) return TSL_OK;
    const FrameParams& P0 = hp[0];
    const int blocks = (total + 255) / 256;
    prof_begin(m, TSL_K_VOXELIZE, sa);
    if (any_points) hipLaunchKernelGGL(k_voxelize_points<K>, dim3(blocks, n), dim3(256), 0, sa, B);
    if (any_depth) hipLaunchKernelGGL(k_voxelize_depth<K>, dim3(tiles, n), dim3(256), 0, sa, B);
    prof_end(m, sa);
    if (!P0.group) {          // stable radix sort of the sensor-voxel keys (per frame: batches of one)
        const FrameDev& F = B.f[0];
        prof_begin(m, TSL_K_SORT, sa);
        size_t tb = m->sort_temp_bytes;
        TSL_HIP(rocprim::radix_sort_pairs(m->fset[m->cur * TSL_NB].sort_temp, tb, reinterpret_cast<K*>(F.keys), reinterpret_cast<K*>(F.keys_s), F.vals, F.vals_s, (size_t)total, 0u, (unsigned)(3 * P0.pcl_bits + 1), sa));
        prof_end(m, sa);
        prof_begin(m, TSL_K_RAYS, sa);
        hipLaunchKernelGGL(k_build_rays<K>, dim3(blocks), dim3(256), 0, sa, B.p[0], F, (const K*)reinterpret_cast<K*>(F.keys_s));
        prof_end(m, sa);
    }
    return launch_segments(m, B, hp, total, sa);
}

// Issue the queued frames: phase A of the whole batch on the batch's stream (it depends on the images, the poses and the map
// GEOMETRY only -- its one access to the map is first-touch brick allocation and the occupancy byte -- so it runs beside phase
// B of the previous batch), then phase B frame by frame on the main stream.
template <typename K>
static int launch_batch_t(tsl_tsdf* m)
{
    const int n = m->npend;
    if (n == 0) return TSL_OK;
    m->npend = 0;                                   // nothing below re-enters through ms()
    // rocPRIM (and the checks below) read the THREAD's last HIP error, which any earlier call of the process may have left behind -- the caller's own
    // HIP / torch calls included.  A stale error must not drop this batch: it is discarded here.
    (void)hipGetLastError();
    m->last_batch_n = n;
    const int bi = m->cur;
    BatchHost& H = m->batch[bi];
    const bool serial = m->overlap == 0;
    {   // has the pipeline run dry?  (phase B of the batch issued last has completed)
        const BatchHost& L = m->batch[(bi + TSL_NBATCH - 1) % TSL_NBATCH];
        if (!L.b_pending || hipEventQuery(L.b_done) == hipSuccess) { m->ramp = 0; ++m->dry_launches; } else if (m->ramp < 1000) ++m->ramp;
        m->shape_hash = (m->shape_hash ^ (uint32_t)n) * 16777619u;
    }
    hipStream_t sa = serial ? m->stream_ : H.st;
    // back-pressure: the host never runs more than TSL_INFLIGHT batches ahead of the device (bounded queues, bounded lifetime of the
    // callers' input buffers); waiting for the batch issued TSL_INFLIGHT batches ago also tells which frames have been consumed
    const int ring = (int)(m->batch_seq % TSL_INFLIGHT);
    if (m->ring_upto[ring] > 0) {
        TSL_HIP(hipEventSynchronize(m->ring_ev[ring]));
        if (m->ring_upto[ring] > m->frames_consumed) m->frames_consumed = m->ring_upto[ring];
    }
    // (TSL_FAULT_NO_BDONE_WAIT: fault injection for tests/test_pipeline_overlap_gpu.py -- without this wait phase A of a batch overwrites working sets the
    //  replay three batches back still reads; the back-to-back parity tests must notice)
#ifdef TSL_TEST_HOOKS      // compiled into lib/libtaichislam_hip_testhooks.so only: the product library has no switch that makes the map wrong
    static const bool fault_no_wait = std::getenv("TSL_FAULT_NO_BDONE_WAIT") != nullptr;
#else
    constexpr bool fault_no_wait = false;
#endif
    if (!serial && H.b_pending && !fault_no_wait) TSL_HIP(hipStreamWaitEvent(sa, H.b_done, 0));      // phase B of this batch's previous frames still reads the sets
    if (!serial && m->esdf_gate_set) {
        // an ESDF update in flight has taken its brick snapshot (tsl_esdf.hip).  The gate stays armed until every phase-A stream has waited for
        // it: the batch after next uses a third stream, which is ordered behind neither the update's stream nor the first waiter
        TSL_HIP(hipStreamWaitEvent(sa, m->esdf_gate_ev, 0));
        m->esdf_gate_mask |= 1u << (bi % TSL_NSTREAMS);
        if (m->esdf_gate_mask == (1u << TSL_NSTREAMS) - 1u) m->esdf_gate_set = false;
    }
    for (int k = 0; k < m->nproducers; ++k) {       // device inputs: phase A waits for what their producers had queued (tsl_tsdf_input_stream)
        if (m->producers[k] == sa) continue;
        // nothing pending on the producer: no wait.  (Not only a saving: an event recorded on an idle stream still lands in the hardware
        // queue the runtime maps that stream to -- four queues for all streams -- and completes behind whatever another stream has in
        // that queue: phase A waited for the ESDF rounds of the previous frame that way, every third frame.)
        if (hipStreamQuery(m->producers[k]) == hipSuccess) continue;
        if (!m->in_ev[0]) for (auto& e : m->in_ev) TSL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        hipEvent_t e = m->in_ev[m->in_ev_next]; m->in_ev_next = (m->in_ev_next + 1) % 8;
        TSL_HIP(hipEventRecord(e, m->producers[k]));
        TSL_HIP(hipStreamWaitEvent(sa, e, 0));
    }
    m->nproducers = 0;
    BatchDev B; ParamPack PP;
    for (int q = 0; q < n; ++q) { FSet& S = m->fset[bi * TSL_NB + q]; B.f[q] = S.F; B.p[q] = S.Pd; PP.p[q] = m->pend[q]; }
    for (int q = n; q < TSL_NB; ++q) { B.f[q] = B.f[0]; B.p[q] = B.p[0]; PP.p[q] = PP.p[0]; }
    B.n = n;
    SetPtrs SP;                                     // (the parameter blocks and the working sets together exceed the 4 KiB of kernel arguments)
    for (int q = 0; q < TSL_NB; ++q) { SP.p[q] = const_cast<FrameParams*>(B.p[q]); SP.header[q] = reinterpret_cast<int*>(B.f[q].stats); }
    hipLaunchKernelGGL(k_set_params, dim3(n), dim3(64), 0, sa, PP, SP);
    {   // host inputs: out of the pinned buffers, into the sets' device staging buffers (the frames' parameters already point there)
        StageCopy C; int most = 0; bool any_host = false;
        for (int q = 0; q < TSL_NB; ++q) {
            FSet& S = m->fset[bi * TSL_NB + q];
            const bool on = q < n;
            C.src[2 * q] = static_cast<const uint4*>(S.pin_dev); C.dst[2 * q] = static_cast<uint4*>(S.stage_in); C.n16[2 * q] = on ? (int)((S.copy_in + 15) / 16) : 0;
            C.src[2 * q + 1] = reinterpret_cast<const uint4*>(static_cast<const char*>(S.pin_dev) + S.copy_tex_off); C.dst[2 * q + 1] = static_cast<uint4*>(S.stage_tex); C.n16[2 * q + 1] = on ? (int)((S.copy_tex + 15) / 16) : 0;
            most = std::max(most, std::max(C.n16[2 * q], C.n16[2 * q + 1]));
            any_host = any_host || C.n16[2 * q] || C.n16[2 * q + 1];
            if (on) { S.copy_in = 0; S.copy_tex = 0; }
        }
        if (any_host) {
            prof_begin(m, TSL_K_VOXELIZE, sa);
            hipLaunchKernelGGL(k_stage_host, dim3(std::min(64, (most + 1023) / 1024), 2 * n), dim3(256), 0, sa, C);
            prof_end(m, sa);
            if (!serial) { TSL_HIP(hipEventRecord(H.c_done, sa)); H.c_recorded = true; }
        }
    }
    if (m->phases & 1) { int rc = enqueue_phase_a<K>(m, B, m->pend, sa); if (rc) return rc; }
    const bool seq_bricks = m->pend[0].seq && m->seq_impl && m->pend[0].variant == 2;
    if (seq_bricks && (m->phases & 1)) { int rc = launch_seq_group(m, B, m->pend, bi, sa); if (rc) return rc; }      // sequential semantics: replay runs per (frame, brick), still map-independent
    m->frames_issued += n;
    if (!serial) {
        TSL_HIP(hipEventRecord(H.a_done, sa)); H.a_recorded = true;
        TSL_HIP(hipStreamWaitEvent(m->stream_, H.a_done, 0));
    }
    // Split launches (option "split_launch", off by default; full batches of the overlapped pipeline): the brick kernel runs twice.  The PARTS
    // of the heavy bricks depend on the batch's phase A only -- a part reads its frame's rays and writes a slab slot of its own -- so they
    // are launched here, on the phase-A stream, and run beside the previous batch's phase B; the UNITS (they read and write the map) and
    // k_apply_slab follow on the main stream, whose chain is shorter by the parts.  Bit-exact, and measured SLOWER (26.9 k against 30.3 k
    // frames/s, the two launches take 377 us together against 214 us for one): the brick kernel is bound by the SIMDs' issue rate, not by
    // the length of the chain, and two launches side by side slow each other by more than the overlap saves.
    bool any = false; for (int q = 0; q < n; ++q) any = any || m->pend[q].total > 0;
    const bool split = !serial && m->split_launch && (m->phases & 2) && m->pend[0].variant == 2 && !m->pend[0].seq && n > 2 && any;
    if (split) {
        hipEvent_t pa = nullptr, pb = nullptr;
        const bool timed = prof_slot(m, TSL_K_INTEGRATE, 0, &pa, &pb);       // counted with the units launch: one "launch" of the batch in the statistics
        int rc = launch_brick(m, B, m->pend[0], 2, sa, timed ? pa : nullptr, timed ? pb : nullptr);
        if (rc) return rc;
        TSL_HIP(hipEventRecord(H.p_done, sa));
    }
    // ---- phase B: apply to the map on the main stream: the brick kernel takes the whole batch in one launch (frame order is kept per
    //      brick inside it); the global-atomics variants run frame by frame ----
    if (m->phases & 2) {
        int rc = TSL_OK;
        if (m->pend[0].variant == 2) {
            if (any) {
                hipEvent_t ea = nullptr, eb = nullptr;
                // (a timing slot is only taken where its events are attached to a dispatch: round 4 took one for the sequential branches as well and never
                //  recorded it -- hipEventElapsedTime on it left "invalid resource handle" as the thread's last error, which the NEXT rocPRIM call,
                //  of whatever handle, returned as its own: bench.py's first reference-source vector lost its frames that way)
                const bool timed = !m->pend[0].seq && prof_slot(m, TSL_K_INTEGRATE, 1, &ea, &eb);
                if (seq_bricks) { prof_begin(m, TSL_K_INTEGRATE); rc = launch_seq_apply(m, B, m->pend[0], bi); prof_end(m); }
                else if (m->pend[0].seq) { prof_begin(m, TSL_K_INTEGRATE); rc = launch_apply_sequential(m, B, m->pend[0]); prof_end(m); }
                else if (split) {
                    rc = launch_brick(m, B, m->pend[0], 1, m->stream_, timed ? ea : nullptr, timed ? eb : nullptr);
                    TSL_HIP(hipStreamWaitEvent(m->stream_, H.p_done, 0));
                    if (!rc) rc = launch_slab_apply(m, B, m->pend[0]);
                }
                else rc = launch_apply_batch(m, B, m->pend[0], timed ? ea : nullptr, timed ? eb : nullptr);
            }
        } else {
            for (int q = 0; q < n && !rc; ++q) {
                FSet& S = m->fset[bi * TSL_NB + q];
                m->P = m->pend[q];
                if (m->P.total > 0) rc = launch_apply(m, S, m->P.total);
            }
        }
        if (rc) return rc;
    }
    // one event marks the end of the batch's phase B: the back-pressure ring's, which the batch slot borrows as its "sets are free again"
    // event (a ring entry is recorded again TSL_INFLIGHT batches later, a slot is reused after TSL_NBATCH)
    TSL_HIP(hipEventRecord(m->ring_ev[ring], m->stream_)); m->ring_upto[ring] = m->frames_issued; m->batch_seq++;
    if (!serial) { H.b_done = m->ring_ev[ring]; H.b_pending = true; }
    m->cur = (bi + 1) % TSL_NBATCH;
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}
int flush_pending(tsl_tsdf* m)
{
    if (m->npend == 0) return TSL_OK;
    const FrameParams keep = m->P;                  // the launch helpers read the frame being issued from m->P
    const int rc = m->pcl_bits <= 10 ? launch_batch_t<uint32_t>(m) : launch_batch_t<uint64_t>(m);
    m->P = keep;
    if (rc && !m->deferred_rc) m->deferred_rc = rc;
    return rc;
}
// (the first call behind a synchronisation also discards a stale "last error" of the thread -- the caller's own HIP / torch calls leave theirs there, and
//  the launch checks below and rocPRIM would report it as this library's)
hipStream_t ms(tsl_tsdf* m) { if (m->clean) (void)hipGetLastError(); m->clean = false; (void)flush_pending(m); return m->stream_; }

static int batch_cap(const tsl_tsdf* m)
{ return (m->overlap > 0 && m->variant == 2 && m->P.group && (!m->semantics || m->seq_impl)) ? (m->overlap < TSL_NB ? m->overlap : TSL_NB) : 1; }
// working set the next queued frame will use; issues the queued frames first when the new one cannot join them
static int reserve_slot(tsl_tsdf* m, int points, int* set_index)
{
    const int slot = map_slot(m, m->active);
    if (m->npend && (m->npend >= batch_cap(m) || m->pend_points != points || m->pend[0].slot != slot)) { int rc = flush_pending(m); if (rc) return rc; }
    *set_index = m->cur * TSL_NB + m->npend;
    return TSL_OK;
}
// Host buffers: the visited rows are copied BY THE HOST into a pinned, device-mapped buffer of the frame's working set (the caller may reuse
// its buffers after return) and phase A reads them from there, across the host link: 307 kB per 640 x 480 frame at recast_step 2, each pixel
// once.  No copy call, no event, no stream synchronisation (round 3 staged through hipMemcpy2DAsync + a stream sync per call: 12.7 k frames/s
// from pageable images; an asynchronous copy from the pinned buffer still cost 50-75 us of HOST time per call inside the runtime).  The
// buffer is written again when the set comes round, three batches later: the host first makes sure phase A of the set's previous batch is done.
// `in`: `rows` rows of `row_bytes` bytes, `src_pitch` bytes apart (rows == 1: one contiguous block); they are stored back to back.
// pick > 1 (depth images): a row is uint16 pixels and only every pick-th of them is visited (dense_tsdf.py:194-195): the row is stored as
// row_bytes / 2 / pick pixels -- a quarter of a 640 x 480 image at recast_step 2, 154 kB, is all that crosses the link.
template <int PICK>
static void pick_pixels(uint16_t* __restrict__ dst, const uint16_t* __restrict__ src, int n, int pick)
{
    if (PICK) { for (int i = 0; i < n; ++i) dst[i] = src[(size_t)i * PICK]; }
    else for (int i = 0; i < n; ++i) dst[i] = src[(size_t)i * pick];
}
static int stage_host(tsl_tsdf* m, int si, const void* in, size_t row_bytes, int rows, size_t src_pitch, const void* tex, size_t tex_bytes, void** in_dev, void** tex_dev,
                      int pick = 1)
{
    const int npick = pick > 1 ? (int)(row_bytes / sizeof(uint16_t)) / pick : 0;
    const size_t out_row = pick > 1 ? (size_t)npick * sizeof(uint16_t) : row_bytes;
    const size_t in_bytes = out_row * (size_t)rows;
    FSet& S = m->fset[si];
    BatchHost& H = m->batch[si / TSL_NB];
    // the pinned buffer is free again once the copy kernel of the slot's previous batch has run (round 4 read it in place during all of phase A)
    if (m->overlap == 0) TSL_HIP(hipStreamSynchronize(m->stream_));          // one frame at a time on the main stream: nothing else orders the set's previous reader
    else if (H.c_recorded) TSL_HIP(hipEventSynchronize(H.c_done));
    const size_t tex_off = (in_bytes + 255) & ~(size_t)255;
    const size_t need = tex_off + ((tex && tex_bytes) ? tex_bytes : 0) + 256;
    if (S.pin_bytes < need) {
        if (S.pin) (void)hipHostFree(S.pin);
        S.pin = nullptr; S.pin_bytes = 0; S.pin_dev = nullptr;
        // COHERENT (fine-grained) host memory: the device reads it uncached, so what the host wrote before the batch was issued is what phase A
        // sees.  With hipHostMallocMapped alone the kind of memory is the runtime's choice; TSL_PIN_LEGACY=1 is the developer A/B.
#ifdef TSL_TEST_HOOKS
        static const bool legacy = std::getenv("TSL_PIN_LEGACY") != nullptr;
#else
        constexpr bool legacy = false;
#endif
        TSL_HIP(hipHostMalloc(&S.pin, need + need / 4, legacy ? hipHostMallocMapped : (hipHostMallocMapped | hipHostMallocCoherent)));
        S.pin_bytes = need + need / 4;
        TSL_HIP(hipHostGetDevicePointer(&S.pin_dev, S.pin, 0));
    }
    char* pin = static_cast<char*>(S.pin);
    if (in_bytes && pick > 1) {
        for (int r = 0; r < rows; ++r) {
            uint16_t* d = reinterpret_cast<uint16_t*>(pin + (size_t)r * out_row);
            const uint16_t* sp = reinterpret_cast<const uint16_t*>(static_cast<const char*>(in) + (size_t)r * src_pitch);
            if (pick == 2) pick_pixels<2>(d, sp, npick, 2); else if (pick == 4) pick_pixels<4>(d, sp, npick, 4); else pick_pixels<0>(d, sp, npick, pick);
        }
    } else if (in_bytes) {
        if (rows > 1 && src_pitch != row_bytes) { for (int r = 0; r < rows; ++r) std::memcpy(pin + (size_t)r * row_bytes, static_cast<const char*>(in) + (size_t)r * src_pitch, row_bytes); }
        else std::memcpy(pin, in, in_bytes);
    }
    // device staging buffers of the set (phase A of the slot's previous batch reads them until a_done: they are only re-allocated behind it)
    const size_t in16 = (in_bytes + 15) & ~(size_t)15, tex16 = ((tex && tex_bytes) ? tex_bytes + 15 : 0) & ~(size_t)15;
    if (S.stage_in_bytes < in16 || S.stage_tex_bytes < tex16) {
        if (m->overlap != 0 && H.a_recorded) TSL_HIP(hipEventSynchronize(H.a_done));
        int rc = grow(&S.stage_in, &S.stage_in_bytes, in16); if (rc) return rc;
        if (tex16) { rc = grow(&S.stage_tex, &S.stage_tex_bytes, tex16); if (rc) return rc; }
    }
    *in_dev = S.stage_in; *tex_dev = nullptr;
    S.copy_in = in_bytes; S.copy_tex = 0; S.copy_tex_off = tex_off;
    if (tex && tex_bytes) { std::memcpy(pin + tex_off, tex, tex_bytes); *tex_dev = S.stage_tex; S.copy_tex = tex_bytes; }
    return TSL_OK;
}

// frame scratch: shared part + TSL_NSETS per-frame working sets, allocated on the first integrate call
static int ensure_frame_scratch(tsl_tsdf* m)
{
    if (m->scratch_ready) return TSL_OK;
    const tsl_tsdf_cfg* cfg = &m->cfg;
    FrameDev& F = m->F;
    int rc;
    const size_t np = (size_t)F.max_points;
    if ((rc = dev_alloc(m, (void**)&F.slot_tab, sizeof(int) * (size_t)m->nb3, 0xff))) return rc;
    if ((rc = dev_alloc(m, (void**)&F.touched, sizeof(int) * (size_t)F.max_frame_bricks, 0))) return rc;
    if ((rc = dev_alloc(m, (void**)&F.touched_b, sizeof(int) * (size_t)F.max_frame_bricks, 0))) return rc;
    if ((rc = dev_alloc(m, (void**)&F.acc, 16 * (size_t)F.max_frame_bricks * TSL_BRK3, 0))) return rc;
    if ((rc = dev_alloc(m, (void**)&F.dbg, sizeof(long long) * 16384 * 16, 0))) return rc;
    if (cfg->texture_enabled) { if ((rc = dev_alloc(m, (void**)&F.accw, 4 * (size_t)F.max_frame_bricks * TSL_BRK3, 0))) return rc; }
    for (int si = 0; si < TSL_NSETS; ++si) {
        FSet& S = m->fset[si];
        S.F = F;
        FrameDev& G = S.F;
        auto own = [&](void** p, size_t bytes) -> int { int r = dev_alloc(m, p, bytes, 0); if (!r) S.owned.push_back(*p); return r; };
        if (si >= TSL_NB) {         // every further batch slot merges its split bricks through its own slab (owned by the slot's first set)
            FrameDev& G0 = m->fset[(si / TSL_NB) * TSL_NB].F;
            if (si % TSL_NB == 0) {
                if ((rc = own((void**)&G.acc, 16 * (size_t)F.max_frame_bricks * TSL_BRK3))) return rc;
                if (cfg->texture_enabled) { if ((rc = own((void**)&G.accw, 4 * (size_t)F.max_frame_bricks * TSL_BRK3))) return rc; }
            } else { G.acc = G0.acc; G.accw = G0.accw; }
        }
        if ((rc = own((void**)&G.keys, 8 * np))) return rc;
        if ((rc = own((void**)&G.keys_s, 8 * np))) return rc;
        if ((rc = own((void**)&G.vals, 4 * np))) return rc;
        if ((rc = own((void**)&G.vals_s, 4 * np))) return rc;
        if ((rc = own((void**)&G.pix, 8 * np))) return rc;
        if ((rc = own((void**)&G.rayA, 16 * np))) return rc;
        if ((rc = own((void**)&G.rayN, 4 * np))) return rc;
        if ((rc = own((void**)&G.rayFirst, 4 * np))) return rc;
        {   // sensor-voxel hash table: >= 2 slots per possible point (a frame uses the power-of-two part that holds 2 per VISITED pixel)
            int lg = 10; while ((1ll << lg) < 2 * (long long)np) ++lg;
            G.hlog2 = lg;
            void* tab = nullptr;
            if ((rc = dev_alloc(m, &tab, sizeof(HSlot) << lg, 0))) return rc;         // all slots empty
            S.owned.push_back(tab); G.htab = static_cast<HSlot*>(tab);
            if ((rc = own((void**)&G.act, 4 * np))) return rc;
            if ((rc = own((void**)&G.actx, 4 * np))) return rc;
        }
        if (cfg->texture_enabled) { if ((rc = own((void**)&G.colpix, 8 * np))) return rc; }
        S.header_bytes = 256;
        if ((rc = own(&S.header, S.header_bytes))) return rc;
        G.stats = reinterpret_cast<tsl_frame_stats*>(S.header);
        G.nrays = reinterpret_cast<int*>((char*)S.header + 80);
        G.counters = reinterpret_cast<int*>((char*)S.header + 96);
        if ((rc = own((void**)&G.seg, 8 * (size_t)F.seg_cap))) return rc;
        if ((rc = own((void**)&G.seg_sorted, 8 * (size_t)F.seg_cap))) return rc;
        if ((rc = own((void**)&G.bhist, sizeof(int) * (size_t)m->nb3))) return rc;
        if ((rc = own((void**)&G.bcursor, sizeof(int) * (size_t)m->nb3))) return rc;
        if ((rc = own((void**)&G.boffset, sizeof(int) * (size_t)m->nb3))) return rc;
        if ((rc = own((void**)&G.bnseg, sizeof(int) * (size_t)m->nb3))) return rc;
        if ((rc = own((void**)&G.bslab, sizeof(int) * (size_t)m->nb3))) return rc;
        if ((rc = own((void**)&G.act_b, sizeof(int) * (size_t)(F.max_frame_bricks + 8)))) return rc;
        G.part_cap = F.seg_cap / 256 + F.max_frame_bricks + 8;
        if ((rc = own((void**)&G.part_tab, sizeof(int4) * 4 * (size_t)G.part_cap))) return rc;
        G.unit_cap = TSL_NB * F.max_frame_bricks;
        if (si % TSL_NB == 0) {
            if ((rc = own((void**)&G.unit_tab, sizeof(int4) * 4 * (size_t)G.unit_cap))) return rc;
            if ((rc = own((void**)&G.heavy_tab, sizeof(int4) * (size_t)(F.max_frame_bricks + 8)))) return rc;
        }
        if ((rc = own(&S.sort_temp, m->sort_temp_bytes + 256))) return rc;
        if ((rc = own((void**)&S.Pd, sizeof(FrameParams)))) return rc;
    }
    TSL_HIP(hipStreamSynchronize(m->stream_));                  // the fills ran on the main stream; phase A uses the batch streams
    // the runtime binds a stream to a hardware queue on its first submission (a few hundred microseconds): do that here, not in the first
    // batch that happens to use the slot (a 20-frame burst reaches the third batch slot for the first time inside the measured region)
    for (int bi = 0; bi < TSL_NSTREAMS; ++bi) {
        TSL_HIP(hipMemsetAsync(m->fset[bi * TSL_NB].header, 0, 4, m->batch[bi].st));
        TSL_HIP(hipEventRecord(m->batch[bi].a_done, m->batch[bi].st));
    }
    TSL_HIP(hipMemsetAsync(m->fset[0].header, 0, 4, m->copy_st));
    for (int bi = 0; bi < TSL_NSTREAMS; ++bi) TSL_HIP(hipStreamSynchronize(m->batch[bi].st));
    TSL_HIP(hipStreamSynchronize(m->copy_st));
    for (int k = 0; k < TSL_INFLIGHT; ++k) TSL_HIP(hipEventRecord(m->ring_ev[k], m->stream_));      // (events too are set up on their first record)
    TSL_HIP(hipStreamSynchronize(m->stream_));
    m->scratch_ready = true;
    return TSL_OK;
}

// queue one frame (m->P holds its parameters); the batch is issued when it is full or when anything else needs the map
static int queue_frame(tsl_tsdf* m, const void* depth_dev, const void* xyz_dev, int64_t npts)
{
    FrameParams& P = m->P;
    const int total = xyz_dev ? (int)npts : P.hh * P.ww;
    TSL_REQUIRE(total <= m->F.max_points, "integrate: more pixels/points than max_points");
    { int rc = ensure_frame_scratch(m); if (rc) return rc; }
    if (P.variant == 2) { int rc = check_variant2(m); if (rc) return rc; }
    P.input = xyz_dev ? xyz_dev : depth_dev; P.total = total; P.points = xyz_dev ? 1 : 0;
    {   // the table is empty between frames, so every frame may use its own power-of-two part of it
        int lg = 10; while ((1ll << lg) < 2 * (long long)total) ++lg;
        P.hlog2 = lg < m->fset[0].F.hlog2 ? lg : m->fset[0].F.hlog2;
    }
    TSL_REQUIRE(!P.tex || P.variant == 2, "texture integration needs the brick-binned path (variant 2)");
    const int cap = batch_cap(m);
    { int si = 0; int rc = reserve_slot(m, P.points, &si); if (rc) return rc; }
    m->clean = false;
    m->pend_points = P.points;
    m->pend[m->npend] = P;
    m->pend[m->npend].group = (P.group && P.variant == 2) ? 1 : 0;      // the hash grouping feeds the brick-binned path; the global-atomics variants group by sorting
    m->last_set = m->cur * TSL_NB + m->npend;
    m->npend++;
    // ramp-up: the first batches after the pipeline ran dry are half batches -- a burst gets its first phase A (and with it the
    // whole chain) started sooner, a steady stream is back at full batches after two of them
    if (m->npend >= cap || (cap > 4 && m->ramp < m->ramp_batches && m->npend >= m->ramp_size)) return flush_pending(m);
    // Option "adaptive": frames are only held back while the device has phase-A work to do.  When phase A of the batch issued last has
    // completed (or nothing was issued yet), the queued frames go out at once -- a 30 Hz sensor gets every frame integrated on arrival
    // -- and when the producer outruns the device the batches fill up by themselves.  One event query per queued frame.  Off by
    // default: a burst then starts with a batch of one, and the serial head of the phase-B chain costs it 10 % (20 frames: 13.0 k vs
    // 14.4 k frames/s).
    if (cap > 1 && m->adaptive) {
        const int last = (m->cur + TSL_NBATCH - 1) % TSL_NBATCH;
        const BatchHost& H = m->batch[last];
        if (!H.a_recorded || hipEventQuery(H.a_done) == hipSuccess) return flush_pending(m);
    }
    return TSL_OK;
}

static int sort_temp_size(tsl_tsdf* m, size_t* bytes)   /* called from create: raw stream */
{
    size_t a = 0, b = 0;
    TSL_HIP(rocprim::radix_sort_pairs(nullptr, a, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                      (size_t)m->F.max_points, 0u, 32u, m->stream_));
    TSL_HIP(rocprim::radix_sort_pairs(nullptr, b, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                      (size_t)m->F.max_points, 0u, 64u, m->stream_));
    *bytes = a > b ? a : b;
    return TSL_OK;
}

static void fill_frame_params(tsl_tsdf* m, const double R[9], const double T[3])
{
    const int s = m->active;
    convert_pose(&m->baseR[(size_t)s * 9], &m->baseT[(size_t)s * 3], R, T, m->P.R, m->P.T);   // submap_enabled is always True for DenseTSDF
    m->P.slot = map_slot(m, s);
    m->P.variant = m->variant; m->P.split = (m->variant == 2 && m->split > 8) ? 8 : m->split;      // variant 2: a ray's 16 private segment slots are split over its lanes
}

}  // namespace tsl

using namespace tsl;

// ======================================================================================================
// C-ABI
// ======================================================================================================
extern "C" {

const char* tsl_version(void) { return "taichislam_hip 0.1 (gfx950)"; }
const char* tsl_last_error(void) { return g_err.c_str(); }
int tsl_selftest(int which, int64_t* mismatches)
{
    TSL_REQUIRE(mismatches && which >= 0 && which <= 2, "tsl_selftest: bad argument");
    unsigned long long* bad = nullptr;
    TSL_HIP(hipMalloc((void**)&bad, sizeof(unsigned long long)));
    TSL_HIP(hipMemset(bad, 0, sizeof(unsigned long long)));
    if (which == 2) (void)selftest_seqdiv(bad);
    else hipLaunchKernelGGL(k_selftest, dim3(8192), dim3(256), 0, 0, which, bad);
    unsigned long long h = ~0ull;
    hipError_t e = hipMemcpy(&h, bad, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(bad);
    TSL_HIP(e);
    *mismatches = (int64_t)h;
    return TSL_OK;
}
int tsl_device_count(int* n)
{
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { c = 0; }
    if (n) *n = c;
    return TSL_OK;
}

static int init_handle(tsl_tsdf* m, const tsl_tsdf_cfg* cfg, int device)
{
    m->cfg = *cfg; m->device = device; m->bytes = 0;
    (void)hipGetLastError();                        // (a stale error of the thread is not this handle's: rocPRIM's size queries below would return it)
    TSL_HIP(hipStreamCreateWithFlags(&m->stream_, hipStreamNonBlocking));
    m->overlap = TSL_NB; m->last_set = 0;
    for (auto& S : m->fset) { S.sort_temp = nullptr; S.header = nullptr; S.Pd = nullptr; S.stage_in = nullptr; S.stage_in_bytes = 0; S.stage_tex = nullptr; S.stage_tex_bytes = 0; S.pin = nullptr; S.pin_bytes = 0; S.pin_dev = nullptr; S.copy_in = S.copy_tex = S.copy_tex_off = 0; }
    for (auto& H : m->batch) { H.st = nullptr; H.a_done = nullptr; H.b_done = nullptr; H.p_done = nullptr; H.c_done = nullptr; H.b_pending = false; H.a_recorded = false; H.c_recorded = false; }
    m->frames_issued = 0; m->frames_consumed = 0; m->batch_seq = 0;
    for (int k = 0; k < TSL_INFLIGHT; ++k) { m->ring_ev[k] = nullptr; m->ring_upto[k] = 0; }
    m->cur = 0; m->npend = 0; m->pend_points = 0; m->deferred_rc = 0;
    const int blk = cfg->num_voxel_per_blk_axis;
    m->N = (int)std::ceil(cfg->map_size_xy / cfg->voxel_scale / (double)blk) * blk;          // dense_tsdf.py:24
    m->Nz = (int)std::ceil(cfg->map_size_z / cfg->voxel_scale / (double)blk) * blk;          // dense_tsdf.py:25
    TSL_REQUIRE(m->N > 0 && m->Nz > 0 && m->N <= 32768 && m->Nz <= 32768, "tsl_tsdf_create: map extent must fit int16 voxel indices");
    m->nbx = (m->N + 15) / 16; m->nbz = (m->Nz + 15) / 16; m->nb3 = m->nbx * m->nbx * m->nbz;
    m->nsub = cfg->is_global_map ? 1 : (cfg->max_submap_num > 0 ? cfg->max_submap_num : 1); // dense_tsdf.py:86-88
    m->npose = cfg->max_submap_num > m->nsub ? cfg->max_submap_num : m->nsub;
    {   // sensor-centred scratch grid  dense_tsdf.py:67-70
        int grp = (int)(3.2 * cfg->max_ray_length / (double)blk / cfg->voxel_scale);
        if (grp < 1) grp = 1;
        const int ext = blk * grp;
        int off = -ext / 2; if ((-ext) % 2 != 0) off -= 1;                                   // Python floor division
        m->pcl_lo = off; m->pcl_ext = ext;
        int bits = 1; while ((1 << bits) < ext) ++bits;
        m->pcl_bits = bits;
        TSL_REQUIRE(bits <= 20, "tsl_tsdf_create: sensor grid too large");
    }
    FrameParams& P = m->P; std::memset(&P, 0, sizeof(P));
    for (int i = 0; i < 3; ++i) P.R[i * 4] = 1.0f;
    P.vs = (float)cfg->voxel_scale;
    {   // RN(1/vs): take the float neighbour whose product with vs is closest to 1 (products of two floats are exact in double)
        const float y0 = (float)(1.0 / (double)P.vs);
        float best = y0; double err = std::fabs(1.0 - (double)y0 * (double)P.vs);
        const float cand[2] = { std::nextafterf(y0, 0.0f), std::nextafterf(y0, INFINITY) };
        for (float c : cand) { const double e = std::fabs(1.0 - (double)c * (double)P.vs); if (e < err) { err = e; best = c; } }
        P.rvs = best; P.fastdiv = 0;
    }
    P.thr_max = (float)(cfg->max_ray_length * 1000.0); P.thr_min = (float)(cfg->min_ray_length * 1000.0);
    P.max_ray_f = (float)cfg->max_ray_length;
    P.max_steps_f = (float)(cfg->max_ray_length / cfg->voxel_scale);
    P.internal_f = (float)cfg->internal_voxels;
    P.pcl_lo = m->pcl_lo; P.pcl_ext = m->pcl_ext; P.pcl_bits = m->pcl_bits; P.pcl_blk = blk; P.seq = 0;
    P.step = cfg->recast_step; P.same_proj = cfg->color_same_proj;
    m->surf_thres = (float)(cfg->voxel_scale * 1.8);                                        // dense_tsdf.py:39
    m->disp_floor = (float)cfg->disp_floor; m->disp_ceiling = (float)cfg->disp_ceiling;
    m->baseR.assign((size_t)m->npose * 9, 0.0); m->baseT.assign((size_t)m->npose * 3, 0.0);
    m->baseRf.assign((size_t)m->npose * 9, 0.0f); m->baseTf.assign((size_t)m->npose * 3, 0.0f);
    // identity default instead of the reference's zero matrices (mapping_common.py:106)
    for (int s = 0; s < m->npose; ++s) for (int i = 0; i < 3; ++i) { m->baseR[(size_t)s * 9 + i * 4] = 1.0; m->baseRf[(size_t)s * 9 + i * 4] = 1.0f; }
    std::memset(m->gbaseR, 0, sizeof(m->gbaseR)); std::memset(m->gbaseT, 0, sizeof(m->gbaseT));
    for (int i = 0; i < 3; ++i) m->gbaseR[i * 4] = 1.0;
    m->P.group = 1; m->phases = 3; m->wg = 512; m->spt = 2; m->adaptive = 0; m->ramp_batches = 2; m->ramp_size = 4; m->bgrid = 75; m->ugrid = 75; m->pgrid = 25; m->split_launch = 0; m->chunks = 4; m->unit_max = 1024 * TSL_NB; m->unit_half = 1 << 20; m->unit_floor = 4096; m->batch_gen = 0;
    // (unit_half >= unit: the middle tier of k_plan is off by default -- measured neutral-to-negative once the brick kernel runs on 75 % of the slots)
    { hipDeviceProp_t pr; TSL_HIP(hipGetDeviceProperties(&pr, device)); m->ncu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    m->active = 0; m->variant = 2; m->split = 2; m->semantics = 0; m->seq_impl = 1; m->seq_ready = false; m->seq_bytes0 = 0; m->seq_d = nullptr; m->seq_tuple_cap = 0;
    m->prof_on = false; m->prof_open = false; m->prof_group = false; m->prof_mask = ~0u; std::memset(m->prof_ms, 0, sizeof(m->prof_ms)); std::memset(m->prof_n, 0, sizeof(m->prof_n));
    m->sort_temp = nullptr; m->sort_temp_bytes = 0;
    m->xbuf = nullptr; m->xbuf_bytes = 0; m->mesh_v = m->mesh_n = m->mesh_c = nullptr; m->mesh_count = nullptr; m->mesh_cap = 0; m->mesh_flags = nullptr;
    m->fuse_acc = nullptr; m->fuse_cnt = nullptr; m->fuse_cacc = nullptr; m->fuse_dirty = false; m->mrg_nunion = -1;
    m->esdf = nullptr; m->esdf_par = nullptr; m->esdf_ok = nullptr; m->esdf_mode = 0; m->esdf_grid = 0; m->fuse_direct = false; m->merge_exchange = 0; m->mrg_racc = m->mrg_rcnt = m->mrg_rec = nullptr; m->mrg_racc_bytes = m->mrg_rcnt_bytes = m->mrg_rec_bytes = 0; m->esdf_orphans = 0; m->esdf_last = nullptr; m->esdf_in = nullptr; m->esdf_read = nullptr; m->esdf_overlap = true; m->esdf_ctr_idx = 0; for (int k = 0; k < 2; ++k) { m->fseq_keys[k] = m->fseq_vals[k] = nullptr; m->fseq_bytes[k] = m->fseq_vbytes[k] = 0; } m->fseq_temp = nullptr; m->fseq_tbytes = 0; m->fseq_ctr = nullptr; m->esdf_gate = nullptr; m->esdf_gate_ev = nullptr; m->esdf_gate_set = false; m->esdf_gate_mask = 0; m->esdf_valid = false; m->esdf_force_full = false; m->pose_dev = nullptr;

    // ---- map storage ----
    MapDev& M = m->M; std::memset(&M, 0, sizeof(M));
    M.N = m->N; M.Nz = m->Nz; M.nbx = m->nbx; M.nbz = m->nbz; M.nb3 = m->nb3; M.nsub = m->nsub; M.hN = m->N / 2; M.hNz = m->Nz / 2;
    // default pool: one full 512^3 volume of bricks (0.8 GB) per handle, four of them when the handle holds several submaps
    // (SubmapMapping's default collection of up to 1024 submaps shares the pool); max_bricks overrides
    int64_t want = cfg->max_bricks > 0 ? cfg->max_bricks : 32768 * (int64_t)(m->nsub > 4 ? 4 : (m->nsub > 1 ? m->nsub : 1));
    const int64_t all = (int64_t)m->nsub * m->nb3;
    if (want > all) want = all;
    M.max_bricks = (int)want;
    int rc;
    if ((rc = dev_alloc(m, (void**)&M.table, sizeof(int) * (size_t)all, 0xff))) return rc;
    if ((rc = dev_alloc(m, (void**)&M.tw, sizeof(uint32_t) * (size_t)want * TSL_BRK3, 0))) return rc;
    if ((rc = dev_alloc(m, (void**)&M.obs, (size_t)want * TSL_BRK3, 0))) return rc;
    if ((rc = dev_alloc(m, (void**)&M.occ, (size_t)want * TSL_BRK3, 0))) return rc;
    if (cfg->texture_enabled) { if ((rc = dev_alloc(m, (void**)&M.col, sizeof(uint16_t) * 4 * (size_t)want * TSL_BRK3, 0))) return rc; }
    if ((rc = dev_alloc(m, (void**)&M.owner, sizeof(int) * (size_t)want, 0))) return rc;
    if ((rc = dev_alloc(m, (void**)&M.touch, (size_t)want, 0))) return rc;
    if ((rc = dev_alloc(m, (void**)&M.pool_top, sizeof(int) * 4, 0))) return rc;
    M.err = M.pool_top + 1;

    // ---- frame scratch sizes (the buffers themselves are allocated by the first integrate call: a global map that only
    //      receives fusions or imports never pays for them) ----
    FrameDev& F = m->F; std::memset(&F, 0, sizeof(F));
    F.max_points = cfg->max_points > 0 ? cfg->max_points : 640 * 480;
    F.max_frame_bricks = cfg->max_frame_bricks > 0 ? cfg->max_frame_bricks : 4096;
    if (F.max_frame_bricks > M.max_bricks) F.max_frame_bricks = M.max_bricks;
    {   // ray segments: ~ (steps/16 + 3 axis crossings) per ray; 32 per point is a generous bound
        const int64_t cap = (int64_t)F.max_points * 32;
        F.seg_cap = (int)(cap > (1ll << 30) ? (1ll << 30) : cap);
    }
    if ((rc = sort_temp_size(m, &m->sort_temp_bytes))) return rc;
    // The pipeline's events order streams of ONE device against each other (and tell the host that a batch has finished, nothing about memory): they
    // need no system-scope fence -- the cache write-back + invalidate a default event adds when it is recorded sat between k_apply_slab of a batch and
    // the brick kernel of the next (round 5, TSL_EV_SYS=1 restores the default for A/B).  Results reach the host through stream synchronisations and copies.
#ifdef TSL_TEST_HOOKS
    static const bool ev_sys = std::getenv("TSL_EV_SYS") != nullptr;
#else
    constexpr bool ev_sys = false;
#endif
    const unsigned evf = hipEventDisableTiming | (ev_sys ? 0u : (unsigned)hipEventDisableSystemFence);
    for (int bi = 0; bi < TSL_NBATCH; ++bi) {
        BatchHost& H = m->batch[bi];
        // more batch slots than streams: the device runs four hardware queues efficiently (main + TSL_NSTREAMS), a slot beyond that shares
        // the stream of slot bi - TSL_NSTREAMS (its phase A is ordered behind that slot's, which is two or three batches older)
        if (bi < TSL_NSTREAMS) TSL_HIP(hipStreamCreateWithFlags(&H.st, hipStreamNonBlocking)); else H.st = m->batch[bi % TSL_NSTREAMS].st;
        TSL_HIP(hipEventCreateWithFlags(&H.a_done, evf));
        TSL_HIP(hipEventCreateWithFlags(&H.p_done, evf));
        TSL_HIP(hipEventCreateWithFlags(&H.c_done, evf));
    }
    for (int k = 0; k < TSL_INFLIGHT; ++k) TSL_HIP(hipEventCreateWithFlags(&m->ring_ev[k], evf));
    TSL_HIP(hipStreamCreateWithFlags(&m->copy_st, hipStreamNonBlocking));
    TSL_HIP(hipHostMalloc((void**)&m->h_stats, sizeof(tsl_frame_stats), hipHostMallocDefault));
    TSL_HIP(hipHostMalloc((void**)&m->h_ints, sizeof(long long) * 256, hipHostMallocDefault));
    std::memset(m->h_stats, 0, sizeof(tsl_frame_stats));

    // ---- export buffers  dense_tsdf.py:53-60,129-134 ----
    m->max_disp = cfg->max_disp_particles > 0 ? cfg->max_disp_particles : 1024 * 1024;
    if ((rc = dev_alloc(m, (void**)&m->exp_xyz, sizeof(float) * 3 * (size_t)m->max_disp, 0))) return rc;
    if ((rc = dev_alloc(m, (void**)&m->exp_rgb, sizeof(float) * 3 * (size_t)m->max_disp, 0))) return rc;
    if ((rc = dev_alloc(m, (void**)&m->exp_val, sizeof(float) * (size_t)m->max_disp, 0))) return rc;
    if ((rc = dev_alloc(m, (void**)&m->num_particles, sizeof(int) * 4, 0))) return rc;
    if ((rc = dev_alloc(m, (void**)&m->colormap, sizeof(float) * 3 * 1024, 0))) return rc;
    if ((rc = dev_alloc(m, (void**)&m->pose_dev, sizeof(float) * 12 * (size_t)m->npose, 0))) return rc;
    {   // enable the fast exact division only if the device proves it equal to IEEE division for this voxel size
        unsigned long long* bad = reinterpret_cast<unsigned long long*>(m->num_particles + 2);
        TSL_HIP(hipMemsetAsync(bad, 0, sizeof(unsigned long long), m->stream_));
        hipLaunchKernelGGL(k_verify_div, dim3(8192), dim3(256), 0, m->stream_, P.vs, P.rvs, bad);
        unsigned long long nbad = 1;
        TSL_HIP(hipMemcpyAsync(&nbad, bad, sizeof(nbad), hipMemcpyDeviceToHost, m->stream_));
        TSL_HIP(hipStreamSynchronize(m->stream_));
        TSL_HIP(hipMemsetAsync(bad, 0, sizeof(unsigned long long), m->stream_));
        P.fastdiv = nbad == 0 ? 1 : 0;
    }
    TSL_HIP(hipStreamSynchronize(m->stream_));
    return TSL_OK;
}

int tsl_tsdf_create(const tsl_tsdf_cfg* cfg, int device, tsl_tsdf** out)
{
    TSL_REQUIRE(cfg && out, "tsl_tsdf_create: null argument");
    TSL_REQUIRE(cfg->voxel_scale > 0 && cfg->num_voxel_per_blk_axis >= 1 && cfg->recast_step >= 1, "tsl_tsdf_create: bad config");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available"); return TSL_ERR_NO_DEVICE; }
    TSL_REQUIRE(device >= 0 && device < ndev, "tsl_tsdf_create: bad device index");
    TSL_HIP(hipSetDevice(device));
    tsl_tsdf* m = new tsl_tsdf();          // value-initialised: every pointer starts null, so a partly built handle can be destroyed
    const int rc = init_handle(m, cfg, device);
    if (rc) { const std::string keep = g_err; tsl_tsdf_destroy(m); g_err = keep; return rc; }
    *out = m;
    return TSL_OK;
}

void tsl_tsdf_destroy(tsl_tsdf* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream_) (void)hipStreamSynchronize(m->stream_);
    for (auto& H : m->batch) if (H.st) (void)hipStreamSynchronize(H.st);
    if (m->stream_) (void)hipStreamSynchronize(m->stream_);
    (void)hipDeviceSynchronize();
    for (int bi = 0; bi < TSL_NBATCH; ++bi) {
        BatchHost& H = m->batch[bi];
        if (H.st && bi < TSL_NSTREAMS) (void)hipStreamDestroy(H.st);
        if (H.a_done) (void)hipEventDestroy(H.a_done);
        if (H.p_done) (void)hipEventDestroy(H.p_done);
        if (H.c_done) (void)hipEventDestroy(H.c_done);
    }
    for (auto& S : m->fset) {

        for (void* p : S.owned) if (p) (void)hipFree(p);
        if (S.stage_in) (void)hipFree(S.stage_in);
        if (S.stage_tex) (void)hipFree(S.stage_tex);
        if (S.pin) (void)hipHostFree(S.pin);

    }
    esdf_release(m);
    seq_release(m);
    void* ptrs[] = { m->M.table, m->M.tw, m->M.obs, m->M.occ, m->M.col, m->M.owner, m->M.pool_top, m->F.slot_tab, m->F.touched, m->F.touched_b, m->F.acc, m->F.accw, m->F.dbg,
                     m->exp_xyz, m->exp_rgb, m->exp_val, m->num_particles, m->colormap, m->pose_dev, m->xbuf,
                     m->mesh_v, m->mesh_n, m->mesh_c, m->mesh_count, m->mesh_flags, m->esdf, m->esdf_fl, m->esdf_region, m->esdf_list, m->esdf_queue, m->esdf_ctr, m->esdf_inq, m->esdf_nbr, m->esdf_par, m->esdf_ok, m->esdf_note, m->fseq_keys[0], m->fseq_keys[1], m->fseq_vals[0], m->fseq_vals[1], m->fseq_temp, m->fseq_ctr, m->esdf_exp_xyz, m->esdf_exp_val, m->esdf_exp_count, m->M.touch, m->fuse_acc, m->fuse_cnt, m->fuse_cacc, m->seq_keys[0], m->seq_keys[1], m->seq_vals[0], m->seq_vals[1], m->seq_ctr, m->seq_temp,
                     m->mrg_mask, m->mrg_list, m->mrg_pacc, m->mrg_pcnt, m->mrg_racc, m->mrg_rcnt, m->mrg_rec };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (m->h_stats) (void)hipHostFree(m->h_stats);
    if (m->h_ints) (void)hipHostFree(m->h_ints);
    for (auto& s : m->prof) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto& e : m->prof_free) (void)hipEventDestroy(e);
    for (auto& e : m->in_ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : m->ring_ev) if (e) (void)hipEventDestroy(e);
    if (m->copy_st) (void)hipStreamDestroy(m->copy_st);
    if (m->stream_) (void)hipStreamDestroy(m->stream_);
    delete m;
}

int tsl_tsdf_get_dims(const tsl_tsdf* m, int32_t* N, int32_t* Nz, int32_t* bxy, int32_t* bz)
{
    TSL_REQUIRE(m, "null handle");
    const int blk = m->cfg.num_voxel_per_blk_axis;
    if (N) *N = m->N; if (Nz) *Nz = m->Nz;
    if (bxy) *bxy = m->N / blk; if (bz) *bz = m->Nz / blk;                                   // dense_tsdf.py:27-28
    return TSL_OK;
}
// The device reports exhausted capacity through one sticky word (bit 0 brick pool, 1 frame bricks / parts, 2 ray segments,
// 3 sensor voxel too crowded).  Every call that synchronises reads it, turns it into TSL_ERR_CAPACITY once and clears it: a frame or
// a brick that was dropped is never dropped silently.
static int take_dev_err(tsl_tsdf* m)
{
    TSL_HIP(hipMemcpyAsync(&m->h_ints[28], m->M.err, sizeof(int), hipMemcpyDeviceToHost, m->stream_));
    TSL_HIP(hipStreamSynchronize(m->stream_));
    const int e = m->h_ints[28];
    if (!e) return TSL_OK;
    set_error(std::string("device capacity exhausted:") + ((e & 1) ? " brick pool (max_bricks)" : "") + ((e & 2) ? " frame scratch (max_frame_bricks)" : "") +
              ((e & 4) ? " ray segments / ray-step tuples of the sequential semantics (seq_tuple_cap)" : "") + ((e & 8) ? " more than 16384 points in one sensor voxel" : "") + "; the affected frames / bricks were not integrated");
    TSL_HIP(hipMemsetAsync(m->M.err, 0, sizeof(int), m->stream_));
    return TSL_ERR_CAPACITY;
}
int tsl_tsdf_sync(tsl_tsdf* m)
{
    TSL_REQUIRE(m, "null handle");
    // nothing was queued or enqueued through this handle since the last sync (every entry point takes the main stream through ms(),
    // which marks the handle): a second sync in a row -- bench.py's barrier after its own sync -- costs nothing
    if (m->clean && m->npend == 0 && !m->deferred_rc) return TSL_OK;
    TSL_HIP(hipSetDevice(m->device));
    int rc = flush_pending(m);
    for (auto& H : m->batch) if (H.st) TSL_HIP(hipStreamSynchronize(H.st));
    if (m->copy_st) TSL_HIP(hipStreamSynchronize(m->copy_st));
    const int ec = take_dev_err(m);                // synchronises the main stream
    m->frames_consumed = m->frames_issued;         // everything issued has run
    if (!rc && m->deferred_rc) { rc = m->deferred_rc; }
    m->deferred_rc = 0;
    m->clean = true;
    return rc ? rc : ec;
}
int tsl_tsdf_memory_bytes(const tsl_tsdf* m, int64_t* b) { TSL_REQUIRE(m && b, "null"); *b = m->bytes; return TSL_OK; }

// A batch that could not be issued (flush_pending inside ms()) leaves its error in deferred_rc: every entry point that hands results to the caller
// reports it -- a reader must never return the map of frames that were silently dropped (round 5: an export of 0 voxels with status OK).
static int take_deferred(tsl_tsdf* m) { const int rc = m->deferred_rc; m->deferred_rc = 0; return rc; }
static int read_int(tsl_tsdf* m, const int* dev, int* out)
{
    TSL_HIP(hipMemcpyAsync(m->h_ints, dev, sizeof(int), hipMemcpyDeviceToHost, ms(m)));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    *out = m->h_ints[0];
    return take_deferred(m);
}
static int check_dev_err(tsl_tsdf* m) { (void)ms(m); return take_dev_err(m); }
int tsl_tsdf_get_option(tsl_tsdf* m, const char* name, int* value)
{
    TSL_REQUIRE(m && name && value, "null");
    if (!std::strcmp(name, "group")) { *value = m->P.group; return TSL_OK; }
    if (!std::strcmp(name, "esdf_mode")) { *value = m->esdf_mode; return TSL_OK; }
    if (!std::strcmp(name, "fuse_window_misses")) { return read_int(m, m->M.pool_top + 2, value); }      // corner splats of k_fuse_splat_lds that fell outside the 15^3 window since the map was created
    if (!std::strcmp(name, "esdf_orphans")) { int rc = esdf_finish(m); if (rc) return rc; *value = (int)m->esdf_orphans; return TSL_OK; }      // lowered voxels whose supporting neighbour was not found (0, or the wavefront has a hole)
    if (!std::strcmp(name, "semantics")) { *value = m->semantics; return TSL_OK; }
    if (!std::strcmp(name, "seq_impl")) { *value = m->seq_impl; return TSL_OK; }
    if (!std::strcmp(name, "last_heavy_bricks") || !std::strcmp(name, "last_slab_slots")) {
        // developer statistics of the batch issued last: bricks walked in parts / slab slots (= parts) they used
        TSL_REQUIRE(m->scratch_ready, "nothing integrated yet");
        int rc = tsl_tsdf_sync(m); if (rc) return rc;
        const int bi = (m->cur + TSL_NBATCH - 1) % TSL_NBATCH;
        int v[40];
        TSL_HIP(hipMemcpy(v, m->fset[bi * TSL_NB].F.counters, sizeof(v), hipMemcpyDeviceToHost));
        *value = name[5] == 'h' ? v[14] : v[13];
        return TSL_OK;
    }
    if (!std::strcmp(name, "seq_longest_run")) {
        // sequential semantics, developer statistic: the most updates any one voxel received in a frame of the batch issued last, summed over the batch's
        // frames -- the dependent chain that bounds the replay of that batch (one wave applies it, ~43 cycles per update)
        TSL_REQUIRE(m->scratch_ready, "nothing integrated yet");
        int rc = tsl_tsdf_sync(m); if (rc) return rc;
        const int bi = (m->cur + TSL_NBATCH - 1) % TSL_NBATCH;
        long long sum = 0;
        for (int q = 0; q < m->last_batch_n; ++q) { int v[40]; TSL_HIP(hipMemcpy(v, m->fset[bi * TSL_NB + q].F.counters, sizeof(v), hipMemcpyDeviceToHost)); sum += (unsigned)v[31]; }
        *value = (int)(sum > 0x7fffffffll ? 0x7fffffffll : sum);
        return TSL_OK;
    }
    if (!std::strcmp(name, "seq_long_voxels")) {
        // sequential semantics, developer statistic: voxels of the batch issued last that were replayed by a wave of their own (>= 64 updates in a frame)
        TSL_REQUIRE(m->scratch_ready, "nothing integrated yet");
        int rc = tsl_tsdf_sync(m); if (rc) return rc;
        const int bi = (m->cur + TSL_NBATCH - 1) % TSL_NBATCH;
        int v[40]; TSL_HIP(hipMemcpy(v, m->fset[bi * TSL_NB].F.counters, sizeof(v), hipMemcpyDeviceToHost));
        *value = v[28] + v[30];
#ifdef TSL_SEQ_DBG
        fprintf(stderr, "seq dbg (long role, last batch): evaluations %d, commits that moved T %d, commits that moved W only %d\n", v[32], v[33], v[34]);
#endif
        return TSL_OK;
    }
    if (!std::strcmp(name, "fastdiv")) { *value = m->P.fastdiv; return TSL_OK; }
    if (!std::strcmp(name, "seq_verify_mismatches")) {      // developer aid (TSL_SEQ_VERIFY=1): -1 when off; the records go to stderr
        int rc = tsl_tsdf_sync(m); if (rc) return rc;
        std::vector<int> buf(4 + 4 * 64);
        rc = seq_verify_report(m, buf.data(), (int)buf.size()); if (rc) return rc;
        *value = buf[0];
        for (int e = 0; e < buf[0] && e < 64; ++e)
            fprintf(stderr, "seq_verify: batch %d frame %d stage %d slot %d what %d T %d  (stage 3: voxel %d expected %d found %d)\n", buf[4 + 4 * e], buf[5 + 4 * e] & 255, buf[5 + 4 * e] >> 8, buf[6 + 4 * e], buf[7 + 4 * e] & 15, buf[7 + 4 * e] >> 4,
                    buf[7 + 4 * e] & 4095, (buf[7 + 4 * e] >> 12) & 1023, (buf[7 + 4 * e] >> 22) & 1023);
        return TSL_OK;
    }
    if (!std::strcmp(name, "batch_shape_hash")) { *value = (int)(m->shape_hash & 0x7fffffffu); return TSL_OK; }
    if (!std::strcmp(name, "dry_launches")) { *value = m->dry_launches; return TSL_OK; }
    if (!std::strcmp(name, "overlapped_launches")) { *value = (int)m->batch_seq - m->dry_launches; return TSL_OK; }      // batches whose phase A was issued while phase B of the batch before was still pending
    if (!std::strcmp(name, "variant")) { *value = m->variant; return TSL_OK; }
    if (!std::strcmp(name, "split")) { *value = m->split; return TSL_OK; }
    set_error("unknown option"); return TSL_ERR_ARG;
}
int tsl_tsdf_bricks_in_use(tsl_tsdf* m, int32_t* n)
{
    TSL_REQUIRE(m && n, "null"); TSL_HIP(hipSetDevice(m->device));
    int v = 0; int rc = read_int(m, m->M.pool_top, &v); if (rc) return rc;
    *n = v > m->M.max_bricks ? m->M.max_bricks : v;
    return TSL_OK;
}

int tsl_tsdf_reset(tsl_tsdf* m)
{
    TSL_REQUIRE(m, "null handle"); TSL_HIP(hipSetDevice(m->device));
    const int pending = tsl_tsdf_sync(m);          // a capacity error of the discarded contents is still reported, after the reset
    if (pending && pending != TSL_ERR_CAPACITY) return pending;
    int used = 0; int rc = tsl_tsdf_bricks_in_use(m, &used); if (rc) return rc;
    m->esdf_valid = false;
    if (used > 0) hipLaunchKernelGGL(k_reset_bricks, dim3(used < 4096 ? used : 4096), dim3(256), 0, ms(m), m->M, used);
    TSL_HIP(hipMemsetAsync(m->M.pool_top, 0, sizeof(int), ms(m)));
    TSL_HIP(hipGetLastError());
    return pending;
}

int tsl_tsdf_set_intrinsics(tsl_tsdf* m, const double Kd[9], const double Kc[9])
{
    TSL_REQUIRE(m, "null handle");
    if (Kd) { m->P.fx = (float)Kd[0]; m->P.fy = (float)Kd[4]; m->P.cx = (float)Kd[2]; m->P.cy = (float)Kd[5]; }
    if (Kc) { m->P.fxc = (float)Kc[0]; m->P.fyc = (float)Kc[4]; m->P.cxc = (float)Kc[2]; m->P.cyc = (float)Kc[5]; }
    return TSL_OK;
}
int tsl_tsdf_set_base_pose(tsl_tsdf* m, const double R[9], const double T[3])
{ TSL_REQUIRE(m && R && T, "null"); std::memcpy(m->gbaseR, R, 72); std::memcpy(m->gbaseT, T, 24); return TSL_OK; }
int tsl_tsdf_set_base_pose_submap(tsl_tsdf* m, int sid, const double R[9], const double T[3])
{
    TSL_REQUIRE(m && R && T, "null"); TSL_REQUIRE(sid >= 0 && sid < m->npose, "set_base_pose_submap: submap id out of range");
    std::memcpy(&m->baseR[(size_t)sid * 9], R, 72); std::memcpy(&m->baseT[(size_t)sid * 3], T, 24);
    for (int a = 0; a < 9; ++a) m->baseRf[(size_t)sid * 9 + a] = (float)R[a];
    for (int a = 0; a < 3; ++a) m->baseTf[(size_t)sid * 3 + a] = (float)T[a];
    return TSL_OK;
}
int tsl_tsdf_get_active_submap(const tsl_tsdf* m, int32_t* sid) { TSL_REQUIRE(m && sid, "null"); *sid = m->active; return TSL_OK; }
int tsl_tsdf_set_active_submap(tsl_tsdf* m, int32_t sid)
{ TSL_REQUIRE(m, "null"); TSL_REQUIRE(sid >= 0 && sid < m->npose, "set_active_submap: id out of range"); m->active = sid; return TSL_OK; }
int tsl_tsdf_set_colormap(tsl_tsdf* m, const float* rgb)
{
    TSL_REQUIRE(m && rgb, "null"); TSL_HIP(hipSetDevice(m->device));
    TSL_HIP(hipMemcpyAsync(m->colormap, rgb, sizeof(float) * 3 * 1024, hipMemcpyHostToDevice, ms(m)));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    return TSL_OK;
}

// packed: the buffer is a staged host image that holds the visited pixels only, ww = w / recast_step per visited row (a parameter, not a flag
// on the handle: a call that fails half way cannot leave it behind for the next one -- ADVICE r3); otherwise the caller's full image
static int integrate_depth_dev_impl(tsl_tsdf* m, const double R[9], const double T[3], const void* depth_dev, int h, int w,
                                    const void* tex_dev, int th, int tw, bool packed)
{
    TSL_REQUIRE(m && R && T && depth_dev, "integrate_depth: null argument");
    TSL_REQUIRE(h > 0 && w > 0, "integrate_depth: bad image size");
    TSL_REQUIRE(!(m->cfg.texture_enabled && tex_dev) || true, "");
    TSL_REQUIRE(m->active < m->nsub || m->cfg.is_global_map, "integrate: active submap beyond max_submap_num");
    TSL_HIP(hipSetDevice(m->device));
    fill_frame_params(m, R, T);
    FrameParams& P = m->P;
    P.H = h; P.W = w;
    P.hh = (int)((float)h / (float)P.step); P.ww = (int)((float)w / (float)P.step);           // dense_tsdf.py:192,194
    P.rstride = packed ? w / P.step : P.step * w;
    P.cstride = packed ? 1 : P.step;
    P.th = th; P.tw = tw; P.tex = (m->cfg.texture_enabled && tex_dev) ? 1 : 0; P.tex_input = (const uint8_t*)tex_dev;
    TSL_REQUIRE(!P.tex || (th > 0 && tw > 0 && (!P.same_proj || (th >= h && tw >= w))), "integrate_depth: texture smaller than the depth image");
    m->h_stats->p_used = (int64_t)P.hh * P.ww;
    return queue_frame(m, depth_dev, nullptr, 0);
}
int tsl_tsdf_integrate_depth_dev(tsl_tsdf* m, const double R[9], const double T[3], const void* depth_dev, int h, int w,
                                 const void* tex_dev, int th, int tw)
{ return integrate_depth_dev_impl(m, R, T, depth_dev, h, w, tex_dev, th, tw, false); }

int tsl_tsdf_integrate_depth(tsl_tsdf* m, const double R[9], const double T[3], const uint16_t* depth, int h, int w,
                             const uint8_t* tex, int th, int tw)
{
    TSL_REQUIRE(m && depth, "integrate_depth: null argument"); TSL_REQUIRE(h > 0 && w > 0, "integrate_depth: bad image size");
    TSL_HIP(hipSetDevice(m->device));
    int si = 0; int rc = reserve_slot(m, 0, &si); if (rc) return rc;
    const bool use_tex = tex && m->cfg.texture_enabled && th > 0 && tw > 0;
    void *ddev = nullptr, *tdev = nullptr;
    // only the pixels the kernel visits (every recast_step-th of every recast_step-th row, dense_tsdf.py:192-195) cross the bus: a quarter of a
    // 640x480 image at recast_step 2
    const int step = m->P.step, hh = (int)((float)h / (float)step);
    rc = stage_host(m, si, depth, (size_t)w * sizeof(uint16_t), step > 1 ? hh : h, (size_t)(step > 1 ? step : 1) * w * sizeof(uint16_t),
                    use_tex ? tex : nullptr, use_tex ? (size_t)th * tw * 3 : 0, &ddev, &tdev, step); if (rc) return rc;
    return integrate_depth_dev_impl(m, R, T, ddev, h, w, tdev, th, tw, step > 1);
}

int tsl_tsdf_integrate_points_dev(tsl_tsdf* m, const double R[9], const double T[3], const void* xyz_dev, const void* rgb_dev, int64_t n)
{
    TSL_REQUIRE(m && R && T, "integrate_points: null argument"); TSL_REQUIRE(n >= 0 && (n == 0 || xyz_dev), "integrate_points: bad input");
    TSL_HIP(hipSetDevice(m->device));
    fill_frame_params(m, R, T);
    m->P.tex = (m->cfg.texture_enabled && rgb_dev) ? 1 : 0; m->P.tex_input = (const uint8_t*)rgb_dev;
    m->h_stats->p_used = n;
    return queue_frame(m, nullptr, xyz_dev, n);
}

int tsl_tsdf_integrate_points(tsl_tsdf* m, const double R[9], const double T[3], const float* xyz, const uint8_t* rgb, int64_t n)
{
    TSL_REQUIRE(m, "null handle"); TSL_REQUIRE(n >= 0 && (n == 0 || xyz), "integrate_points: bad input");
    TSL_HIP(hipSetDevice(m->device));
    int si = 0; int rc = reserve_slot(m, 1, &si); if (rc) return rc;
    const bool use_tex = rgb && m->cfg.texture_enabled && n;
    void *xdev = nullptr, *cdev = nullptr;
    rc = stage_host(m, si, xyz, (size_t)n * 3 * sizeof(float), 1, 0, use_tex ? rgb : nullptr, use_tex ? (size_t)n * 3 : 0, &xdev, &cdev); if (rc) return rc;
    return tsl_tsdf_integrate_points_dev(m, R, T, xdev, cdev, n);
}

/* HIP stream that will read the input of the NEXT integrate_*_dev call (phase A of the batch the frame joins).  A caller that
 * produces its device buffers on its own stream passes that stream as `producer`: the reading stream is made to wait for what the
 * producer has queued so far (one event record + one stream wait, events are cached).  producer = NULL: the legacy default stream;
 * `ordered` = 0 skips the ordering (the caller guarantees the buffers are complete). */
int tsl_tsdf_input_stream(tsl_tsdf* m, int points, int ordered, void* producer, void** hip_stream)
{
    TSL_REQUIRE(m, "input_stream: null argument"); TSL_HIP(hipSetDevice(m->device));
    int si = 0; int rc = reserve_slot(m, points ? 1 : 0, &si); if (rc) return rc;        // may issue the queued frames first
    hipStream_t consumer = m->overlap == 0 ? m->stream_ : m->batch[si / TSL_NB].st;
    if (hip_stream) *hip_stream = (void*)consumer;
    // The ordering is established when the batch is ISSUED (launch_batch_t): one event per distinct producer stream and batch instead
    // of one per frame.  An event recorded later on the producer's stream also covers the work that was queued there before this call.
    if (ordered && (hipStream_t)producer != consumer) {
        bool seen = false;
        for (int k = 0; k < m->nproducers; ++k) seen = seen || m->producers[k] == (hipStream_t)producer;
        if (!seen) {
            if (m->nproducers == 4) { rc = flush_pending(m); if (rc) return rc; }          // a fifth stream inside one batch: issue what is queued
            m->producers[m->nproducers++] = (hipStream_t)producer;
        }
    }
    return TSL_OK;
}

/* one call per frame for callers that hand over device buffers filled on a stream of their own: tsl_tsdf_input_stream(ordered) +
 * tsl_tsdf_integrate_depth_dev + tsl_tsdf_frames_consumed */
int tsl_tsdf_integrate_depth_stream(tsl_tsdf* m, const double R[9], const double T[3], const void* depth_dev, int h, int w,
                                    const void* tex_dev, int th, int tw, void* producer, int64_t* queued_total, int64_t* consumed)
{
    int rc = tsl_tsdf_input_stream(m, 0, 1, producer, nullptr); if (rc) return rc;
    rc = tsl_tsdf_integrate_depth_dev(m, R, T, depth_dev, h, w, tex_dev, th, tw); if (rc) return rc;
    return tsl_tsdf_frames_consumed(m, queued_total, consumed);
}

/* frames queued but not yet issued to the device (0 right after a batch went out) */
int tsl_tsdf_queued_frames(const tsl_tsdf* m, int32_t* n) { TSL_REQUIRE(m && n, "null"); *n = m->npend; return TSL_OK; }

/* frames (counted since the handle was created) whose input buffers the device has finished reading: host-side bookkeeping of
 * the back-pressure ring (launch_batch_t) and of tsl_tsdf_sync, no device query */
int tsl_tsdf_frames_consumed(tsl_tsdf* m, int64_t* queued_total, int64_t* consumed)
{
    TSL_REQUIRE(m, "null");
    if (queued_total) *queued_total = m->frames_issued + m->npend;
    if (consumed) *consumed = m->frames_consumed;
    return TSL_OK;
}

int tsl_tsdf_last_frame_stats(tsl_tsdf* m, tsl_frame_stats* out)
{
    TSL_REQUIRE(m && out, "null"); TSL_HIP(hipSetDevice(m->device));
    const int64_t used = m->h_stats->p_used;
    tsl_frame_stats tmp;
    if (!m->scratch_ready) { std::memset(out, 0, sizeof(*out)); return TSL_OK; }      // nothing was ever integrated
    { int rc2 = tsl_tsdf_sync(m); if (rc2) return rc2; }
    TSL_HIP(hipMemcpyAsync(m->h_ints, m->fset[m->last_set].F.stats, sizeof(tsl_frame_stats), hipMemcpyDeviceToHost, ms(m)));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    std::memcpy(&tmp, m->h_ints, sizeof(tmp));
    tmp.p_used = used;
    *out = tmp;
    return TSL_OK;
}

int tsl_tsdf_count_active(tsl_tsdf* m, int64_t* n)
{
    TSL_REQUIRE(m && n, "null"); TSL_HIP(hipSetDevice(m->device));
    long long* tmp = reinterpret_cast<long long*>(m->num_particles + 2);      // 8-byte scratch word
    TSL_HIP(hipMemsetAsync(tmp, 0, sizeof(long long), ms(m)));
    hipLaunchKernelGGL(k_count_active, dim3(m->nb3 < 4096 ? m->nb3 : 4096), dim3(256), 0, ms(m), m->M, map_slot(m, m->active), tmp);
    long long v = 0;
    TSL_HIP(hipMemcpyAsync(&v, tmp, sizeof(long long), hipMemcpyDeviceToHost, ms(m)));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    *n = v;
    return take_deferred(m);
}

static int export_common(tsl_tsdf* m, int mode, int16_t* idx, uint16_t* t, uint16_t* w, int8_t* occ, uint16_t* col, int64_t cap, int64_t* n)
{
    TSL_REQUIRE(m && n, "null"); TSL_REQUIRE(cap >= 0, "bad capacity"); TSL_HIP(hipSetDevice(m->device));
    const size_t c = (size_t)cap;
    const size_t o_idx = 0, o_t = o_idx + c * 6, o_w = o_t + c * 2, o_occ = o_w + c * 2, o_col = ((o_occ + c + 15) / 16) * 16, total = o_col + c * 6 + 64;
    int rc = grow(&m->xbuf, &m->xbuf_bytes, total); if (rc) return rc;
    char* base = (char*)m->xbuf;
    int* counter = m->num_particles + 2;
    TSL_HIP(hipMemsetAsync(counter, 0, sizeof(int), ms(m)));
    hipLaunchKernelGGL(k_export_sparse, dim3(m->nb3 < 4096 ? m->nb3 : 4096), dim3(256), 0, ms(m), m->M, map_slot(m, m->active), mode,
                       (int16_t*)(base + o_idx), (uint16_t*)(base + o_t), (uint16_t*)(base + o_w), (int8_t*)(base + o_occ),
                       (col && m->M.col) ? (uint16_t*)(base + o_col) : (uint16_t*)nullptr, (long long)cap, counter);
    int cnt = 0; rc = read_int(m, counter, &cnt); if (rc) return rc;
    *n = cnt;
    const size_t k = (size_t)(cnt < cap ? cnt : cap);
    if (k) {
        if (idx) TSL_HIP(hipMemcpy(idx, base + o_idx, k * 6, hipMemcpyDeviceToHost));
        if (t) TSL_HIP(hipMemcpy(t, base + o_t, k * 2, hipMemcpyDeviceToHost));
        if (w) TSL_HIP(hipMemcpy(w, base + o_w, k * 2, hipMemcpyDeviceToHost));
        if (occ) TSL_HIP(hipMemcpy(occ, base + o_occ, k, hipMemcpyDeviceToHost));
        if (col && m->M.col) TSL_HIP(hipMemcpy(col, base + o_col, k * 6, hipMemcpyDeviceToHost));
    }
    return TSL_OK;
}
int tsl_tsdf_export_sparse(tsl_tsdf* m, int16_t* idx, uint16_t* t, uint16_t* w, int8_t* occ, uint16_t* col, int64_t cap, int64_t* n)
{ return export_common(m, 0, idx, t, w, occ, col, cap, n); }
int tsl_tsdf_export_occupied(tsl_tsdf* m, int16_t* idx, int8_t* occ, int64_t cap, int64_t* n)
{ return export_common(m, 1, idx, nullptr, nullptr, occ, nullptr, cap, n); }

int tsl_tsdf_import_sparse(tsl_tsdf* m, int sid, const int16_t* idx, const uint16_t* t, const uint16_t* w, const int8_t* occ, const uint16_t* col, int64_t n)
{
    TSL_REQUIRE(m, "null handle"); TSL_REQUIRE(n >= 0, "bad count"); if (n == 0) return TSL_OK;
    TSL_REQUIRE(idx && t && w, "import_sparse: null arrays");
    TSL_REQUIRE(sid >= 0 && (m->cfg.is_global_map || sid < m->nsub), "import_sparse: submap id out of range");
    TSL_HIP(hipSetDevice(m->device));
    m->esdf_valid = false;
    const size_t c = (size_t)n;
    const size_t o_idx = 0, o_t = o_idx + c * 6, o_w = o_t + c * 2, o_occ = o_w + c * 2, o_col = ((o_occ + c + 15) / 16) * 16, total = o_col + c * 6 + 64;
    int rc = grow(&m->xbuf, &m->xbuf_bytes, total); if (rc) return rc;
    char* base = (char*)m->xbuf;
    TSL_HIP(hipMemcpy(base + o_idx, idx, c * 6, hipMemcpyHostToDevice));
    TSL_HIP(hipMemcpy(base + o_t, t, c * 2, hipMemcpyHostToDevice));
    TSL_HIP(hipMemcpy(base + o_w, w, c * 2, hipMemcpyHostToDevice));
    if (occ) TSL_HIP(hipMemcpy(base + o_occ, occ, c, hipMemcpyHostToDevice));
    const bool hc = col && m->M.col;
    if (hc) TSL_HIP(hipMemcpy(base + o_col, col, c * 6, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_import_sparse, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ms(m), m->M, map_slot(m, sid),
                       (const int16_t*)(base + o_idx), (const uint16_t*)(base + o_t), (const uint16_t*)(base + o_w),
                       occ ? (const int8_t*)(base + o_occ) : (const int8_t*)nullptr, hc ? (const uint16_t*)(base + o_col) : (const uint16_t*)nullptr, (long long)n);
    TSL_HIP(hipStreamSynchronize(ms(m)));
    return check_dev_err(m);
}

static int export_particles(tsl_tsdf* m, tsl_tsdf* dst, int mode, int keep, int slice_index, float dz, int32_t* n)
{
    TSL_REQUIRE(m, "null handle"); TSL_HIP(hipSetDevice(m->device));
    if (!dst) dst = m;
    TSL_REQUIRE(dst->device == m->device, "export: destination map lives on another device");
    if (!keep) TSL_HIP(hipMemsetAsync(dst->num_particles, 0, sizeof(int), ms(m)));          // :342-343 / :372-373
    const PoseF B = pose_of(m, m->active);
    hipLaunchKernelGGL(k_export_particles, dim3(m->nb3 < 4096 ? m->nb3 : 4096), dim3(256), 0, ms(m), m->M, map_slot(m, m->active), mode, B,
                       m->cfg.is_global_map, m->P.vs, m->surf_thres, m->disp_floor, m->disp_ceiling, slice_index, dz, m->colormap,
                       dst->exp_xyz, dst->exp_rgb, dst->exp_val, (long long)dst->max_disp, dst->num_particles);
    int cnt = 0; int rc = read_int(m, dst->num_particles, &cnt); if (rc) return rc;
    if (n) *n = cnt;
    return TSL_OK;
}
int tsl_tsdf_surface_voxels(tsl_tsdf* m, tsl_tsdf* dst, int add_to_cur, int32_t* n) { return export_particles(m, dst, 0, add_to_cur, 0, 0.0f, n); }
int tsl_tsdf_slice_voxels(tsl_tsdf* m, float z, float dz, int clear_last, int32_t* n)
{
    TSL_REQUIRE(m, "null handle");
    // slice_z is an f16 field (dense_tsdf.py:72,388); _index = int(z/voxel_scale) (:370)
    _Float16 zh = (_Float16)z; const float zq = (float)zh;
    return export_particles(m, nullptr, 1, !clear_last, (int)(zq / m->P.vs), dz, n);
}
int tsl_tsdf_read_exports(tsl_tsdf* m, float* xyz, float* rgb, float* val, int64_t n)
{
    TSL_REQUIRE(m, "null handle"); TSL_REQUIRE(n >= 0 && n <= m->max_disp, "read_exports: n out of range"); TSL_HIP(hipSetDevice(m->device));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    if (n == 0) return TSL_OK;
    if (xyz) TSL_HIP(hipMemcpy(xyz, m->exp_xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
    if (rgb) TSL_HIP(hipMemcpy(rgb, m->exp_rgb, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
    if (val) TSL_HIP(hipMemcpy(val, m->exp_val, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    return TSL_OK;
}
/* export_TSDF_xyz / export_color / export_TSDF as DEVICE pointers (f32 [max_disp_particles][3] / [3] / [1], valid for the lifetime of the
 * handle) and the particle count of the last cvt_* call: what taichislam_node.py:350-351 copies to numpy with .to_numpy(), for consumers
 * that stay on the GPU (torch tensors, a renderer).  Synchronises the handle's stream: the buffers are complete on return. */
int tsl_tsdf_exports_dev(tsl_tsdf* m, void** xyz_dev, void** rgb_dev, void** val_dev, int32_t* n)
{
    TSL_REQUIRE(m, "null handle"); TSL_HIP(hipSetDevice(m->device));
    if (xyz_dev) *xyz_dev = m->exp_xyz; if (rgb_dev) *rgb_dev = m->exp_rgb; if (val_dev) *val_dev = m->exp_val;
    int v = 0; const int rc = read_int(m, m->num_particles, &v); if (rc) return rc;      // (synchronises)
    if (n) *n = v;
    return TSL_OK;
}
int tsl_tsdf_set_export_row(tsl_tsdf* m, int field, int64_t row, const float v[3])
{
    TSL_REQUIRE(m && v && (field == 0 || field == 1), "set_export_row: bad argument"); TSL_REQUIRE(row >= 0 && row < m->max_disp, "set_export_row: row out of range");
    TSL_HIP(hipSetDevice(m->device));
    TSL_HIP(hipMemcpyAsync((field ? m->exp_rgb : m->exp_xyz) + 3 * row, v, sizeof(float) * 3, hipMemcpyHostToDevice, ms(m)));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    return TSL_OK;
}

int tsl_tsdf_pack_pointcloud2(tsl_tsdf* m, int has_rgb, int64_t n, void* out_host)
{
    TSL_REQUIRE(m && (n == 0 || out_host), "pack_pointcloud2: null argument"); TSL_REQUIRE(n >= 0 && n <= m->max_disp, "pack_pointcloud2: n out of range");
    TSL_HIP(hipSetDevice(m->device));
    if (n == 0) return TSL_OK;
    const int stride = has_rgb ? 6 : 3;
    int rc = grow(&m->xbuf, &m->xbuf_bytes, sizeof(float) * (size_t)stride * (size_t)n); if (rc) return rc;
    hipLaunchKernelGGL(k_pack_pointcloud2, dim3((unsigned)(((long long)n * stride + 255) / 256)), dim3(256), 0, ms(m), m->exp_xyz, m->exp_rgb, (float*)m->xbuf, (long long)n, stride);
    TSL_HIP(hipMemcpyAsync(out_host, m->xbuf, sizeof(float) * (size_t)stride * (size_t)n, hipMemcpyDeviceToHost, ms(m)));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    return TSL_OK;
}

int tsl_tsdf_num_particles(tsl_tsdf* m, int32_t* n) { TSL_REQUIRE(m && n, "null"); TSL_HIP(hipSetDevice(m->device)); int v = 0; int rc = read_int(m, m->num_particles, &v); *n = v; return rc; }
int tsl_tsdf_set_num_particles(tsl_tsdf* m, int32_t n)
{
    TSL_REQUIRE(m, "null"); TSL_HIP(hipSetDevice(m->device));
    m->h_ints[8] = n;
    TSL_HIP(hipMemcpyAsync(m->num_particles, &m->h_ints[8], sizeof(int), hipMemcpyHostToDevice, ms(m)));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    return TSL_OK;
}

int tsl_tsdf_prof_enable(tsl_tsdf* m, int on)
{
    TSL_REQUIRE(m, "null");
    m->prof_on = on != 0;
    m->prof_mask = (on == 0 || on == 1) ? ~0u : (unsigned)on >> 1;      // on = 1: every kernel; on = 2*mask: only the kernel ids in mask
    if (on) {   // a pool of timing events up front: creating them inside the measured region would cost what is being measured
        TSL_HIP(hipSetDevice(m->device));
        while (m->prof_free.size() < 256) { hipEvent_t e; TSL_HIP(hipEventCreate(&e)); m->prof_free.push_back(e); }
    }
    return TSL_OK;
}
int tsl_tsdf_prof_query(tsl_tsdf* m, int kid, double* total_ms, int64_t* launches)
{
    TSL_REQUIRE(m, "null"); TSL_REQUIRE(kid >= 0 && kid < TSL_K_COUNT, "bad kernel id"); TSL_HIP(hipSetDevice(m->device));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    for (auto& s : m->prof) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { m->prof_ms[s.kid] += ms; m->prof_n[s.kid] += s.count; }
        m->prof_free.push_back(s.a); m->prof_free.push_back(s.b);
    }
    m->prof.clear();
    (void)hipGetLastError();                        // a pair that was never recorded must not leave its error behind for the next caller
    if (total_ms) *total_ms = m->prof_ms[kid];
    if (launches) *launches = m->prof_n[kid];
    m->prof_ms[kid] = 0.0; m->prof_n[kid] = 0;
    return TSL_OK;
}

/* developer aid (not part of the public header): read and optionally clear the TSL_TIMING cycle counters */
int tsl_tsdf_debug_counters(tsl_tsdf* m, int64_t* out, int reset)
{
    TSL_REQUIRE(m && out, "null"); TSL_HIP(hipSetDevice(m->device));
    int rc = tsl_tsdf_sync(m); if (rc) return rc;
    TSL_REQUIRE(m->scratch_ready, "debug_counters: nothing integrated yet");
    TSL_HIP(hipMemcpy(out, m->F.dbg, sizeof(long long) * 16384 * 16, hipMemcpyDeviceToHost));
    if (reset) TSL_HIP(hipMemset(m->F.dbg, 0, sizeof(long long) * 16384 * 16));
    return TSL_OK;
}

/* backend knobs used by bench.py / tests to A/B kernel variants: name in {"variant","split"} */
int tsl_tsdf_set_option(tsl_tsdf* m, const char* name, int value)
{
    TSL_REQUIRE(m && name, "null");
    if (!std::strcmp(name, "variant")) {
        TSL_REQUIRE(value >= 0 && value <= 2, "variant must be 0, 1 or 2");
        TSL_REQUIRE(value == 2 || !m->semantics, "variant: the sequential semantics run on variant 2 only (set semantics 0 first)");
        if (value != 2 && m->variant == 2 && m->scratch_ready) {
            // the global-atomics variants expect their brick scratch to be zero between launches; the brick-binned path leaves the
            // parts' sums of its last batches in the same buffers
            int rc = tsl_tsdf_sync(m); if (rc) return rc;
            for (int bi = 0; bi < TSL_NBATCH; ++bi) TSL_HIP(hipMemsetAsync(m->fset[bi * TSL_NB].F.acc, 0, 16 * (size_t)m->F.max_frame_bricks * TSL_BRK3, m->stream_));
            TSL_HIP(hipStreamSynchronize(m->stream_));
        }
        m->variant = value; return TSL_OK;
    }
    if (!std::strcmp(name, "group")) { TSL_REQUIRE(value != 0 || !m->semantics, "group: the sequential semantics need the hash grouping (set semantics 0 first)"); int rc = tsl_tsdf_sync(m); if (rc) return rc; m->P.group = value != 0; return TSL_OK; }
    if (!std::strcmp(name, "semantics")) {
        TSL_REQUIRE(value == 0 || value == 1, "semantics must be 0 (batched exact sums) or 1 (sequential replay of the reference)");
        TSL_REQUIRE(value == 0 || m->M.max_bricks <= (1 << 17), "sequential semantics: maps with at most 2^17 bricks");
        // the sequential replay lives on the hash-grouped brick path only: with another variant / grouping the frames would be
        // integrated with the batched sums while get_option reported semantics = 1
        TSL_REQUIRE(value == 0 || (m->variant == 2 && m->P.group), "sequential semantics needs variant 2 and the hash grouping (group 1)");
        TSL_REQUIRE(value == 0 || m->F.max_points <= (1 << 21), "sequential semantics: at most 2^21 points per frame");
        int rc = tsl_tsdf_sync(m); if (rc) return rc;
        if (value && m->seq_impl && !m->cfg.is_global_map) { TSL_HIP(hipSetDevice(m->device)); rc = seq_prepare(m); if (rc) return rc; }      // ~4.1 GB of replay scratch at the default sizes (header)
        m->semantics = value; m->P.seq = value; return TSL_OK;
    }
    if (!std::strcmp(name, "seq_impl")) {
        TSL_REQUIRE(value == 0 || value == 1, "seq_impl must be 1 (per-brick replay runs) or 0 (round 3's global sorts)");
        int rc = tsl_tsdf_sync(m); if (rc) return rc;
        m->seq_impl = value; return TSL_OK;
    }
    if (!std::strcmp(name, "seq_tuple_cap")) {
        TSL_REQUIRE(value >= (1 << 16), "seq_tuple_cap: at least 2^16 ray steps per frame");
        int rc = tsl_tsdf_sync(m); if (rc) return rc;
        const long long old_cap = m->seq_tuple_cap;
        m->seq_tuple_cap = value;
        if (m->seq_ready) {          // the scratch exists already (allocated when the mode was switched on): it is rebuilt at the new size
            TSL_HIP(hipSetDevice(m->device));
            seq_release(m);
            if (m->semantics && m->seq_impl && (rc = seq_prepare(m))) {
                // the new size does not fit: back to the old one, reported HERE; if even that fails the handle leaves the literal mode rather than
                // meeting the failure in the middle of a batch
                m->seq_tuple_cap = old_cap;
                if (seq_prepare(m)) m->semantics = 0;
                return rc;
            }
        }
        return TSL_OK;
    }
    if (!std::strcmp(name, "fastdiv")) { if (value == 0) m->P.fastdiv = 0; return TSL_OK; }
    if (!std::strcmp(name, "mesh_gather")) { m->mesh_gather = value != 0; return TSL_OK; }
    if (!std::strcmp(name, "esdf_full")) { m->esdf_force_full = value != 0; return TSL_OK; }
    if (!std::strcmp(name, "esdf_mode")) { TSL_REQUIRE(value == 0 || value == 1, "esdf_mode: 0 = regional recompute (default), 1 = raise / lower wavefront with parent directions"); int rc = esdf_finish(m); if (rc) return rc; m->esdf_mode = value; m->esdf_valid = false; return TSL_OK; }
    if (!std::strcmp(name, "esdf_round_cap")) { m->esdf_round_cap = value; return TSL_OK; }
    if (!std::strcmp(name, "merge_exchange")) { TSL_REQUIRE(value == 0 || value == 1, "merge_exchange: 0 = all-reduce of the packed sums, 1 = reduce-scatter + all-gather of finalised voxels"); m->merge_exchange = value; return TSL_OK; }
    if (!std::strcmp(name, "fuse_direct")) { m->fuse_direct = value != 0; return TSL_OK; }      // 1 = round 5's splat (global atomics per corner) on the global map, for A/B
    if (!std::strcmp(name, "esdf_grid")) { TSL_REQUIRE(value >= 0, "esdf_grid: workgroups of a relaxation round (0 = four per CU)"); m->esdf_grid = value; return TSL_OK; }
    if (!std::strcmp(name, "esdf_overlap")) { int rc = esdf_finish(m); if (rc) return rc; m->esdf_overlap = value != 0; return TSL_OK; }
    if (!std::strcmp(name, "unit")) { TSL_REQUIRE(value >= 0 && value <= (1 << 20), "unit must be 0..2^20 segments"); int rc = tsl_tsdf_sync(m); if (rc) return rc; m->unit_max = value; return TSL_OK; }
    if (!std::strcmp(name, "unit_floor")) { TSL_REQUIRE(value >= 0 && value <= (1 << 20), "unit_floor must be 0..2^20 segments"); int rc = tsl_tsdf_sync(m); if (rc) return rc; m->unit_floor = value; return TSL_OK; }
    if (!std::strcmp(name, "unit_half")) { TSL_REQUIRE(value >= 0 && value <= (1 << 20), "unit_half must be 0..2^20 segments"); int rc = tsl_tsdf_sync(m); if (rc) return rc; m->unit_half = value; return TSL_OK; }
    if (!std::strcmp(name, "chunks")) { TSL_REQUIRE(value >= 1 && value <= 8, "chunks must be 1..8"); int rc = tsl_tsdf_sync(m); if (rc) return rc; m->chunks = value; return TSL_OK; }
    if (!std::strcmp(name, "adaptive")) { m->adaptive = value != 0; return TSL_OK; }
    if (!std::strcmp(name, "ramp_size")) { TSL_REQUIRE(value >= 1 && value <= TSL_NB, "ramp_size must be 1..8 frames"); m->ramp_size = value; return TSL_OK; }
    if (!std::strcmp(name, "ramp")) { TSL_REQUIRE(value >= 0 && value <= 16, "ramp must be 0..16 half batches"); m->ramp_batches = value; return TSL_OK; }
    if (!std::strcmp(name, "ugrid") || !std::strcmp(name, "pgrid")) { TSL_REQUIRE(value >= 5 && value <= 200, "ugrid / pgrid must be 5..200 (percent of the resident workgroup slots)"); { int rc = tsl_tsdf_sync(m); if (rc) return rc; } (name[0] == 'u' ? m->ugrid : m->pgrid) = value; return TSL_OK; }
    if (!std::strcmp(name, "split_launch")) { TSL_REQUIRE(value == 0 || value == 1, "split_launch must be 0 or 1"); { int rc = tsl_tsdf_sync(m); if (rc) return rc; } m->split_launch = value; return TSL_OK; }
    if (!std::strcmp(name, "bgrid")) { TSL_REQUIRE(value >= 10 && value <= 200, "bgrid must be 10..200 (percent of the resident workgroup slots)"); { int rc = tsl_tsdf_sync(m); if (rc) return rc; } m->bgrid = value; return TSL_OK; }
    if (!std::strcmp(name, "spt")) { TSL_REQUIRE(value == 2 || value == 4, "spt must be 2 or 4"); { int rc = tsl_tsdf_sync(m); if (rc) return rc; } m->spt = value; return TSL_OK; }
    if (!std::strcmp(name, "wg")) { TSL_REQUIRE(value == 256 || value == 512, "wg must be 256 or 512"); { int rc = tsl_tsdf_sync(m); if (rc) return rc; } m->wg = value; return TSL_OK; }
    if (!std::strcmp(name, "phases")) { int rc = tsl_tsdf_sync(m); if (rc) return rc; m->phases = value & 3; return TSL_OK; }      // developer timing aid: 1 = phase A only, 2 = phase B only (map contents are then meaningless)
    if (!std::strcmp(name, "overlap")) { int rc = tsl_tsdf_sync(m); if (rc) return rc; m->overlap = value < 0 ? 0 : (value > TSL_NB ? TSL_NB : value); for (auto& H : m->batch) H.b_pending = false; return TSL_OK; }
    if (!std::strcmp(name, "split")) { TSL_REQUIRE(value >= 1 && value <= 64 && (64 % value) == 0, "split must divide 64"); m->split = value; return TSL_OK; }
    set_error("unknown option"); return TSL_ERR_ARG;
}

}  // extern "C"
