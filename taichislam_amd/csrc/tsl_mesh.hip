// tsl_mesh.hip -- marching cubes over every active TSDF voxel.  Replaces MarchingCubeMesher.generate_mesh_kernel and
// helpers (taichi_slam/mapping/marching_cube_mesher.py:44-187, reference root).
//
// One thread per voxel of every allocated brick.  The 8 corner values of the cell are kept per thread in LDS (dynamic
// edge -> corner indexing would otherwise spill to scratch), the case table is the packed Bourke table (one u64 per
// cube case) staged in LDS, and triangles are emitted with a wave-level inclusive scan of the per-voxel triangle counts
// and ONE atomic per wave on the facet counter (the reference does one atomic per triangle, :114).
#include "tsl_tsdf.hpp"
#include "mc_tables_data.h"

namespace tsl {

#define MC_EPS 1e-6f                                                         // marching_cube_mesher.py:6

__device__ __forceinline__ float rd_tsdf(const MapDev& M, int s, int i, int j, int k, int* obs)
{
    if (!in_volume(M, i, j, k)) { *obs = 0; return 0.0f; }                   // reading outside / an inactive cell yields 0 (A7)
    int l; const int b = brick_of(M, i, j, k, &l);
    const int p = pool_lookup_ro(M, s, b);
    if (p < 0) { *obs = 0; return 0.0f; }
    const size_t v = (size_t)p * TSL_BRK3 + l;
    *obs = M.obs[v];
    return h2f((h16)(M.tw[v] & 0xffffu));
}
__device__ __forceinline__ h16 rd_tsdf_h(const MapDev& M, int s, int i, int j, int k)
{
    if (!in_volume(M, i, j, k)) return 0;
    int l; const int b = brick_of(M, i, j, k, &l);
    const int p = pool_lookup_ro(M, s, b);
    return p < 0 ? (h16)0 : (h16)(M.tw[(size_t)p * TSL_BRK3 + l] & 0xffffu);
}
// generate_normal :84-93 -- f16 central differences, f16 normalisation (invlen = 1/norm; invlen * v)
__device__ __forceinline__ void gen_normal(const MapDev& M, int s, const float* p, float* out)
{
    const int q0 = (int)rnd_f(p[0]), q1 = (int)rnd_f(p[1]), q2 = (int)rnd_f(p[2]);
    const h16 n0 = hsub(rd_tsdf_h(M, s, q0 + 1, q1, q2), rd_tsdf_h(M, s, q0 - 1, q1, q2));
    const h16 n1 = hsub(rd_tsdf_h(M, s, q0, q1 + 1, q2), rd_tsdf_h(M, s, q0, q1 - 1, q2));
    const h16 n2 = hsub(rd_tsdf_h(M, s, q0, q1, q2 + 1), rd_tsdf_h(M, s, q0, q1, q2 - 1));
    const h16 nrm = hsqrt(hadd(hadd(hmul(n0, n0), hmul(n1, n1)), hmul(n2, n2)));
    const h16 inv = f2h(1.0f / h2f(nrm));
    out[0] = h2f(hmul(inv, n0)); out[1] = h2f(hmul(inv, n1)); out[2] = h2f(hmul(inv, n2));
}
// cube corner q -> offset (grid_xyz :196-206) and edge e -> its two corners (edges_grid_id :208-221)
__device__ __forceinline__ void corner_off(int q, int* d) { d[0] = ((q + 1) >> 1) & 1; d[1] = (q >> 1) & 1; d[2] = (q >> 2) & 1; }
__device__ __forceinline__ void edge_corners(int e, int* a, int* b)
{
    if (e < 4) { *a = e; *b = (e + 1) & 3; }
    else if (e < 8) { *a = e; *b = 4 + ((e - 3) & 3); }
    else { *a = e - 8; *b = e - 4; }
}

__device__ __forceinline__ uint2 rd_col(const MapDev& M, int s, int i, int j, int k)
{
    if (!in_volume(M, i, j, k)) return make_uint2(0u, 0u);
    int l; const int b = brick_of(M, i, j, k, &l);
    const int p = pool_lookup_ro(M, s, b);
    return p < 0 ? make_uint2(0u, 0u) : reinterpret_cast<const uint2*>(M.col)[(size_t)p * TSL_BRK3 + l];
}

__global__ void __launch_bounds__(256) k_marching_cubes(MapDev M, int nused, int step, float thres, float vs, long long max_tri,
                                                        float* __restrict__ verts, float* __restrict__ normals, float* __restrict__ colors, int* counter)
{
    __shared__ unsigned long long s_tri[256];
    __shared__ float s_val[8][256];
    s_tri[threadIdx.x] = MC_TRI_PACKED[threadIdx.x];
    __syncthreads();
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        const int owner = M.owner[p];
        const int s = owner / M.nb3, b = owner - s * M.nb3;
        const int bk = b % M.nbz, bj = (b / M.nbz) % M.nbx, bi = b / (M.nbz * M.nbx);
        for (int l0 = 0; l0 < TSL_BRK3; l0 += 256) {
            const int l = l0 + threadIdx.x;
            const size_t v = (size_t)p * TSL_BRK3 + l;
            const int i = bi * 16 + (l >> 8) - M.hN, j = bj * 16 + ((l >> 4) & 15) - M.hN, k = bk * 16 + (l & 15) - M.hNz;
            int ntri = 0, cube = 0;
            unsigned long long tri = ~0ull;
            if (M.obs[v] > 0 && h2f((h16)(M.tw[v] & 0xffffu)) < thres) {                         // :184
                bool bad = false;
                for (int q = 0; q < 8; ++q) {                                                     // :133-138
                    int d[3]; corner_off(q, d);
                    int o; const float val = rd_tsdf(M, s, i + d[0] * step, j + d[1] * step, k + d[2] * step, &o);
                    s_val[q][threadIdx.x] = val;
                    if (o == 0) bad = true;
                    if (val < 0.0f) cube |= 1 << q;                                               // :141-144
                }
                if (!bad) {
                    tri = s_tri[cube];
                    for (int t = 0; t < 5; ++t) if (((tri >> (12 * t)) & 0xfull) != 0xfull) ++ntri;   // :173-177 (triTable[cube][3t] != -1)
                }
            }
            // wave-level emission: inclusive scan of ntri, one atomic per wave
            int inc = ntri;
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane_id() >= d) inc += o; }
            const int wave_total = __shfl(inc, 63);
            int base = 0;
            if (wave_total) {
                if (lane_id() == 63) base = __hip_atomic_fetch_add(counter, wave_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                base = __shfl(base, 63);
            }
            long long idx = (long long)base + inc - ntri;
            for (int t = 0; t < 5 && ntri; ++t) {
                if (((tri >> (12 * t)) & 0xfull) == 0xfull) continue;
                if (idx < max_tri) {                                                              // Q10: clamp by the returned index
                    for (int q = 0; q < 3; ++q) {
                        const int e = (int)((tri >> (4 * (3 * t + q))) & 0xfull);
                        int ca, cb; edge_corners(e, &ca, &cb);
                        int da[3], db[3]; corner_off(ca, da); corner_off(cb, db);
                        const float v0 = s_val[ca][threadIdx.x], v1 = s_val[cb][threadIdx.x];
                        const float p0[3] = { (float)(i + da[0] * step), (float)(j + da[1] * step), (float)(k + da[2] * step) };
                        const float p1[3] = { (float)(i + db[0] * step), (float)(j + db[1] * step), (float)(k + db[2] * step) };
                        float pv[3], mu = 0.0f;
                        if (fabsf(0.0f - v0) < MC_EPS) { pv[0] = p0[0]; pv[1] = p0[1]; pv[2] = p0[2]; }            // vertexInterp :44-60
                        else if (fabsf(0.0f - v1) < MC_EPS) { pv[0] = p1[0]; pv[1] = p1[1]; pv[2] = p1[2]; }
                        else { mu = (0.0f - v0) / h2f(hsub(f2h(v1), f2h(v0))); for (int a = 0; a < 3; ++a) pv[a] = p0[a] + mu * (p1[a] - p0[a]); }   // (f16 - f16: an f16 operation)
                        if (colors) {                                                            // vertexInterp_color :62-82 (Q13: only channel 0 is tested)
                            const uint2 ca2 = rd_col(M, s, i + da[0] * step, j + da[1] * step, k + da[2] * step);
                            const uint2 cb2 = rd_col(M, s, i + db[0] * step, j + db[1] * step, k + db[2] * step);
                            const h16 c0[3] = { (h16)(ca2.x & 0xffffu), (h16)(ca2.x >> 16), (h16)(ca2.y & 0xffffu) };
                            const h16 c1[3] = { (h16)(cb2.x & 0xffffu), (h16)(cb2.x >> 16), (h16)(cb2.y & 0xffffu) };
                            float vc[3] = { h2f(c0[0]), h2f(c0[1]), h2f(c0[2]) };
                            if (h2f(c0[0]) == 0.0f) { for (int a = 0; a < 3; ++a) vc[a] = h2f(c1[a]); }
                            else if (!(h2f(c1[0]) == 0.0f)) { for (int a = 0; a < 3; ++a) vc[a] = h2f(f2h(h2f(c0[a]) + mu * h2f(hsub(c1[a], c0[a])))); }   // (p_color is an f16 variable: first assigned c1)
                            const size_t oc = ((size_t)idx * 3 + q) * 3;
                            for (int a = 0; a < 3; ++a) colors[oc + a] = vc[a];
                        }
                        float nn[3]; gen_normal(M, s, pv, nn);                                   // :100-102
                        const size_t o = ((size_t)idx * 3 + q) * 3;
                        for (int a = 0; a < 3; ++a) { verts[o + a] = pv[a] * vs; normals[o + a] = nn[a]; }   // :41-42,:97-99
                    }
                }
                ++idx;
            }
        }
    }
}


// ---- step 1 (every caller of the reference: scripts/taichislam_node.py:338, tests/marching_cube_test.py:21): brick + halo staged in LDS ----
// A cell needs its 8 corners (voxel .. voxel + 1) and, per emitted vertex, the central differences around the voxel nearest to the
// vertex (corner .. corner + 1, each +- 1): everything a brick's cells read lies in [-1, 17]^3 around the brick.  That 19^3 tile
// (f16 TSDF bits + observed bits, 14.3 KiB) is gathered ONCE per brick through the 27 surrounding brick-table entries; the corner
// reads (8 per voxel) and the normal reads (6 per vertex, up to 90 per voxel) then come from LDS instead of one table lookup +
// one scattered 4-byte load each.  Emission: the brick's triangles are counted first (pass 1), reserved with ONE atomic, then written (pass 2).
#define MC_T 19
#define MC_T3 (MC_T * MC_T * MC_T)
__device__ __forceinline__ int mc_tile(int x, int y, int z) { return (x * MC_T + y) * MC_T + z; }      // tile coords = brick coords + 1

#define MC_SLAB 2         // x-layers of a brick per work item of k_marching_cubes_lds
#define MC_LIST 4096             // triangles listed in LDS at a time
// the cell at brick-local index l: its case (through *cube when asked for) and how many triangles the case has; 0 when the cell's own
// voxel is unobserved or not below the threshold (:184), or a corner is unobserved (:133-138)
__device__ __forceinline__ int mc_cell(const uint16_t* s_t, const uint32_t* s_o, const unsigned long long* s_tri, int l, float thres, int* cube)
{
    const int lx = l >> 8, ly = (l >> 4) & 15, lz = l & 15;
    const int t0 = mc_tile(lx + 1, ly + 1, lz + 1);
    if (!(((s_o[t0 >> 5] >> (t0 & 31)) & 1u) && h2f(s_t[t0]) < thres)) return 0;
    int c = 0;
    for (int q = 0; q < 8; ++q) {
        int d[3]; corner_off(q, d);
        const int t = mc_tile(lx + 1 + d[0], ly + 1 + d[1], lz + 1 + d[2]);
        if (!((s_o[t >> 5] >> (t & 31)) & 1u)) return 0;
        if (h2f(s_t[t]) < 0.0f) c |= 1 << q;                                                      // :141-144
    }
    const unsigned long long tri = s_tri[c];
    int n = 0;
    for (int t = 0; t < 5; ++t) if (((tri >> (12 * t)) & 0xfull) != 0xfull) ++n;                  // :173-177 (triTable[cube][3t] != -1)
    if (cube) *cube = c;
    return n;
}

// Which bricks can hold a triangle at all: a cell emits one only if its eight corners are not all on one side of zero (the case table is empty
// for 0 and 255, :141-177).  One coalesced pass over the stored values notes per brick whether it holds a value < 0 and whether it holds one that
// is not (a voxel nobody wrote reads 0, as in the reference); k_marching_cubes_lds then skips a brick -- before it gathers the 19^3 tile from 27
// bricks -- when it and the seven bricks its cells' corners reach into are all on one side.  In a map of a closed surface most bricks are.
__global__ void __launch_bounds__(256) k_mc_summary(MapDev M, int nused, uint32_t* __restrict__ flags, int* counter)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *counter = 0;                  // :182 (the triangle counter of the mesh that follows)
    if (nused < 0) nused = min(M.pool_top[0], M.max_bricks);                // (the bricks in use, read here: the host does not wait for the count)
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        const uint4* tw = reinterpret_cast<const uint4*>(M.tw + (size_t)p * TSL_BRK3);
        uint32_t f = 0u;
#pragma unroll
        for (int q = 0; q < TSL_BRK3 / 4 / 256; ++q) {
            const uint4 v = tw[q * 256 + threadIdx.x];
            const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int u = 0; u < 4; ++u) f |= h2f((h16)(w4[u] & 0xffffu)) < 0.0f ? 1u : 2u;
        }
        const int neg = __syncthreads_or((int)(f & 1u)), nn = __syncthreads_or((int)(f & 2u));
        if (threadIdx.x == 0) flags[p] = (neg ? 1u : 0u) | (nn ? 2u : 0u);
    }
}

__global__ void __launch_bounds__(256) k_marching_cubes_lds(MapDev M, int nused, float thres, float vs, long long max_tri,
                                                            float* __restrict__ verts, float* __restrict__ normals, float* __restrict__ colors, int* counter,
                                                            const uint32_t* __restrict__ flags)
{
    __shared__ int s_skip;
    if (nused < 0) nused = min(M.pool_top[0], M.max_bricks);
    __shared__ unsigned long long s_tri[256];
    __shared__ uint32_t s_list[MC_LIST];
    __shared__ uint16_t s_t[MC_T3];                    // TSDF f16 bits (0 where nothing is stored: reading an inactive cell yields 0, A7)
    __shared__ uint32_t s_o[(MC_T3 + 31) / 32];        // TSDF_observed > 0
    __shared__ int s_nb[27];
    __shared__ int s_wtot[5], s_base;
    s_tri[threadIdx.x] = MC_TRI_PACKED[threadIdx.x];
    // a work item is a SLAB of a brick, MC_SLAB of its 16 x-layers: the surface crosses few bricks -- 56 of the 442 of a 128^3 sphere -- and a brick
    // per workgroup left three quarters of the CUs idle while those few walked their cells and emitted their vertices (one wave per SIMD: every
    // dependent instruction at its full latency)
    for (int item = blockIdx.x; item < nused * (16 / MC_SLAB); item += gridDim.x) {
        const int p = item / (16 / MC_SLAB), x0 = (item % (16 / MC_SLAB)) * MC_SLAB;
        const int owner = M.owner[p];
        const int s = owner / M.nb3, b = owner - s * M.nb3;
        const int bk = b % M.nbz, bj = (b / M.nbz) % M.nbx, bi = b / (M.nbz * M.nbx);
        __syncthreads();                                                     // the previous item's tile is no longer read
        if (threadIdx.x < 64) {                                               // (the first wave)
            int np = -1; bool corner = false;
            if (threadIdx.x < 27) {
                const int di = (int)threadIdx.x / 9 - 1, dj = ((int)threadIdx.x / 3) % 3 - 1, dk = (int)threadIdx.x % 3 - 1;
                const int i = bi + di, j = bj + dj, k = bk + dk;
                np = (i < 0 || i >= M.nbx || j < 0 || j >= M.nbx || k < 0 || k >= M.nbz) ? -1 : pool_lookup_ro(M, s, (i * M.nbx + j) * M.nbz + k);
                s_nb[threadIdx.x] = np;
                corner = di >= 0 && dj >= 0 && dk >= 0;                         // the eight bricks the corners of this brick's cells lie in
            }
            const uint32_t f = corner ? (np >= 0 ? flags[np] : 2u) : 0u;        // (no brick: the corners read 0)
            const bool neg = __ballot((f & 1u) != 0u) != 0ull, nn = __ballot((f & 2u) != 0u) != 0ull;
            if (threadIdx.x == 0) s_skip = (neg && nn) ? 0 : 1;
        }
        for (int i = threadIdx.x; i < (MC_T3 + 31) / 32; i += 256) s_o[i] = 0u;
        __syncthreads();
        if (s_skip) continue;                                                // (uniform) every corner of every cell on one side of zero: no triangle
        // the tile entries of the slab's cells, their corners and the normals' neighbours -- x-layers x0 .. x0 + MC_SLAB + 2 of the 19^3 tile --
        // gathered MC_GB at a time as ONE batch of independent loads (loaded from a valid address unconditionally so that nothing separates
        // the requests: a loop of dependent gathers costs a memory latency per entry)
        constexpr int NT = (MC_SLAB + 3) * MC_T * MC_T, PER = (NT + 255) / 256, MC_GB = (PER + 1) / 2;
        for (int c = 0; c < PER; c += MC_GB) {
            uint32_t tw[MC_GB]; int8_t ob[MC_GB]; bool ok[MC_GB]; int tt[MC_GB];
#pragma unroll
            for (int u = 0; u < MC_GB; ++u) {
                const int tl = (int)threadIdx.x + (c + u) * 256;
                const int t = tl < NT && c + u < PER ? x0 * MC_T * MC_T + tl : MC_T3;
                tt[u] = t;
                const int tc = t < MC_T3 ? t : 0;
                const int tz = tc % MC_T, ty = (tc / MC_T) % MC_T, tx = tc / (MC_T * MC_T);
                const int np = s_nb[(((tx + 15) >> 4) * 3 + ((ty + 15) >> 4)) * 3 + ((tz + 15) >> 4)];
                ok[u] = t < MC_T3 && np >= 0 && in_volume(M, bi * 16 + tx - 1 - M.hN, bj * 16 + ty - 1 - M.hN, bk * 16 + tz - 1 - M.hNz);
                const size_t v = (size_t)(ok[u] ? np : p) * TSL_BRK3 + ((((tx + 15) & 15) << 8) | (((ty + 15) & 15) << 4) | ((tz + 15) & 15));
                tw[u] = M.tw[v]; ob[u] = M.obs[v];
            }
#pragma unroll
            for (int u = 0; u < MC_GB; ++u) {
                const int t = tt[u];
                if (t >= MC_T3) continue;
                s_t[t] = ok[u] ? (uint16_t)(tw[u] & 0xffffu) : (uint16_t)0;
                if (ok[u] && ob[u] > 0) atomicOr(&s_o[t >> 5], 1u << (t & 31));
            }
        }
        __syncthreads();
        // ---- pass 1: the brick's triangle count.  ONE reservation per brick (a reservation per wave and 256 cells -- the first form of
        //      this kernel -- put ~2 000 returning atomics on one address per mesh, ~12 ns each in the L2) ----
        int mine = 0;
        for (int l0 = x0 << 8; l0 < (x0 + MC_SLAB) << 8; l0 += 256) mine += mc_cell(s_t, s_o, s_tri, l0 + threadIdx.x, thres, nullptr);
        int incl = mine;
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane_id() >= d) incl += o; }
        if (lane_id() == 63) s_wtot[threadIdx.x >> 6] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
            s_wtot[4] = tot;
            s_base = tot ? __hip_atomic_fetch_add(counter, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
        __syncthreads();
        const int total = s_wtot[4];
        if (total == 0) continue;                                         // (uniform) nothing to emit from this brick
        int first = incl - mine;                                          // this thread's first triangle among the brick's
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) first += s_wtot[w];
        // ---- pass 2: the triangles as a LIST in LDS (cell | triangle of the case << 12 | case << 16, in the order of the reservation), then
        //      one VERTEX per thread and step.  Walking the cells again and letting each thread emit its own cells' triangles (the first
        //      form) keeps 5 - 10 % of the lanes busy -- the surface crosses few of a wave's 64 cells -- through 15 serial vertex slots of
        //      ~600 instructions (the f16 normal arithmetic is emulated bit for bit): 148 us per mesh of a 128^3 sphere, this way 1/4. ----
        for (int w0 = 0; w0 < total; w0 += MC_LIST) {                     // (one window unless the brick is noise: MC_LIST triangles at a time)
            if (w0) __syncthreads();
            int at = first;
            for (int l0 = x0 << 8; l0 < ((x0 + MC_SLAB) << 8) && mine; l0 += 256) {
                const int l = l0 + threadIdx.x;
                int cube = 0;
                const int n = mc_cell(s_t, s_o, s_tri, l, thres, &cube);
                for (int t = 0; t < n; ++t, ++at)
                    if (at >= w0 && at < w0 + MC_LIST) s_list[at - w0] = (uint32_t)l | (uint32_t)t << 12 | (uint32_t)cube << 16;
            }
            __syncthreads();
            const int nv = 3 * min(total - w0, MC_LIST);
            for (int v = threadIdx.x; v < nv; v += 256) {
                const uint32_t en = s_list[v / 3];
                const int q = v % 3, l = (int)(en & 0xfffu), t = (int)((en >> 12) & 7u);
                const long long idx = (long long)s_base + w0 + v / 3;
                if (idx >= max_tri) continue;                                                     // Q10: clamp by the returned index
                const unsigned long long tri = s_tri[en >> 16];
                // the triangles of a case are its non-empty slots in order: slot of the t-th one
                int slot = 0;
                for (int u = 0, seen = 0; u < 5; ++u) if (((tri >> (12 * u)) & 0xfull) != 0xfull) { if (seen == t) slot = u; ++seen; }
                const int lx = l >> 8, ly = (l >> 4) & 15, lz = l & 15;
                const int i = bi * 16 + lx - M.hN, j = bj * 16 + ly - M.hN, k = bk * 16 + lz - M.hNz;
                const int e = (int)((tri >> (4 * (3 * slot + q))) & 0xfull);
                int ca, cb; edge_corners(e, &ca, &cb);
                int da[3], db[3]; corner_off(ca, da); corner_off(cb, db);
                const float v0 = h2f(s_t[mc_tile(lx + 1 + da[0], ly + 1 + da[1], lz + 1 + da[2])]), v1 = h2f(s_t[mc_tile(lx + 1 + db[0], ly + 1 + db[1], lz + 1 + db[2])]);
                const float p0[3] = { (float)(i + da[0]), (float)(j + da[1]), (float)(k + da[2]) };
                const float p1[3] = { (float)(i + db[0]), (float)(j + db[1]), (float)(k + db[2]) };
                float pv[3], mu = 0.0f;
                if (fabsf(0.0f - v0) < MC_EPS) { pv[0] = p0[0]; pv[1] = p0[1]; pv[2] = p0[2]; }            // vertexInterp :44-60
                else if (fabsf(0.0f - v1) < MC_EPS) { pv[0] = p1[0]; pv[1] = p1[1]; pv[2] = p1[2]; }
                else { mu = (0.0f - v0) / h2f(hsub(f2h(v1), f2h(v0))); for (int a = 0; a < 3; ++a) pv[a] = p0[a] + mu * (p1[a] - p0[a]); }      // (valp2 - valp1 of two f16 values is an f16 operation)
                if (colors) {                                                            // vertexInterp_color :62-82 (Q13); colours stay in HBM
                    const uint2 ca2 = rd_col(M, s, i + da[0], j + da[1], k + da[2]);
                    const uint2 cb2 = rd_col(M, s, i + db[0], j + db[1], k + db[2]);
                    const h16 c0[3] = { (h16)(ca2.x & 0xffffu), (h16)(ca2.x >> 16), (h16)(ca2.y & 0xffffu) };
                    const h16 c1[3] = { (h16)(cb2.x & 0xffffu), (h16)(cb2.x >> 16), (h16)(cb2.y & 0xffffu) };
                    float vc[3] = { h2f(c0[0]), h2f(c0[1]), h2f(c0[2]) };
                    if (h2f(c0[0]) == 0.0f) { for (int a = 0; a < 3; ++a) vc[a] = h2f(c1[a]); }
                    else if (!(h2f(c1[0]) == 0.0f)) { for (int a = 0; a < 3; ++a) vc[a] = h2f(f2h(h2f(c0[a]) + mu * h2f(hsub(c1[a], c0[a])))); }   // (p_color is an f16 variable: first assigned c1)
                    const size_t oc = ((size_t)idx * 3 + q) * 3;
                    for (int a = 0; a < 3; ++a) colors[oc + a] = vc[a];
                }
                // generate_normal :84-93 around the voxel nearest to the vertex: it is one of the edge's two corners
                const int q0 = (int)rnd_f(pv[0]) - i + lx + 1, q1 = (int)rnd_f(pv[1]) - j + ly + 1, q2 = (int)rnd_f(pv[2]) - k + lz + 1;
                const h16 n0 = hsub(s_t[mc_tile(q0 + 1, q1, q2)], s_t[mc_tile(q0 - 1, q1, q2)]);
                const h16 n1 = hsub(s_t[mc_tile(q0, q1 + 1, q2)], s_t[mc_tile(q0, q1 - 1, q2)]);
                const h16 n2 = hsub(s_t[mc_tile(q0, q1, q2 + 1)], s_t[mc_tile(q0, q1, q2 - 1)]);
                const h16 nrm = hsqrt(hadd(hadd(hmul(n0, n0), hmul(n1, n1)), hmul(n2, n2)));
                const h16 inv = f2h(1.0f / h2f(nrm));
                const size_t o = ((size_t)idx * 3 + q) * 3;
                verts[o] = pv[0] * vs; verts[o + 1] = pv[1] * vs; verts[o + 2] = pv[2] * vs;                  // :41-42,:97-99
                normals[o] = h2f(hmul(inv, n0)); normals[o + 1] = h2f(hmul(inv, n1)); normals[o + 2] = h2f(hmul(inv, n2));   // :100-102
            }
        }
    }
}

}  // namespace tsl

using namespace tsl;

extern "C" {

int tsl_mesh_generate(tsl_tsdf* m, int step, float surface_thres, int64_t max_tri, int32_t* n_tri)
{
    TSL_REQUIRE(m && n_tri, "mesh_generate: null argument"); TSL_REQUIRE(step >= 1 && max_tri > 0, "mesh_generate: bad step / max_triangles");
    TSL_HIP(hipSetDevice(m->device));
    int rc;
    if (m->mesh_cap < max_tri) {
        if (m->mesh_v) { (void)hipFree(m->mesh_v); (void)hipFree(m->mesh_n); if (m->mesh_c) (void)hipFree(m->mesh_c); m->mesh_v = m->mesh_n = m->mesh_c = nullptr; }
        if ((rc = dev_alloc(m, (void**)&m->mesh_v, sizeof(float) * 9 * (size_t)max_tri, 0))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->mesh_n, sizeof(float) * 9 * (size_t)max_tri, 0))) return rc;
        if (m->M.col) { if ((rc = dev_alloc(m, (void**)&m->mesh_c, sizeof(float) * 9 * (size_t)max_tri, 0))) return rc; }
        m->mesh_cap = max_tri;
    }
    if (!m->mesh_count) { if ((rc = dev_alloc(m, (void**)&m->mesh_count, sizeof(int) * 4, 0))) return rc; }
    const bool tiles = step == 1 && !m->mesh_gather;     // the LDS-tile kernel: it reads the number of bricks in use on the device (one host round trip per mesh less)
    int nused = 0; if (!tiles && (rc = tsl_tsdf_bricks_in_use(m, &nused))) return rc;
    if (!m->mesh_flags) { if ((rc = dev_alloc(m, (void**)&m->mesh_flags, sizeof(uint32_t) * (size_t)m->M.max_bricks, 0))) return rc; }
    (void)ms(m);                                         // the queued frames are issued first: the mesh is of the map behind them
    prof_begin(m, TSL_K_MESH);
    if (!tiles) TSL_HIP(hipMemsetAsync(m->mesh_count, 0, sizeof(int), ms(m)));                          // :182
    if (tiles) {
        const int g1 = m->M.max_bricks < 2048 ? m->M.max_bricks : 2048, g2 = m->M.max_bricks < 1024 ? m->M.max_bricks * (16 / MC_SLAB) : 8192;
        hipLaunchKernelGGL(k_mc_summary, dim3(g1), dim3(256), 0, ms(m), m->M, -1, m->mesh_flags, m->mesh_count);
        hipLaunchKernelGGL(k_marching_cubes_lds, dim3(g2), dim3(256), 0, ms(m), m->M, -1,
                           surface_thres, m->P.vs, (long long)max_tri, m->mesh_v, m->mesh_n, m->M.col ? m->mesh_c : (float*)nullptr, m->mesh_count,
                           (const uint32_t*)m->mesh_flags);
    }
    else if (nused > 0)             // coarser meshes (step > 1) reach beyond the brick's halo: every value through the brick table
        hipLaunchKernelGGL(k_marching_cubes, dim3(nused < 16384 ? nused : 16384), dim3(256), 0, ms(m), m->M, nused, step,
                           surface_thres, m->P.vs, (long long)max_tri, m->mesh_v, m->mesh_n, m->M.col ? m->mesh_c : (float*)nullptr, m->mesh_count);
    prof_end(m);
    TSL_HIP(hipGetLastError());
    TSL_HIP(hipMemcpyAsync(m->h_ints, m->mesh_count, sizeof(int), hipMemcpyDeviceToHost, ms(m)));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    *n_tri = m->h_ints[0];
    return TSL_OK;
}

int tsl_mesh_read(tsl_tsdf* m, float* verts, float* normals, float* colors, int64_t n_vertices)
{
    TSL_REQUIRE(m, "mesh_read: null handle"); TSL_REQUIRE(n_vertices >= 0 && n_vertices <= 3 * m->mesh_cap, "mesh_read: more vertices than the mesh buffers hold");
    TSL_HIP(hipSetDevice(m->device));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    if (n_vertices == 0) return TSL_OK;
    if (verts) TSL_HIP(hipMemcpy(verts, m->mesh_v, sizeof(float) * 3 * (size_t)n_vertices, hipMemcpyDeviceToHost));
    if (normals) TSL_HIP(hipMemcpy(normals, m->mesh_n, sizeof(float) * 3 * (size_t)n_vertices, hipMemcpyDeviceToHost));
    if (colors && m->mesh_c) TSL_HIP(hipMemcpy(colors, m->mesh_c, sizeof(float) * 3 * (size_t)n_vertices, hipMemcpyDeviceToHost));
    return TSL_OK;
}

/* mesh_vertices / mesh_normals / mesh_colors as DEVICE pointers (f32 [3 * max_triangles][3]; colours NULL for untextured maps) and the
 * triangle count of the last tsl_mesh_generate -- taichislam_node.py:342 hands the mesh to the renderer without a host copy.  The
 * pointers change when a later tsl_mesh_generate asks for a larger max_triangles. */
int tsl_mesh_buffers_dev(tsl_tsdf* m, void** verts_dev, void** normals_dev, void** colors_dev, int32_t* n_tri)
{
    TSL_REQUIRE(m, "mesh_buffers_dev: null handle"); TSL_REQUIRE(m->mesh_v && m->mesh_count, "mesh_buffers_dev: call tsl_mesh_generate first");
    TSL_HIP(hipSetDevice(m->device));
    if (verts_dev) *verts_dev = m->mesh_v; if (normals_dev) *normals_dev = m->mesh_n; if (colors_dev) *colors_dev = m->mesh_c;
    TSL_HIP(hipMemcpyAsync(m->h_ints, m->mesh_count, sizeof(int), hipMemcpyDeviceToHost, ms(m)));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    if (n_tri) *n_tri = m->h_ints[0];
    return TSL_OK;
}

}  // extern "C"
