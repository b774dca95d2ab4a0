// tsl_fuse.hip -- submap -> global map fusion.  Replaces fuse_submaps_kernel / fuse_with_interploation
// (taichi_slam/mapping/dense_tsdf.py:272-318, reference root).
//
// Every observed voxel of every submap is transformed with its submap pose and splatted onto 7 of the 8 surrounding
// global voxels (the (0,0,0) corner is skipped in the reference, Q8) with trilinear weights.  The reference applies the
// weighted running average voxel by voxel in a racy read-modify-write; here the contributions {w*t, w} are summed in
// exact 2^-24 fixed point (int64 atomics) plus a contribution/occupancy count, and the average is formed once per
// voxel -- order-free, so N GPUs that each splat their own submaps and all-reduce(sum) the three arrays produce the
// same bits as one GPU fusing everything (tsl_tsdf_fuse_accumulate_dev / _finalize_dev).  With option "semantics" = 1 on the global map
// tsl_tsdf_fuse_submaps instead replays the reference's running average literally, splat by splat (tsl_sequential.hip).
#include "tsl_tsdf.hpp"

namespace tsl {

struct PoseTab { const float* p; };     // [npose][12]: R row-major, T

__global__ void __launch_bounds__(256) k_fuse_splat(MapDev S, MapDev G, PoseTab poses, float vs, int nused,
                                                    unsigned long long* acc, int* cnt, int npose, unsigned long long* cacc)
{
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        const int owner = S.owner[p];
        const int s = owner / S.nb3, b = owner - s * S.nb3;
        if (s >= npose) continue;
        const float* Rp = poses.p + (size_t)s * 12;
        float R[9], T[3];
        for (int a = 0; a < 9; ++a) R[a] = Rp[a];
        for (int a = 0; a < 3; ++a) T[a] = Rp[9 + a];
        const int bk = b % S.nbz, bj = (b / S.nbz) % S.nbx, bi = b / (S.nbz * S.nbx);
        for (int l = threadIdx.x; l < TSL_BRK3; l += 256) {
            const size_t v = (size_t)p * TSL_BRK3 + l;
            if (S.obs[v] <= 0) continue;                                                        // :292
            const int i = bi * 16 + (l >> 8) - S.hN, j = bj * 16 + ((l >> 4) & 15) - S.hN, k = bk * 16 + (l & 15) - S.hNz;
            const float p0 = (float)i * vs, p1 = (float)j * vs, p2 = (float)k * vs;             // mapping_common.py:221-232 with the GLOBAL map's voxel scale
            float f[3]; int lo[3];
            for (int a = 0; a < 3; ++a) {
                const float x = ((R[a * 3] * p0 + R[a * 3 + 1] * p1) + R[a * 3 + 2] * p2) + T[a];      // :293
                f[a] = x / vs; lo[a] = (int)floorf(f[a]);                                     // :294-296
            }
            const uint32_t tw = S.tw[v];
            const float tsdf = h2f((h16)(tw & 0xffffu)), wsrc = h2f((h16)(tw >> 16));
            const int occ = (int)S.occ[v];
            for (int c = 1; c < 8; ++c) {                                                        // :297-300 (corner 0 skipped)
                const int ci = lo[0] + ((c >> 2) & 1), cj = lo[1] + ((c >> 1) & 1), ck = lo[2] + (c & 1);
                const float wt = ((1.0f - fabsf((float)ci - f[0])) * (1.0f - fabsf((float)cj - f[1]))) * (1.0f - fabsf((float)ck - f[2]));   // :303
                const float w_tsdf = wsrc * wt;                                                  // :307
                if (!in_volume(G, ci, cj, ck)) continue;
                int gl; const int gb = brick_of(G, ci, cj, ck, &gl);
                const int gp = pool_claim<false>(G, 0, gb);
                if (gp < 0) continue;
                const size_t dst = (size_t)gp * TSL_BRK3 + gl;
                __hip_atomic_fetch_add(acc + dst * 2, (unsigned long long)to_fix(w_tsdf * tsdf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // :275 numerator
                __hip_atomic_fetch_add(acc + dst * 2 + 1, (unsigned long long)to_fix(w_tsdf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // :274
                __hip_atomic_fetch_add(cnt + dst, (1 << 16) + occ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                                 // :279-280
                if (cacc) {                                                            // :277 colour, exact weighted sums per channel
                    const uint2 cs = reinterpret_cast<const uint2*>(S.col)[v];
                    const float cf[3] = { h2f((h16)(cs.x & 0xffffu)), h2f((h16)(cs.x >> 16)), h2f((h16)(cs.y & 0xffffu)) };
                    for (int a = 0; a < 3; ++a)
                        __hip_atomic_fetch_add(cacc + dst * 3 + a, (unsigned long long)to_fix(w_tsdf * cf[a]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// The same splat with the sums gathered in LDS first (round 6; untextured maps -- the common case and the multi-GPU merge).  k_fuse_splat issues 21 global atomics
// per source voxel (7 corners x {num, den, count}): 57 M int64 / int32 atomics for one 2.7 M-voxel submap, 0.86 ms, and the write counter saw 8x the bytes the
// fusion has to move.  Neighbouring source voxels splat onto the SAME global voxels -- a global voxel takes ~7 contributions -- so a workgroup takes an 8^3
// block of a source brick, whose 7 corner splats land in a window of at most 15^3 global voxels whatever the pose (a cube of 7 voxel spans has a diagonal of
// 12.2: 13 floors, + 1 for the far corner, + 1 of slack for the rounding of the pose product), sums them there with LDS atomics and flushes every window voxel
// that got something with ONE set of global atomics: 3.5x fewer.  Exact integer sums: the grouping cannot change a bit.  (A corner outside the window -- not
// seen, the window is sized for the worst pose -- goes straight to memory like before.)
#define FW 15
#define FW3 (FW * FW * FW)
__global__ void __launch_bounds__(256) k_fuse_splat_lds(MapDev S, MapDev G, PoseTab poses, float vs, int nused, unsigned long long* acc, int* cnt, int npose)
{
    __shared__ unsigned long long s_num[FW3], s_den[FW3];
    __shared__ int s_cnt[FW3];
    for (int e = threadIdx.x; e < FW3; e += 256) { s_num[e] = 0ull; s_den[e] = 0ull; s_cnt[e] = 0; }
    __syncthreads();
    for (int unit = blockIdx.x; unit < nused * 8; unit += gridDim.x) {
        const int p = unit >> 3, sb = unit & 7;
        const int owner = S.owner[p];
        const int s = owner / S.nb3, b = owner - s * S.nb3;
        if (s >= npose) continue;                                    // (uniform)
        const float* Rp = poses.p + (size_t)s * 12;
        float R[9], T[3];
        for (int a = 0; a < 9; ++a) R[a] = Rp[a];
        for (int a = 0; a < 3; ++a) T[a] = Rp[9 + a];
        const int bk = b % S.nbz, bj = (b / S.nbz) % S.nbx, bi = b / (S.nbz * S.nbx);
        const int l0 = ((sb >> 2) & 1) * 8 * 256 + ((sb >> 1) & 1) * 8 * 16 + (sb & 1) * 8;      // first voxel of the 8^3 block inside the brick
        // the block's two voxels of this thread
        int lv[2]; bool on[2];
        for (int h = 0; h < 2; ++h) { const int t = (int)threadIdx.x + h * 256; lv[h] = l0 + (t >> 6) * 256 + ((t >> 3) & 7) * 16 + (t & 7); on[h] = S.obs[(size_t)p * TSL_BRK3 + lv[h]] > 0; }      // :292
        if (!__syncthreads_or(on[0] || on[1])) continue;
        // window origin: the least floor over the block's eight corner voxels, one below (every thread computes the same eight points)
        int org[3] = { 1 << 30, 1 << 30, 1 << 30 };
        for (int c8 = 0; c8 < 8; ++c8) {
            const int l = l0 + ((c8 >> 2) & 1) * 7 * 256 + ((c8 >> 1) & 1) * 7 * 16 + (c8 & 1) * 7;
            const int i = bi * 16 + (l >> 8) - S.hN, j = bj * 16 + ((l >> 4) & 15) - S.hN, k = bk * 16 + (l & 15) - S.hNz;
            const float p0 = (float)i * vs, p1 = (float)j * vs, p2 = (float)k * vs;
            for (int a = 0; a < 3; ++a) { const float x = ((R[a * 3] * p0 + R[a * 3 + 1] * p1) + R[a * 3 + 2] * p2) + T[a]; org[a] = min(org[a], (int)floorf(x / vs) - 1); }
        }
        for (int h = 0; h < 2; ++h) {
            if (!on[h]) continue;
            const int l = lv[h];
            const size_t v = (size_t)p * TSL_BRK3 + l;
            const int i = bi * 16 + (l >> 8) - S.hN, j = bj * 16 + ((l >> 4) & 15) - S.hN, k = bk * 16 + (l & 15) - S.hNz;
            const float p0 = (float)i * vs, p1 = (float)j * vs, p2 = (float)k * vs;             // mapping_common.py:221-232 with the GLOBAL map's voxel scale
            float f[3]; int lo[3];
            for (int a = 0; a < 3; ++a) {
                const float x = ((R[a * 3] * p0 + R[a * 3 + 1] * p1) + R[a * 3 + 2] * p2) + T[a];      // :293
                f[a] = x / vs; lo[a] = (int)floorf(f[a]);                                     // :294-296
            }
            const uint32_t tw = S.tw[v];
            const float tsdf = h2f((h16)(tw & 0xffffu)), wsrc = h2f((h16)(tw >> 16));
            const int occ = (int)S.occ[v];
            for (int c = 1; c < 8; ++c) {                                                        // :297-300 (corner 0 skipped)
                const int ci = lo[0] + ((c >> 2) & 1), cj = lo[1] + ((c >> 1) & 1), ck = lo[2] + (c & 1);
                const float wt = ((1.0f - fabsf((float)ci - f[0])) * (1.0f - fabsf((float)cj - f[1]))) * (1.0f - fabsf((float)ck - f[2]));   // :303
                const float w_tsdf = wsrc * wt;                                                  // :307
                if (!in_volume(G, ci, cj, ck)) continue;
                const unsigned long long qn = (unsigned long long)to_fix(w_tsdf * tsdf), qd = (unsigned long long)to_fix(w_tsdf);
                const int wi = ci - org[0], wj = cj - org[1], wk = ck - org[2];
                if ((unsigned)wi < (unsigned)FW && (unsigned)wj < (unsigned)FW && (unsigned)wk < (unsigned)FW) {
                    const int e = (wi * FW + wj) * FW + wk;
                    __hip_atomic_fetch_add(&s_num[e], qn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                     // :275 numerator
                    __hip_atomic_fetch_add(&s_den[e], qd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                     // :274
                    __hip_atomic_fetch_add(&s_cnt[e], (1 << 16) + occ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);        // :279-280
                } else {
                    atomicAdd(G.pool_top + 2, 1);                      // (counted: option "fuse_window_misses" -- the tests require 0)
                    int gl; const int gb = brick_of(G, ci, cj, ck, &gl);
                    const int gp = pool_claim<false>(G, 0, gb);
                    if (gp < 0) continue;
                    const size_t dst = (size_t)gp * TSL_BRK3 + gl;
                    __hip_atomic_fetch_add(acc + dst * 2, qn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(acc + dst * 2 + 1, qd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(cnt + dst, (1 << 16) + occ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < FW3; e += 256) {                  // flush: one set of global atomics per window voxel that got something
            const int c = s_cnt[e];
            if (c == 0) continue;
            const int wi = e / (FW * FW), wj = (e / FW) % FW, wk = e % FW;
            int gl; const int gb = brick_of(G, org[0] + wi, org[1] + wj, org[2] + wk, &gl);
            const int gp = pool_claim<false>(G, 0, gb);
            if (gp >= 0) {
                const size_t dst = (size_t)gp * TSL_BRK3 + gl;
                __hip_atomic_fetch_add(acc + dst * 2, s_num[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(acc + dst * 2 + 1, s_den[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(cnt + dst, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_num[e] = 0ull; s_den[e] = 0ull; s_cnt[e] = 0;
        }
        __syncthreads();
    }
}

// finalise from the per-brick scratch of the global map
__global__ void __launch_bounds__(256) k_fuse_finalize(MapDev G, int nused, unsigned long long* acc, int* cnt, unsigned long long* cacc)
{
    for (int p = blockIdx.x; p < nused; p += gridDim.x)
        for (int l = threadIdx.x; l < TSL_BRK3; l += 256) {
            const size_t v = (size_t)p * TSL_BRK3 + l;
            const int c = cnt[v];
            if (c == 0) continue;
            fuse_write_voxel(G, v, (long long)acc[v * 2], (long long)acc[v * 2 + 1], c);
            if (cacc) {
                const float den = from_fix((long long)acc[v * 2 + 1]);
                h16 cc[3];
                for (int a = 0; a < 3; ++a) { cc[a] = f2h(from_fix((long long)cacc[v * 3 + a]) / den); cacc[v * 3 + a] = 0ull; }
                reinterpret_cast<uint2*>(G.col)[v] = make_uint2((uint32_t)cc[0] | ((uint32_t)cc[1] << 16), (uint32_t)cc[2]);
            }
            acc[v * 2] = 0ull; acc[v * 2 + 1] = 0ull; cnt[v] = 0;
        }
}

static int upload_poses(tsl_tsdf* g, const tsl_tsdf* sub)
{
    // dense_tsdf.py:286-290: the f32 pose fields of submaps [0, active) are refreshed from the float64 tables
    const int nsub = sub->active;
    for (int s = 0; s < nsub && s < g->npose; ++s) {
        for (int a = 0; a < 9; ++a) g->baseRf[(size_t)s * 9 + a] = (float)g->baseR[(size_t)s * 9 + a];
        for (int a = 0; a < 3; ++a) g->baseTf[(size_t)s * 3 + a] = (float)g->baseT[(size_t)s * 3 + a];
    }
    std::vector<float> tab((size_t)g->npose * 12);
    for (int s = 0; s < g->npose; ++s) {
        for (int a = 0; a < 9; ++a) tab[(size_t)s * 12 + a] = g->baseRf[(size_t)s * 9 + a];
        for (int a = 0; a < 3; ++a) tab[(size_t)s * 12 + 9 + a] = g->baseTf[(size_t)s * 3 + a];
    }
    TSL_HIP(hipMemcpyAsync(g->pose_dev, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, ms(g)));
    TSL_HIP(hipStreamSynchronize(ms(g)));
    return TSL_OK;
}

static int used_bricks(tsl_tsdf* m, int* n) { return tsl_tsdf_bricks_in_use(m, n); }

// reset `g` and splat every submap of `sub` into g's per-brick accumulators (allocated on first use); *ndst = bricks of g touched.
// with_colour: also sum the colour channels (single-GPU fusion of textured maps); the multi-GPU merge exchanges {sum w*t, sum w, count}
// only, so it passes false and a merged map carries no colour.
// The accumulators are all zero between uses: k_fuse_finalize / k_merge_pack zero what they read.  A caller that stopped in between (a
// step-protocol merge abandoned after merge_begin, a failed launch) leaves them dirty -- `fuse_dirty` -- and the next splat clears
// them wholesale first (the pool indices they were written under are gone after the reset).
int fuse_splat_into_global(tsl_tsdf* g, tsl_tsdf* sub, int* ndst, bool with_colour)
{
    int rc = tsl_tsdf_sync(sub); if (rc) return rc;
    rc = tsl_tsdf_reset(g); if (rc && rc != TSL_ERR_CAPACITY) return rc;                  // :313 (a capacity error of the discarded contents does not matter here)
    const size_t nv = (size_t)g->M.max_bricks * TSL_BRK3;
    if (!g->fuse_acc) {
        if ((rc = dev_alloc(g, &g->fuse_acc, nv * 16, 0))) return rc;
        if ((rc = dev_alloc(g, &g->fuse_cnt, nv * 4, 0))) return rc;
        if (g->M.col) { if ((rc = dev_alloc(g, &g->fuse_cacc, nv * 24, 0))) return rc; }
        g->fuse_dirty = false;
    }
    if (g->fuse_dirty) {
        TSL_HIP(hipMemsetAsync(g->fuse_acc, 0, nv * 16, ms(g)));
        TSL_HIP(hipMemsetAsync(g->fuse_cnt, 0, nv * 4, ms(g)));
        if (g->fuse_cacc) TSL_HIP(hipMemsetAsync(g->fuse_cacc, 0, nv * 24, ms(g)));
        g->fuse_dirty = false;
    }
    unsigned long long* cacc = (with_colour && g->M.col && sub->M.col) ? (unsigned long long*)g->fuse_cacc : nullptr;
    if ((rc = upload_poses(g, sub))) return rc;
    int nsrc = 0; if ((rc = used_bricks(sub, &nsrc))) return rc;
    *ndst = 0;
    if (nsrc > 0) {
        PoseTab pt = { g->pose_dev };
        g->fuse_dirty = true;
        prof_begin(g, TSL_K_FUSE);
        if (cacc || g->fuse_direct)      // textured maps (the colour sums have no LDS form) and the A/B option: one set of global atomics per corner
            hipLaunchKernelGGL(k_fuse_splat, dim3(nsrc < 8192 ? nsrc : 8192), dim3(256), 0, ms(g), sub->M, g->M, pt, g->P.vs, nsrc,
                               (unsigned long long*)g->fuse_acc, (int*)g->fuse_cnt, g->npose, cacc);
        else
            hipLaunchKernelGGL(k_fuse_splat_lds, dim3(nsrc * 8 < 2 * g->ncu ? nsrc * 8 : 2 * g->ncu), dim3(256), 0, ms(g), sub->M, g->M, pt, g->P.vs, nsrc,
                               (unsigned long long*)g->fuse_acc, (int*)g->fuse_cnt, g->npose);
        prof_end(g);
        if ((rc = used_bricks(g, ndst))) return rc;
    }
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}

}  // namespace tsl

using namespace tsl;

extern "C" {

int tsl_tsdf_fuse_submaps(tsl_tsdf* g, tsl_tsdf* sub)
{
    TSL_REQUIRE(g && sub, "fuse_submaps: null handle");
    TSL_REQUIRE(g->cfg.is_global_map, "fuse_submaps: destination must be a global map (is_global_map=True)");
    TSL_REQUIRE(g->device == sub->device, "fuse_submaps: maps live on different devices");
    TSL_HIP(hipSetDevice(g->device));
    int ndst = 0;
    if (g->semantics == 1) {      // the reference-literal sequential fusion (tsl_sequential.hip): same reset, same pose table, then tuples instead of sums
        int rc = tsl_tsdf_sync(sub); if (rc) return rc;
        rc = tsl_tsdf_reset(g); if (rc && rc != TSL_ERR_CAPACITY) return rc;
        if ((rc = upload_poses(g, sub))) return rc;
        int nsrc = 0; if ((rc = used_bricks(sub, &nsrc))) return rc;
        if (nsrc > 0 && (rc = fuse_submaps_sequential(g, sub, g->pose_dev, nsrc))) return rc;
        return tsl_tsdf_sync(g);
    }
    int rc = fuse_splat_into_global(g, sub, &ndst, true); if (rc) return rc;
    unsigned long long* cacc = (g->M.col && sub->M.col) ? (unsigned long long*)g->fuse_cacc : nullptr;
    if (ndst > 0) hipLaunchKernelGGL(k_fuse_finalize, dim3(ndst < 8192 ? ndst : 8192), dim3(256), 0, ms(g), g->M, ndst,
                                     (unsigned long long*)g->fuse_acc, (int*)g->fuse_cnt, cacc);
    TSL_HIP(hipGetLastError());
    g->fuse_dirty = false;                                      // the finalise pass zeroed every sum it read
    return tsl_tsdf_sync(g);                                    // reports an exhausted brick pool of the global map (TSL_ERR_CAPACITY)
}

}  // extern "C"
