// tsl_sequential.hip -- option "semantics" = 1: the reference-literal SEQUENTIAL update of process_new_pcl
// (taichi_slam/mapping/dense_tsdf.py:236-270, reference root) on the GPU.
//
// The reference updates TSDF / W with an unsynchronised read-modify-write per ray step (:264-267), f16 rounding after every update and
// W clamped at 1000 after every update; rays race.  A sequential schedule -- rays in Taichi's struct-for order over the sensor grid
// (pointer block lexicographic, then dense cell), steps in order along a ray -- is one legal outcome of that race, and it is what the
// CPU checker replays in its FAITHFUL mode (the restatement under oracle/, process_new_pcl).  The default path (tsl_integrate.hip) does not compute
// this: it sums a frame's contributions exactly and applies them once (BATCHED).  This file does compute it, bit for bit:
//
//   phase A as in the default path (voxelize -> rays -> segments, bricks allocated), one frame per batch; k_segments also leaves every
//   ray's struct-for key and step count.  Then, on the main stream:
//     k_seq_order   rays -> (struct-for key, ray id), radix sort (rocPRIM)            => rank of every ray
//     k_seq_expand  every (ray, step) -> tuple  key = brick pool index | voxel | rank | step,  value = { w, signed distance } (f32 bits)
//     radix sort of the tuples (rocPRIM)                                              => per voxel: its updates in replay order
//     k_seq_apply   one thread per voxel run: the updates applied one after the other in f16, exactly :264-267; on a textured map the
//                   voxel takes the colour of the LAST ray of its run (:268-269: every step stores its ray's colour) -- the tuple then
//                   carries the ray id instead of the weight, which is read back from the ray record
//   Updates of different voxels commute, so sorting by voxel first and by replay order inside a voxel reproduces the sequential map.
//   The chain of the voxel next to the sensor (every ray of the frame passes through it) is what bounds a frame: ~27 k dependent updates.
#include "tsl_tsdf.hpp"
#include <rocprim/rocprim.hpp>

namespace tsl {

#define SEQ_POOL_SHIFT 46         // tuple key: pool brick (17 bits) | voxel (12) | ray rank (22) | step (12)
#define SEQ_VOX_SHIFT 34
#define SEQ_RANK_SHIFT 12

__global__ void __launch_bounds__(256) k_seq_order(FrameDev F, unsigned long long* keys, uint32_t* vals)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= F.counters[6]) return;
    keys[r] = reinterpret_cast<const unsigned long long*>(F.keys)[r];          // struct-for key of the ray's sensor voxel (k_segments)
    vals[r] = (uint32_t)r;
}

// one thread per ray, in replay order: its steps become tuples (dense_tsdf.py:251-260, the arithmetic of step_voxel / step_term)
template <bool TEX>
__global__ void __launch_bounds__(256) k_seq_expand(MapDev M, FrameDev F, const FrameParams* __restrict__ Pp, const uint32_t* __restrict__ ray_of_rank,
                                                    unsigned long long* tkeys, unsigned long long* tvals, long long cap, unsigned long long* counter)
{
    const FrameParams& P = *Pp;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nrays = F.counters[6];
    if (i >= nrays) return;
    const int r = (int)ray_of_rank[i];
    const int n = F.rayN[r];
    if (n <= 0) return;                                                          // degenerate ray (skipped by the reference restatement too)
    const uint4 rec = F.rayA[r];
    const float pf0 = h2f((h16)(rec.x & 0xffffu)), pf1 = h2f((h16)(rec.x >> 16)), pf2 = h2f((h16)(rec.y & 0xffffu));
    const float d0 = h2f((h16)(rec.y >> 16)), d1 = h2f((h16)(rec.z & 0xffffu)), d2 = h2f((h16)(rec.z >> 16));
    const float w = __uint_as_float(rec.w);
    const float P0 = pf0 + P.T[0], P1 = pf1 + P.T[1], P2 = pf2 + P.T[2];                          // :246
    const unsigned long long base = __hip_atomic_fetch_add(counter, (unsigned long long)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((long long)(base + (unsigned long long)n) > cap) { frame_fail(M, F, 4); return; }
    int cur_b = -1, cur_p = -1;
    for (int j = 1; j <= n; ++j) {
        const float jf = (float)j;
        const float x0 = (d0 * jf) * P.vs + P.T[0], x1 = (d1 * jf) * P.vs + P.T[1], x2 = (d2 * jf) * P.vs + P.T[2];     // :253
        const int i0 = rnd_i(x0 / P.vs), i1 = rnd_i(x1 / P.vs), i2 = rnd_i(x2 / P.vs);                                  // :254
        unsigned long long key = ~0ull, val = 0ull;
        if (in_volume(M, i0, i1, i2)) {
            int l; const int b = brick_of(M, i0, i1, i2, &l);
            if (b != cur_b) { cur_b = b; cur_p = pool_lookup(M, P.slot, b); }       // allocated by k_plan (every brick a segment enters)
            if (cur_p >= 0) {
                const float v0 = P0 - x0, v1 = P1 - x1, v2 = P2 - x2;                                                   // :258
                const float dist = sqrt_rn((v0 * v0 + v1 * v1) + v2 * v2);                                              // :259
                const float dot = (v0 * pf0 + v1 * pf1) + v2 * pf2;
                const float sd = dist * (float)sgn_f(dot);                                                              // :260
                key = ((unsigned long long)cur_p << SEQ_POOL_SHIFT) | ((unsigned long long)l << SEQ_VOX_SHIFT) | ((unsigned long long)i << SEQ_RANK_SHIFT) | (unsigned long long)j;
                val = ((unsigned long long)(TEX ? (uint32_t)r : __float_as_uint(w)) << 32) | (unsigned long long)__float_as_uint(sd);
            }
        }
        tkeys[base + (unsigned long long)(j - 1)] = key;
        tvals[base + (unsigned long long)(j - 1)] = val;
    }
}

// entries [count, bound) of the tuple array may hold tuples of an earlier, longer frame: invalid keys sort to the end
__global__ void __launch_bounds__(256) k_seq_pad(unsigned long long* tkeys, const unsigned long long* __restrict__ counter, long long bound)
{
    const long long n = (long long)*counter;
    for (long long i = n + (long long)blockIdx.x * 256 + threadIdx.x; i < bound; i += (long long)gridDim.x * 256) tkeys[i] = ~0ull;
}

// one thread per tuple; the head of a voxel's run applies the whole run in order  (dense_tsdf.py:264-267)
template <bool TEX>
__global__ void __launch_bounds__(256) k_seq_apply(MapDev M, FrameDev F, const unsigned long long* __restrict__ tkeys, const unsigned long long* __restrict__ tvals,
                                                   const unsigned long long* __restrict__ counter, long long cap)
{
    const long long total = min((long long)*counter, cap);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    bool head = false;
    if (i < total && F.counters[11] == 0) {
        const unsigned long long k = tkeys[i];
        head = k != ~0ull && (i == 0 || (tkeys[i - 1] >> SEQ_VOX_SHIFT) != (k >> SEQ_VOX_SHIFT));
        if (head) {
            const unsigned long long run = k >> SEQ_VOX_SHIFT;
            const size_t v = (size_t)(k >> SEQ_POOL_SHIFT) * TSL_BRK3 + (size_t)((k >> SEQ_VOX_SHIFT) & 4095ull);
            const uint32_t old = M.tw[v];
            h16 T0 = (h16)(old & 0xffffu), W0 = (h16)(old >> 16);
            uint32_t last_ray = 0u;
            for (long long q = i; q < total; ++q) {
                if ((tkeys[q] >> SEQ_VOX_SHIFT) != run) break;
                const unsigned long long tv = tvals[q];
                if (TEX) last_ray = (uint32_t)(tv >> 32);
                const float w = __uint_as_float(TEX ? F.rayA[last_ray].w : (uint32_t)(tv >> 32)), sd = __uint_as_float((uint32_t)tv);
                const h16 Tn = f2h((h2f(hmul(T0, W0)) + w * sd) / (h2f(W0) + w));                                       // :264
                float wn = h2f(W0) + w; if (TSL_WMAX < wn) wn = TSL_WMAX;                                               // :267
                T0 = Tn; W0 = f2h(wn);
            }
            M.tw[v] = (uint32_t)T0 | ((uint32_t)W0 << 16);
            M.obs[v] = 1;                                                                                               // :265
            if (TEX) reinterpret_cast<uint2*>(M.col)[v] = F.colpix[F.rayFirst[last_ray]];                               // :268-269, the run's last writer
            M.touch[k >> SEQ_POOL_SHIFT] = 1;
        }
    }
    const unsigned long long m = __ballot(head);
    if (m && lane_id() == (int)__builtin_ctzll(m)) atomic_add_i64(&F.stats->unique, (long long)popc64(m));
}

// the brick kernel of the default path leaves the set's per-brick histogram / scatter cursor zeroed for the set's next frame; here that kernel
// does not run
__global__ void __launch_bounds__(256) k_seq_cleanup(FrameDev F)
{
    const int nact = min(F.counters[1], F.max_frame_bricks);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nact; i += gridDim.x * 256) { const int b = F.act_b[i]; F.bhist[b] = 0; F.bcursor[b] = 0; }
}

// phase B of one frame, sequential semantics; enqueued on the main stream behind the frame's phase A
int launch_apply_sequential(tsl_tsdf* m, const BatchDev& B, const FrameParams& P)
{
    TSL_REQUIRE(B.n == 1 && P.group && P.variant == 2, "sequential semantics: one frame at a time on the hash-grouped brick path");
    const FrameDev& F = B.f[0];
    hipStream_t q = m->stream_;
    const size_t np = (size_t)m->F.max_points;
    if (!m->seq_keys[0]) {
        // a frame yields at most rays x steps tuples; 2^24 covers 640 x 480 at recast_step 2 four times over (a frame beyond it fails loudly)
        m->seq_cap = 1ll << 24;
        int rc;
        for (int k = 0; k < 2; ++k) {
            if ((rc = dev_alloc(m, (void**)&m->seq_keys[k], 8 * (size_t)m->seq_cap, 0))) return rc;
            if ((rc = dev_alloc(m, (void**)&m->seq_vals[k], 8 * (size_t)m->seq_cap, 0))) return rc;
        }
        if ((rc = dev_alloc(m, (void**)&m->seq_ctr, 64, 0))) return rc;
        size_t a = 0, b = 0;
        TSL_HIP(rocprim::radix_sort_pairs(nullptr, a, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)m->seq_cap, 0u, 64u, q));
        TSL_HIP(rocprim::radix_sort_pairs(nullptr, b, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, np, 0u, 64u, q));
        m->seq_temp_bytes = (a > b ? a : b) + 256;
        if ((rc = dev_alloc(m, &m->seq_temp, m->seq_temp_bytes, 0))) return rc;
    }
    unsigned long long* rk = m->seq_keys[0]; unsigned long long* rk_s = m->seq_keys[1];
    uint32_t* rv = reinterpret_cast<uint32_t*>(m->seq_vals[0]); uint32_t* rv_s = reinterpret_cast<uint32_t*>(m->seq_vals[1]);
    const int blocks = (int)((np + 255) / 256);
    // 1. replay order of the rays.  The ray count lives on the device: the sort covers max_points entries, the unused ones carry the largest key
    TSL_HIP(hipMemsetAsync(rk, 0xff, 8 * np, q));
    hipLaunchKernelGGL(k_seq_order, dim3(blocks), dim3(256), 0, q, F, rk, rv);
    size_t tb = m->seq_temp_bytes;
    TSL_HIP(rocprim::radix_sort_pairs(m->seq_temp, tb, rk, rk_s, rv, rv_s, np, 0u, 64u, q));
    // 2. tuples, 3. their replay order per voxel, 4. apply
    TSL_HIP(hipMemsetAsync(m->seq_ctr, 0, 64, q));
    if (P.tex) hipLaunchKernelGGL(k_seq_expand<true>, dim3(blocks), dim3(256), 0, q, m->M, F, B.p[0], (const uint32_t*)rv_s, m->seq_keys[0], m->seq_vals[0], m->seq_cap, m->seq_ctr);
    else hipLaunchKernelGGL(k_seq_expand<false>, dim3(blocks), dim3(256), 0, q, m->M, F, B.p[0], (const uint32_t*)rv_s, m->seq_keys[0], m->seq_vals[0], m->seq_cap, m->seq_ctr);
    // the tuple count is on the device as well: sort what the frame could have produced at most -- bounded by the frame's own statistics
    // on the host side is not possible without a round trip, so the sort length is fixed by a cheap upper bound: rays <= visited pixels,
    // steps per ray <= max_steps; unused entries keep the key ~0 from the previous fill and sort to the end
    long long bound = (long long)P.total * (long long)(P.max_steps_f + 1.0f);
    if (bound > m->seq_cap) bound = m->seq_cap;
    // (entries beyond this frame's count may hold tuples of an earlier frame: they are re-marked invalid first)
    hipLaunchKernelGGL(k_seq_pad, dim3(1024), dim3(256), 0, q, m->seq_keys[0], m->seq_ctr, bound);
    tb = m->seq_temp_bytes;
    TSL_HIP(rocprim::radix_sort_pairs(m->seq_temp, tb, m->seq_keys[0], m->seq_keys[1], m->seq_vals[0], m->seq_vals[1], (size_t)bound, 0u, 64u, q));
    if (P.tex) hipLaunchKernelGGL(k_seq_apply<true>, dim3((unsigned)((bound + 255) / 256)), dim3(256), 0, q, m->M, F, (const unsigned long long*)m->seq_keys[1], (const unsigned long long*)m->seq_vals[1],
                                  (const unsigned long long*)m->seq_ctr, bound);
    else hipLaunchKernelGGL(k_seq_apply<false>, dim3((unsigned)((bound + 255) / 256)), dim3(256), 0, q, m->M, F, (const unsigned long long*)m->seq_keys[1], (const unsigned long long*)m->seq_vals[1],
                            (const unsigned long long*)m->seq_ctr, bound);
    hipLaunchKernelGGL(k_seq_cleanup, dim3(16), dim3(256), 0, q, F);
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}

}  // namespace tsl
