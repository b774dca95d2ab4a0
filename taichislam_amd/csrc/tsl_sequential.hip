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
#include <type_traits>
#include <cstdlib>

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

// =====================================================================================================================================
// Round 4: the same sequential map built on the brick pipeline (option "seq_impl" = 1, the default) -- no global sort of ray steps.
//
// Updates of different voxels commute, so the sequential map is fixed by the ORDER OF THE UPDATES OF EACH VOXEL: (ray rank in struct-for
// order, step along the ray).  Phase A already delivers, per frame, the ray segments of every 16^3 brick (k_segments / k_plan / k_scatter).
// Added behind it on the batch's phase-A stream (nothing here reads the map, so it runs beside phase B of the batch before):
//   k_seq_keys + ONE radix sort per batch + k_seq_ranks   struct-for rank of every ray of every frame of the batch (keys = frame | struct-for key)
//   k_seq_group   one work item per (frame, brick) -- a brick with more than 2 048 segments is first cut by rank into chunks that fit (k_seq_split); persistent
//                 workgroups, three per CU, claim the items heavy-first (round 5).  The item's segments are put in (rank, first step) order in LDS (a sample
//                 sort), a first walk counts every step's voxel per quarter of the replay sequence, the run offsets of the brick's 4 096 voxels go to the
//                 slot's table, and a placing pass -- wave w over quarter w, 64 consecutive replay positions at a time -- evaluates every step at its
//                 position and writes its 8-byte replay tuple { w, w * sd } behind its voxel's cursor (+ the tuples of the same voxel before it in the group:
//                 12 ballots): every voxel's run contiguous and in replay order.  (Round 4 wrote every step to a "stash" in replay order first and read it
//                 back in the counting sort: -DTSL_SEQ_STASH.)
// Phase B (main stream, batches in order): k_seq_replay, one thread per voxel of every brick the batch touches: the voxel's runs of the batch's
// frames, frame after frame, applied exactly as dense_tsdf.py:264-267 -- no LDS, no barrier, no atomics; the voxel next to the sensor
// (every ray of a frame passes through it) is one long chain in one lane, everything else finishes around it.
// =====================================================================================================================================
#define SQ_NT 256
#define SQ_SORTCAP 2048           // segments sorted in LDS at a time
#ifndef SQ_CHUNKSEGS
#define SQ_CHUNKSEGS 2048         // segments of a CHUNK of a heavy brick (k_seq_split).  The chunks of the bricks next to the sensor are the longest items of k_seq_group's
#endif                            //   launch (300-350 us beside 45 us for the median item); chunks of 1 024 shorten that launch (484 -> 436 us alone) but double the slots the
                                  //   replay walks per voxel there: replay 830 -> 1 080 us per batch, 7 150 -> 6 760 frames/s in a stream (round 5) -- so the sort buffer it is
#define SQ_BSHIFT 8               // rank bucket of a heavy brick = rank >> 8: 256 rays, at most 8 segments per ray and brick (lanes per ray) = SQ_SORTCAP
#define SQ_NBK_MAX 8192           // rank buckets (aliases the 32 KiB of the packed counters): 2 M rays per frame
#define SQ_TUP_L_SHIFT 32
#define SQ_TUP_Z_SHIFT 44

__device__ __forceinline__ void seq_update(h16& T0, h16& W0, float w, float sd)
{
    const h16 Tn = f2h((h2f(hmul(T0, W0)) + w * sd) / (h2f(W0) + w));                                                   // dense_tsdf.py:264
    float wn = h2f(W0) + w; if (TSL_WMAX < wn) wn = TSL_WMAX;                                                           // :267
    T0 = Tn; W0 = f2h(wn);
}
// a ray's weight is 1 / (an f16 value), clamped (finish_ray): 16 bits describe it.  z^2 = RN16(1 / w) exactly (the reciprocal of the
// reciprocal is within 2^-23 of the f16 value, an f16 rounding boundary is 2^-12 away), and a clamped weight maps to 2^-16 -> 65536.
__device__ __forceinline__ h16 seq_w_code(float w) { return f2h(1.0f / w); }
__device__ __forceinline__ float seq_w_of(h16 zz) { float w = 1.0f / h2f(zz); if (w > TSL_W_CLAMP) w = TSL_W_CLAMP; return w; }

// One update of a voxel whose weight has reached Wmax (W = 1000 stays 1000: dense_tsdf.py:267), the state of every long replay run:
//   T' = RN16( RN32( RN16(T * 1000) + c ) / D ),   c = RN32(w * sd),  D = RN32(1000 + w),  with r = RN32(1 / D) from the tuple.
// The quotient is formed without the division: q0 = RN(n r), then two residual corrections q <- RN(q + RN(n - q D) r).  With r the correctly
// rounded reciprocal, q0 is within 2 ulp, the first correction leaves a faithful quotient and the second the correctly rounded one
// (Markstein's theorem; the residuals are exact under FMA) -- valid while nothing overflows or underflows, which k_seq_group guarantees per
// (frame, brick) (|sd| <= 60, no tiny products: SQ_CSR_UNSAFE) and the caller per run (|T| <= 60: the update is a convex combination, so T
// stays there).  tsl_selftest(2) compares it with IEEE division on 2^32 operand pairs; every parity test of the sequential mode runs through it.
__device__ __forceinline__ float seq_div_sat(float n, float D, float r)
{
    float q = n * r;
    float e = __builtin_fmaf(-q, D, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-q, D, n);
    return __builtin_fmaf(e, r, q);
}
// { a = RN16(T * W), n = RN32(a + c), q = n / D, T' = RN16(q) } with D and r = RN32(1 / D) given; W as f16 (1000 once saturated).
// Eight instructions, written out: (i) the compiler turns `(float)a + c` into a conversion and an addition where the mixed-precision FMA
// a * 1.0 + c reads the f16 product as it is (one rounding, the same value); (ii) left to itself it folds the last FMA and the conversion
// into ONE v_fma_mixlo_f16, which rounds the exact FMA result to f16 once -- dense_tsdf.py:264 stores an f32 expression into an f16 field,
// two roundings, and the two differ whenever the f32 rounding lands on an f16 tie (seen: 14 of 1.4 M voxels after 12 frames); (iii) every
// separate asm statement costs a wait state behind it, and the chain of the voxel next to the sensor pays per instruction slot.
// tsl_selftest(2) runs this against the literal expression on 2^32 operand tuples.
// T and W travel as 32-bit registers whose LOW HALF is the f16 value (the f16 instructions read and write that half; nothing is masked in between)
__device__ __forceinline__ uint32_t seq_update_fast(uint32_t T, uint32_t W, float c, float D, float r)
{
    uint32_t t; float n, q;
    asm("v_mul_f16 %0, %3, %4\n\t"                                   // a = RN16(T * W)   (f16 denormals are kept)
        "v_fma_mix_f32 %1, %0, 1.0, %5 op_sel_hi:[1,0,0]\n\t"        // n = RN32(a + c)
        "v_mul_f32 %2, %1, %7\n\t"                                   // q = RN32(n r)
        "v_fma_f32 %0, -%2, %6, %1\n\t"                              // e = n - q D  (exact)
        "v_fmac_f32 %2, %0, %7\n\t"                                  // q = RN32(q + e r): faithful
        "v_fma_f32 %0, -%2, %6, %1\n\t"
        "v_fmac_f32 %2, %0, %7\n\t"                                  // ... correctly rounded (seq_div_sat)
        "v_cvt_f16_f32 %0, %2"                                        // T' = RN16(q)
        : "=&v"(t), "=&v"(n), "=&v"(q)
        : "v"(T), "v"(W), "v"(c), "v"(D), "v"(r));
    return t;
}
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }      // a value every lane of the wave holds: keep it in an SGPR
#define SQ_W_SAT 0x63d0u          // 1000 as f16 bits
#define SQ_LONG 64                // a voxel with a run of at least this many updates in some frame of the batch gets a wave of its own (the long role of k_seq_replay)
#define SQ_XLONG 1024             // ... and with one of at least this many it is listed in front of the others: the longest chains start first
#define SQ_XLONG_CAP 4096
#define SQ_LG 8                   // groups of 64 updates a long run requests per trip
#define SQ_LONG_CAP (1 << 20)     // voxels a batch may hand to the long role of k_seq_replay (beyond: they stay with their lane); the first SQ_XLONG_CAP entries are the longest

// exclusive prefix sums of a[0, C * SQ_NT) in LDS, in place; returns the total.  Every thread of the workgroup calls it with the data in
// place and visible (a barrier before); two barriers inside, the result is visible on return.
template <int C>
__device__ __forceinline__ uint32_t sq_scan_excl(uint32_t* a, uint32_t* s_w)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t v[C]; uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < C; ++q) { v[q] = a[tid * C + q]; sum += v[q]; }
    uint32_t inc = sum;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d); if (lane >= d) inc += o; }
    if (lane == 63) s_w[wid] = inc;
    __syncthreads();
    uint32_t base = inc - sum;
    for (int w = 0; w < wid; ++w) base += s_w[w];
    const uint32_t total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
#pragma unroll
    for (int q = 0; q < C; ++q) { a[tid * C + q] = base; base += v[q]; }
    __syncthreads();
    return total;
}

// the rays of all frames of a batch, keyed (frame | struct-for key of the sensor voxel, left by k_segments); unused entries sort last
__global__ void __launch_bounds__(256) k_seq_keys(BatchDev B, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals, int stride, int keybits)
{
    const int q = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (q >= B.n || i >= stride) return;
    const FrameDev& F = B.f[q];
    const bool on = i < F.counters[6];
    const size_t at = (size_t)q * stride + i;
    keys[at] = on ? (((unsigned long long)q << keybits) | reinterpret_cast<const unsigned long long*>(F.keys)[i]) : ~0ull;      // (frame 15 = all ones: behind every frame)
    vals[at] = on ? (((uint32_t)q << 24) | (uint32_t)i) : 0xffffffffu;
}
// position in the sorted array - rays of the frames before = the ray's rank in its frame's struct-for order (left in the set's `vals`)
__global__ void __launch_bounds__(256) k_seq_ranks(BatchDev B, const uint32_t* __restrict__ vals_sorted, int total)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const uint32_t v = vals_sorted[p];
    if (v == 0xffffffffu) return;
    const int q = (int)(v >> 24), ray = (int)(v & 0xffffffu);
    int base = 0;
#pragma unroll
    for (int k = 0; k < TSL_NB; ++k) if (k < q) base += B.f[k].counters[6];
    B.f[q].vals[ray] = (uint32_t)(p - base);
}

// work items of k_seq_group: one per (frame, brick) -- or, for a brick with more than SQ_SORTCAP segments (the few next to the sensor: every ray
// of the frame crosses them), one per CHUNK of its segments: the segments are first cut by ray rank into buckets of 256 rays (counting sort
// into the frame's spare segment array), consecutive buckets are grouped while they fit the LDS sort.  Every item gets a slot (run offsets) of
// its own; the brick's word holds first slot | items << 20, and the replay walks a voxel's runs slot after slot: chunks are in rank order, so
// that IS the replay order.  (One workgroup per heavy brick took ~0.9 ms for the brick around the sensor; its chunks now run side by side.)
#define SQ_SLOT_BITS 20
#define SQ_CHUNK_MAX 1024
#define SQ_HEAVY 1024             // k_seq_group claims the items with more segments than this first
__global__ void __launch_bounds__(SQ_NT) k_seq_split(MapDev M, BatchDev B, const SeqDev* __restrict__ SD)
{
    const int q = blockIdx.y;
    if (q >= B.n) return;
    const FrameDev& F = B.f[q];
    const SeqDev S = SD[q];
    __shared__ uint32_t s_bk[SQ_NBK_MAX];                        // segments per rank bucket -> bucket ends
    __shared__ uint32_t s_ch[SQ_CHUNK_MAX + 1];                  // chunk boundaries (segment index inside the brick)
    __shared__ uint32_t s_w[4];
    __shared__ int s_n[3];
    const int tid = threadIdx.x;
    const int nrays = F.counters[6];
    const bool failed = F.counters[HDR_FAIL] != 0 || nrays > (SQ_NBK_MAX << SQ_BSHIFT);
    if (nrays > (SQ_NBK_MAX << SQ_BSHIFT) && blockIdx.x == 0 && tid == 0) frame_fail(M, F, 4);
    const int nact = min(F.counters[1], F.max_frame_bricks);
    const uint32_t* __restrict__ rank_of_ray = F.vals;
    for (int bi = blockIdx.x; bi < nact; bi += gridDim.x) {
        const int b = F.act_b[bi];
        const int n = F.bnseg[b], off = F.boffset[b];
        if (tid == 0) { F.bhist[b] = 0; F.bcursor[b] = 0; }                            // the set's per-brick words are zero between frames
        if (failed || n <= 0) { if (tid == 0) F.bslab[b] = 0; continue; }               // (uniform) nothing of a frame that overflowed its scratch is integrated
        if (n <= SQ_CHUNKSEGS) {
            if (tid == 0) {
                const int slot = __hip_atomic_fetch_add(&F.counters[HDR_SEQ_SLOTS], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (slot >= S.slot_cap) { frame_fail(M, F, 4); F.bslab[b] = 0; }
                else { S.items[slot] = make_int4(off, n, slot, 0); F.bslab[b] = slot | (1 << SQ_SLOT_BITS); }
            }
            continue;
        }
        const unsigned long long* const segs = F.seg_sorted + off;
        unsigned long long* const temp = F.seg + off;                 // (k_scatter has consumed the raw segments; [off, off + n) belongs to this brick)
        const int nbk = (nrays + (1 << SQ_BSHIFT) - 1) >> SQ_BSHIFT;
        __syncthreads();
        for (int i = tid; i < SQ_NBK_MAX; i += SQ_NT) s_bk[i] = 0u;
        __syncthreads();
        for (int k = tid; k < n; k += SQ_NT) {
            const int ray = (int)((segs[k] >> (SEG_CNT_BITS + SEG_J_BITS)) & ((1u << STG_RAY_BITS) - 1));
            atomicAdd(&s_bk[rank_of_ray[ray] >> SQ_BSHIFT], 1u);
        }
        __syncthreads();
        (void)sq_scan_excl<SQ_NBK_MAX / SQ_NT>(s_bk, s_w);
        for (int k = tid; k < n; k += SQ_NT) {
            const unsigned long long sg = segs[k];
            const int ray = (int)((sg >> (SEG_CNT_BITS + SEG_J_BITS)) & ((1u << STG_RAY_BITS) - 1));
            temp[atomicAdd(&s_bk[rank_of_ray[ray] >> SQ_BSHIFT], 1u)] = sg;          // afterwards s_bk[k] = END of bucket k
        }
        __syncthreads();
        if (tid == 0) {      // consecutive buckets while they fit the sort buffer
            int nch = 0; uint32_t start = 0u, prev = 0u; bool bad = false;
            s_ch[0] = 0u;
            for (int k = 0; k < nbk && !bad; ++k) {
                const uint32_t e = s_bk[k];
                if (e - start > (uint32_t)SQ_CHUNKSEGS) {
                    // (a single bucket may exceed the chunk size -- it then is a chunk of its own -- but not the sort buffer)
                    if (prev != start) { if (nch + 1 >= SQ_CHUNK_MAX) { bad = true; break; } s_ch[++nch] = prev; start = prev; }
                    if (e - start > (uint32_t)SQ_SORTCAP) { bad = true; break; }      // one bucket beyond the sort buffer (more than 8 segments per ray and brick)
                }
                prev = e;
            }
            if (!bad && (uint32_t)n > start) { if (nch + 1 > SQ_CHUNK_MAX) bad = true; else s_ch[++nch] = (uint32_t)n; }
            int slot0 = -1;
            if (!bad) {
                slot0 = __hip_atomic_fetch_add(&F.counters[HDR_SEQ_SLOTS], nch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (slot0 + nch > S.slot_cap) bad = true;
            }
            if (bad) { frame_fail(M, F, 4); F.bslab[b] = 0; nch = 0; }
            else F.bslab[b] = slot0 | (nch << SQ_SLOT_BITS);
            s_n[0] = nch; s_n[1] = slot0;
        }
        __syncthreads();
        const int nch = s_n[0], slot0 = s_n[1];
        for (int c = tid; c < nch; c += SQ_NT) S.items[slot0 + c] = make_int4(off + (int)s_ch[c], (int)(s_ch[c + 1] - s_ch[c]), slot0 + c, 1);
        __syncthreads();
    }
}

// phase A of the literal mode, one work item (k_seq_split) per workgroup at a time: the item's segments are put in replay order -- (ray rank,
// first step) --, every step gets its replay position, and the steps are grouped by voxel, replay order kept inside a voxel (a stable counting
// sort), as the run offsets (CSR) of the item's slot say.  What each part costs was measured per stage with round 4's kernel (-DTSL_SEQ_TIMING) and per
// work item with this one (-DTSL_SEQ_TRACE, tools/seq_trace_probe.py; DESIGN.md section 4, "The second half of round 5"):
//   * persistent workgroups -- exactly the three per CU the device holds -- claim items through the frame's counter, heavy ones first;
//   * the sort is a sample sort with explicit barriers only (round 4's bitonic network: -DTSL_SEQ_BITONIC);
//   * the walks take the segments longest first, dealt out in alternating directions (a counting sort by length, as the default path's brick
//     kernel does): the 64 lanes of a wave walk segments of equal length.  Where a tuple goes is fixed by the replay order, not by who walks it;
//   * the placing pass has no workgroup barrier: wave w owns quarter w of the replay sequence, the first walk has counted every voxel's steps per quarter
//     (four 16-bit fields of one LDS word), so a wave's cursor for a voxel starts behind the earlier quarters' tuples and only that wave moves it.
template <bool TEX>
__global__ void __launch_bounds__(SQ_NT, 3) k_seq_group(MapDev M, BatchDev B, const SeqDev* __restrict__ SD, unsigned short* __restrict__ perm_all)
{
    // 52 KiB of LDS, three workgroups per CU (every stage waits for LDS or memory round trips: more resident waves is what hides them).  One 16 KiB
    // region holds, in turn: the sort keys; the sorted segments in 32 bits + their replay positions; the voxels' run offsets.
    __shared__ unsigned long long s_seg[SQ_SORTCAP];             // 16 KiB: the item's segments as sort keys: rank 22 | first step 12 | steps 6 | ray 22
    uint32_t* const s_sr = reinterpret_cast<uint32_t*>(s_seg);   //   after the sort, [0, 2048): ray 21 | steps 6 of the k-th segment in replay order
    uint32_t* const s_pre = s_sr + SQ_SORTCAP;                   //   after the sort, [2048, 4096): replay position of its first step 17 | first step 12
#ifdef TSL_SEQ_STASH
    uint32_t* const s_hist = s_sr;                               //   after the walk, all of it: tuples per voxel -> run offsets (the first form; now they go straight to the slot's table)
#endif
    __shared__ unsigned long long s_pack[TSL_BRK3];              // 32 KiB: per voxel, its tuples in each quarter of the replay sequence (16 bits each) -> the quarters' cursors
    // (the item's segments by length, longest first -- 4 KiB of 16-bit indices -- live in GLOBAL memory, a row per workgroup of the launch: with them in LDS
    //  the kernel needed 53.8 KB and only TWO workgroups fitted a CU (the trace of round 5 showed exactly 512 items alive at the start of a launch of 768);
    //  the walk reads an entry per segment, next to the segment's 16-byte ray record)
    unsigned short* const g_perm = perm_all + (size_t)blockIdx.x * SQ_SORTCAP;
    __shared__ int s_bin[64];
    __shared__ uint32_t s_w[4];
    __shared__ unsigned long long s_rb;
    __shared__ int s_claim;
    __shared__ uint32_t s_rw[64];                                // (row, wave) totals of the run-offset scan
    __shared__ uint32_t s_flag[4][64];                           // per wave: start offsets of the next segments inside the group of 64 positions being placed (zero between groups)
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    s_flag[wid][lane] = 0u;
    // PERSISTENT workgroups (round 5): the launch is exactly the workgroups the CUs hold (three per CU), and each claims items through the frame's counter
    // until the batch has none left -- first the heavy ones (a chunk of a brick next to the sensor: ~2 000 segments, 35 k tuples, 200 us), then the light
    // ones (median 450 segments, 38 us), so the launch does not end on a late heavy item.  The first form launched one workgroup per possible item (1 024 per
    // frame, 44 % of them empty): a traced batch of eight frames kept 346 of the 768 workgroup slots busy on average and took 667 us for 231 ms of item time
    // (tools/seq_trace_probe.py) -- slots waited for the dispatcher, not for work.
    for (int dq = 0; dq < B.n; ++dq) {
    const int q = ((int)blockIdx.x + dq) % B.n;          // a workgroup starts on "its" frame and moves on when that frame has no items left
    const FrameDev& F = B.f[q];
    const FrameParams& P = *B.p[q];
    const SeqDev S = SD[q];
    // ONE decision for the whole workgroup: another workgroup of this launch may set the frame's fail word at its tuple-cap check while this one
    // reads it, and four waves that disagree would meet different barriers of the claim loop below.  Thread 0 reads, everybody takes its answer.
    if (tid == 0) { s_claim = F.counters[HDR_FAIL] != 0 ? -1 : min(F.counters[HDR_SEQ_SLOTS], S.slot_cap); }
    __syncthreads();
    const int nitems = uni_i(s_claim);
    __syncthreads();                                     // (the word is written again by the first claim)
    if (nitems < 0) continue;
    const uint32_t* __restrict__ rank_of_ray = F.vals;
#define SQ_TICK(k)         // (stage marks of round 4's -DTSL_SEQ_TIMING build, which did not survive the persistent workgroups and was removed in round 6; per-item times: -DTSL_SEQ_TRACE)
    for (;;) {
        if (tid == 0) s_claim = __hip_atomic_fetch_add(&F.counters[HDR_SEQ_CLAIM], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int claimed = uni_i(s_claim);
        __syncthreads();                                 // (the word is written again by the next claim)
        if (claimed >= 2 * nitems) break;
        const bool heavy_pass = claimed < nitems;        // claims [0, nitems): the heavy items; [nitems, 2 nitems): the others
        const int it = heavy_pass ? claimed : claimed - nitems;
#ifdef TSL_SEQ_TRACE       // developer build: per work item { start, end (100 MHz clock), segments, tuples | CU id << 32 } at dbg[1024 + 4 * (frame * 1024 + item)], items < 1024 per frame
        const long long _tr0 = wall_clock64();
#endif
        const int4 item_v = S.items[it];
        const int4 item = make_int4(uni_i(item_v.x), uni_i(item_v.y), uni_i(item_v.z), uni_i(item_v.w));      // (the same for every lane: scalar loop control below)
        if ((item.y > SQ_HEAVY) != heavy_pass) continue;
        const int m = item.y;
        const unsigned long long* const src = (item.w ? F.seg : F.seg_sorted) + item.x;
        uint32_t* const csr = S.csr + (size_t)item.z * SQ_CSR_STRIDE;
        int P2 = 8; while (P2 < m) P2 <<= 1;
        // ---- the segments, keyed; the counters cleared under the two memory round trips ----
        unsigned long long sg[SQ_SORTCAP / SQ_NT];
#pragma unroll
        for (int r = 0; r < SQ_SORTCAP / SQ_NT; ++r) { const int k = r * SQ_NT + tid; sg[r] = k < m ? src[k] : ~0ull; }
        uint32_t rk[SQ_SORTCAP / SQ_NT];
#pragma unroll
        for (int r = 0; r < SQ_SORTCAP / SQ_NT; ++r)
            rk[r] = sg[r] != ~0ull ? rank_of_ray[(sg[r] >> (SEG_CNT_BITS + SEG_J_BITS)) & ((1u << STG_RAY_BITS) - 1)] : 0u;
        for (int i = tid; i < TSL_BRK3; i += SQ_NT) s_pack[i] = 0ull;
        if (tid < 64) s_bin[tid] = 0;
#ifndef TSL_SEQ_BITONIC
        // ---- the segments in replay order: a SAMPLE SORT (round 5).  The bitonic network that stood here took 27 % of the kernel -- ~55 dependent LDS
        //      round trips per item at three waves per SIMD (and needed an explicit wait in front of its barriers, see the #else branch).  Now: 64 of the
        //      keys, sorted by one wave in registers (21 shuffle stages), cut the key space into 64 buckets; a key finds its bucket by a binary search in
        //      the 63 splitters and its place in the bucket with one LDS add, the buckets are laid out by a 64-entry scan, and a key's final position is
        //      its bucket's start + the number of smaller keys in the bucket (~m / 64 of them): seven barriers, ~12 dependent round trips.  Any splitters
        //      give the sorted order (the bucket of a key is monotone in the key, keys are distinct); bad ones only make a bucket long.
        unsigned long long key[SQ_SORTCAP / SQ_NT];
#pragma unroll
        for (int r = 0; r < SQ_SORTCAP / SQ_NT; ++r) {
            const int k = r * SQ_NT + tid;
            key[r] = ~0ull;
            if (k < m) {
                const unsigned long long ray = (sg[r] >> (SEG_CNT_BITS + SEG_J_BITS)) & ((1u << STG_RAY_BITS) - 1);
                key[r] = ((unsigned long long)rk[r] << 40) | (((sg[r] >> SEG_CNT_BITS) & 0xfffull) << 28) | ((sg[r] & 63ull) << 22) | ray;
                s_seg[k] = key[r];
            }
        }
        SQ_TICK(0)
        __syncthreads();
        unsigned long long* const s_spl = s_seg;                   // (the unsorted keys are in registers; the first 64 words are free once the samples are read)
        if (wid == 0) {
            unsigned long long v = s_seg[((uint32_t)lane * (uint32_t)m) >> 6];
            for (int kk = 2; kk <= 64; kk <<= 1)
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), j) << 32) | (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)v, j);
                    const bool keep_min = ((lane & j) == 0) == ((lane & kk) == 0);
                    v = keep_min ? (o < v ? o : v) : (o > v ? o : v);
                }
            s_spl[lane] = v;                                       // ascending; s_spl[1..63] are the splitters
        }
        __syncthreads();
        int bkt[SQ_SORTCAP / SQ_NT], bpos[SQ_SORTCAP / SQ_NT];
#pragma unroll
        for (int r = 0; r < SQ_SORTCAP / SQ_NT; ++r) {
            bkt[r] = 0; bpos[r] = 0;
            if (r * SQ_NT + tid < m) {
                int lo = 0;                                        // splitters <= key
#pragma unroll
                for (int step = 32; step; step >>= 1) { const int idx = lo + step; if (idx <= 63 && s_spl[idx] <= key[r]) lo = idx; }
                bkt[r] = lo; bpos[r] = atomicAdd(&s_bin[lo], 1);
            }
        }
        __syncthreads();
        if (tid < 64) {
            const int cb = s_bin[tid]; int inc = cb;
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (tid >= d) inc += o; }
            s_bin[tid] = inc - cb;                                 // first place of the bucket
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SQ_SORTCAP / SQ_NT; ++r) if (r * SQ_NT + tid < m) s_seg[s_bin[bkt[r]] + bpos[r]] = key[r];
        __syncthreads();
        int fin[SQ_SORTCAP / SQ_NT];
#pragma unroll
        for (int r = 0; r < SQ_SORTCAP / SQ_NT; ++r) {
            fin[r] = 0;
            if (r * SQ_NT + tid < m) {
                const int o = s_bin[bkt[r]], e = bkt[r] == 63 ? m : s_bin[bkt[r] + 1];
                int c = 0;
                for (int i = o; i < e; ++i) c += s_seg[i] < key[r] ? 1 : 0;
                fin[r] = o + c;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SQ_SORTCAP / SQ_NT; ++r) if (r * SQ_NT + tid < m) s_seg[fin[r]] = key[r];
        if (tid < 64) s_bin[tid] = 0;                              // the length bins below start from zero
#else   // the first form (developer A/B: -DTSL_SEQ_BITONIC)
#pragma unroll
        for (int r = 0; r < SQ_SORTCAP / SQ_NT; ++r) {
            const int k = r * SQ_NT + tid;
            if (k < P2) {
                unsigned long long key = ~0ull;
                if (k < m) {
                    const unsigned long long ray = (sg[r] >> (SEG_CNT_BITS + SEG_J_BITS)) & ((1u << STG_RAY_BITS) - 1);
                    key = ((unsigned long long)rk[r] << 40) | (((sg[r] >> SEG_CNT_BITS) & 0xfffull) << 28) | ((sg[r] & 63ull) << 22) | ray;
                }
                s_seg[k] = key;
            }
        }
        SQ_TICK(0)
        __syncthreads();
        // ---- bitonic network over P2 keys.  Wave w owns keys [w * P2 / 4, (w + 1) * P2 / 4) and the P2 / 8 compare-exchanges inside them: a stage
        //      with distance j <= P2 / 8 never leaves a wave's keys and needs no workgroup barrier ----
        {
            const int per = P2 >> 3;                             // compare-exchanges of a wave per stage
            for (int kk = 2; kk <= P2; kk <<= 1)
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    const bool local = j <= per;
                    // (the explicit wait: behind a barrier-free stage the compiler's scoreboard takes the wave's LDS writes for done -- it emitted a bare
                    //  s_barrier here, and a wave of the next stage read keys whose exchange was still in the LDS queue of another SIMD: a segment of
                    //  the item lost and another one walked twice, one (frame, brick) in a few hundred batches -- round 5, tools/repro_r5.py benchlike,
                    //  caught by TSL_SEQ_VERIFY's brute-force check)
#ifdef TSL_SEQ_BITONIC_NOWAIT      // (round 4's form, kept so that tests/test_barrier_scan_cpu.py can show the scanner the bare s_barrier)
                    if (!local) __syncthreads();
#else
                    if (!local) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); }
#endif
                    for (int u = lane; u < per; u += 64) {
                        const int t = wid * per + u;
                        const int i = 2 * t - (t & (j - 1)), ix = i + j;
                        const unsigned long long a = s_seg[i], c = s_seg[ix];
                        const bool up = (i & kk) == 0;
                        if ((a > c) == up) { s_seg[i] = c; s_seg[ix] = a; }
                    }
                    if (!local) __syncthreads();
                    else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
                }
        }
#endif
        SQ_TICK(1)
        __syncthreads();
        // ---- replay positions (prefix of the step counts, sorted order); the segments by length ----
        int lrank[SQ_SORTCAP / SQ_NT];
        uint32_t cj[SQ_SORTCAP / SQ_NT], sr[SQ_SORTCAP / SQ_NT];     // steps | first step << 8; ray | steps << 21
#pragma unroll
        for (int r = 0; r < SQ_SORTCAP / SQ_NT; ++r) {
            const int k = r * SQ_NT + tid;
            const unsigned long long key = k < m ? s_seg[k] : 0ull;
            const uint32_t cnt = (uint32_t)((key >> 22) & 63ull);
            cj[r] = cnt | ((uint32_t)((key >> 28) & 0xfffull) << 8);
            sr[r] = (uint32_t)(key & 0x1fffffull) | (cnt << 21);                  // (a ray index has 21 bits: max_points <= 2^21 in this mode)
            lrank[r] = k < m ? atomicAdd(&s_bin[63 - (int)cnt], 1) : -1;
        }
        __syncthreads();                                                          // every key is in registers: the region changes hands
#pragma unroll
        for (int r = 0; r < SQ_SORTCAP / SQ_NT; ++r) { const int k = r * SQ_NT + tid; s_sr[k] = sr[r]; s_pre[k] = cj[r] & 63u; }
        __syncthreads();
        if (tid < 64) {
            const int cb = s_bin[tid]; int inc = cb;
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (tid >= d) inc += o; }
            s_bin[tid] = inc - cb;
        }
        const uint32_t T = (uint32_t)uni_i((int)sq_scan_excl<SQ_SORTCAP / SQ_NT>(s_pre, s_w));      // the item's steps: its part of the frame's tuple arrays (two barriers inside)
#pragma unroll
        for (int r = 0; r < SQ_SORTCAP / SQ_NT; ++r) {
            const int k = r * SQ_NT + tid;
            if (lrank[r] >= 0) g_perm[s_bin[63 - (int)(cj[r] & 63u)] + lrank[r]] = (unsigned short)k;
            s_pre[k] |= (cj[r] >> 8) << 17;                                       // (the item's steps number fewer than 2^17)
        }
        if (tid == 0) s_rb = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(&F.counters[HDR_SEQ_TUPLES]), (unsigned long long)T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (g_perm: global stores of one wave read by another behind the barrier -- __syncthreads() waits for the LDS counter only)
        __syncthreads();
        const unsigned long long rb = s_rb;
        if ((long long)(rb + T) > S.cap) { if (tid == 0) frame_fail(M, F, 4); __syncthreads(); continue; }
#ifndef TSL_SEQ_STASH
        // ---- Round 5: NO STASH.  The first form wrote every step as an 8-byte tuple at its replay position (the "stash") and read it back in the counting
        //      sort: 16 bytes per step through memory on top of the 8 of the replay tuple, and the mode is bound by exactly that -- one more 8-byte store
        //      per step costs it 12.6 % (profiles/r05_literal_experiments.txt).  Now the first walk only COUNTS (voxel of every step, per quarter of the replay
        //      sequence: no distance, no store), and the placing pass -- wave w over quarter w, 64 consecutive replay positions at a time -- finds every
        //      position's (segment, step) and evaluates the step THERE: the arithmetic of a step twice, its bytes once.
        const uint32_t Q = (((T + 3u) >> 2) + 63u) & ~63u;       // a quarter of the replay sequence, in whole groups of 64 tuples
        SQ_TICK(2)
        // ---- walk 1: every step's voxel, counted per voxel and quarter ----
        for (int r = 0; r * SQ_NT < m; ++r) {
            const int idx = r * SQ_NT + ((r & 1) ? SQ_NT - 1 - tid : tid);
            if (idx >= m) continue;
            const int k = (int)g_perm[idx];
            const uint32_t sk = s_sr[k], pj = s_pre[k];
            const int ray = (int)(sk & 0x1fffffu), cnt = (int)(sk >> 21), j0 = (int)(pj >> 17);
            const uint4 rec = F.rayA[ray];
            const float d0 = h2f((h16)(rec.y >> 16)), d1 = h2f((h16)(rec.z & 0xffffu)), d2 = h2f((h16)(rec.z >> 16));
            const uint32_t at = pj & 0x1ffffu;
            for (int s = 0; s < cnt; ++s) {
                const float jf = (float)(j0 + s);
                const float x0 = (d0 * jf) * P.vs + P.T[0], x1 = (d1 * jf) * P.vs + P.T[1], x2 = (d2 * jf) * P.vs + P.T[2];     // :253
                const int i0 = rnd_i(div_vs(x0, P.vs, P.rvs, P.fastdiv)), i1 = rnd_i(div_vs(x1, P.vs, P.rvs, P.fastdiv)), i2 = rnd_i(div_vs(x2, P.vs, P.rvs, P.fastdiv));   // :254
                const int l = (((i0 + M.hN) & 15) << 8) | (((i1 + M.hN) & 15) << 4) | ((i2 + M.hNz) & 15);            // the segment lies inside this brick
                const uint32_t pos = at + (uint32_t)s;
                const uint32_t qtr = (pos >= Q ? 1u : 0u) + (pos >= 2u * Q ? 1u : 0u) + (pos >= 3u * Q ? 1u : 0u);
                atomicAdd(&s_pack[l], 1ull << (16u * qtr));
            }
        }
        SQ_TICK(3)
        __syncthreads();
        // ---- run offsets of the brick's voxels for this item, straight into the slot's offset table in global memory (the LDS region that used to hold them
        //      keeps the segments for the placing pass); a voxel's four counts become the quarters' first positions inside its run.  Thread t owns voxels
        //      q * 256 + t: sixteen row scans in registers (wave shuffles), the 64 (row, wave) totals scanned by one wave ----
        {
            uint32_t tot[TSL_BRK3 / SQ_NT], inc[TSL_BRK3 / SQ_NT];
#pragma unroll
            for (int qq = 0; qq < TSL_BRK3 / SQ_NT; ++qq) {
                const int i = qq * SQ_NT + tid;
                const unsigned long long v = s_pack[i];
                const uint32_t c0 = (uint32_t)(v & 0xffffull), c1 = (uint32_t)((v >> 16) & 0xffffull), c2 = (uint32_t)((v >> 32) & 0xffffull), c3 = (uint32_t)(v >> 48);
                tot[qq] = c0 + c1 + c2 + c3;
                s_pack[i] = ((unsigned long long)c0 << 16) | ((unsigned long long)(c0 + c1) << 32) | ((unsigned long long)(c0 + c1 + c2) << 48);
                uint32_t x = tot[qq];
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)x, d); if (lane >= d) x += o; }
                inc[qq] = x;
                if (lane == 63) s_rw[qq * 4 + wid] = x;            // total of (row qq, wave wid)
            }
            __syncthreads();
            if (tid < 64) {
                const uint32_t cb = s_rw[tid]; uint32_t x = cb;
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)x, d); if (tid >= d) x += o; }
                s_rw[tid] = x - cb;                                // tuples in front of (row, wave)
            }
            __syncthreads();
#pragma unroll
            for (int qq = 0; qq < TSL_BRK3 / SQ_NT; ++qq) csr[qq * SQ_NT + tid] = s_rw[qq * 4 + wid] + inc[qq] - tot[qq];
            if (tid == 0) { csr[TSL_BRK3] = T; csr[SQ_CSR_BASE] = (uint32_t)rb; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the offsets are global stores of one wave read by another behind the barrier: it only drains the LDS counter)
        __syncthreads();
        SQ_TICK(4)
        // ---- placing pass = stable counting sort by voxel: wave w takes quarter w of the replay sequence, 64 consecutive positions at a time.  The segment a
        //      position belongs to: the wave carries the last segment that starts at or before the group (kcur); the next 64 segments flag their start
        //      offsets in a wave-private LDS row, one ballot over the row gives every lane the number of starts up to its position.  Tuples of one voxel
        //      inside a group find each other with twelve ballots; the voxel's cursor for this quarter is moved by the first of them ----
        float2* const tup = S.tup + rb;
        bool unsafe = false;
        {
            const uint32_t e = min(T, (uint32_t)(wid + 1) * Q);
            uint32_t g0 = (uint32_t)wid * Q;
            int kcur = 0;
            if (g0 < e) {                                         // (uniform) the last segment whose first position is <= g0
                int lo = 0, hi = m - 1;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((s_pre[mid] & 0x1ffffu) <= g0) lo = mid; else hi = mid - 1; }
                kcur = uni_i(lo);
            }
            uint32_t* const flag = s_flag[wid];
            // the (segment, step) of a group's positions and the rays' records are requested ONE GROUP AHEAD (the record is a global load at the end of a
            // dependent LDS chain: unhidden it cost the pass more than the stash had)
            uint32_t pj_n = 0u, ray_n = 0u; uint4 rec_n = make_uint4(0u, 0u, 0u, 0u); bool valid_n = false;
            auto locate = [&](uint32_t g) {
                const uint32_t t = g + (uint32_t)lane;
                valid_n = t < e;
                const int ki = kcur + 1 + lane;
                const uint32_t ri = ki < m ? (s_pre[ki] & 0x1ffffu) - g : 0xffffffffu;      // offset of the start of segment ki inside the group (starts increase strictly)
                if (ri >= 1u && ri <= 63u) flag[ri] = 1u;
                const unsigned long long adv = __ballot(ri >= 1u && ri <= 64u);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();      // (wave-private row: LDS operations of a wave complete in order)
                const uint32_t fl = flag[lane];
                const unsigned long long starts = __ballot(fl != 0u);
                if (fl) flag[lane] = 0u;                                                       // the row is zero again for the next group
                const int k = kcur + popc64(starts & ((2ull << lane) - 1ull));                 // starts at offsets 1 .. lane (offset 0 is never flagged)
                kcur += popc64(adv);
                if (valid_n) { const uint32_t sk = s_sr[k]; pj_n = s_pre[k]; ray_n = sk & 0x1fffffu; rec_n = F.rayA[ray_n]; }
            };
            if (g0 < e) locate(g0);
            for (; g0 < e; g0 += 64u) {
                const uint32_t t = g0 + (uint32_t)lane;
                const bool valid = valid_n;
                const uint32_t pj = pj_n, ray = ray_n; const uint4 rec = rec_n;
                if (g0 + 64u < e) locate(g0 + 64u);
                __builtin_amdgcn_sched_barrier(0);
                int l = 0; float sd = 0.0f, w = 0.0f; uint32_t off = 0u;
                if (valid) {
                    const float pf0 = h2f((h16)(rec.x & 0xffffu)), pf1 = h2f((h16)(rec.x >> 16)), pf2 = h2f((h16)(rec.y & 0xffffu));
                    const float d0 = h2f((h16)(rec.y >> 16)), d1 = h2f((h16)(rec.z & 0xffffu)), d2 = h2f((h16)(rec.z >> 16));
                    const float P0 = pf0 + P.T[0], P1 = pf1 + P.T[1], P2f = pf2 + P.T[2];                                 // :246
                    const float jf = (float)((int)(pj >> 17) + (int)(t - (pj & 0x1ffffu)));
                    const float x0 = (d0 * jf) * P.vs + P.T[0], x1 = (d1 * jf) * P.vs + P.T[1], x2 = (d2 * jf) * P.vs + P.T[2];     // :253
                    const int i0 = rnd_i(div_vs(x0, P.vs, P.rvs, P.fastdiv)), i1 = rnd_i(div_vs(x1, P.vs, P.rvs, P.fastdiv)), i2 = rnd_i(div_vs(x2, P.vs, P.rvs, P.fastdiv));   // :254
                    l = (((i0 + M.hN) & 15) << 8) | (((i1 + M.hN) & 15) << 4) | ((i2 + M.hNz) & 15);
                    off = csr[l];                                                                                          // (requested here: the distance and the ballots run under it)
                    const float v0 = P0 - x0, v1 = P1 - x1, v2 = P2f - x2;                                                 // :258
                    const float s2 = (v0 * v0 + v1 * v1) + v2 * v2;
                    const float dist = s2 >= 1.2621774483536189e-29f ? sqrt_rn_norm(s2) : sqrt_rn(s2);                      // :259  (2^-96: below it sqrtf rescales)
                    const float dot = (v0 * pf0 + v1 * pf1) + v2 * pf2;
                    sd = dist * (float)sgn_f(dot);                                                                         // :260
                    w = seq_w_of(seq_w_code(__uint_as_float(rec.w)));                                                       // (the weight as the 16-bit code of the first form carried it)
                }
                const unsigned long long vm = __ballot(valid);
                uint32_t mlo = (uint32_t)vm, mhi = (uint32_t)(vm >> 32);
#pragma unroll
                for (int bit = 0; bit < 12; ++bit) {                              // lanes whose bit equals mine stay: m &= ~(ballot ^ (mine ? ~0 : 0))
                    const uint32_t mine = (uint32_t)((int)((uint32_t)l << (31 - bit)) >> 31);
                    const unsigned long long bm = __ballot(mine != 0u);
                    mlo &= ~((uint32_t)bm ^ mine); mhi &= ~((uint32_t)(bm >> 32) ^ mine);
                }
                const unsigned long long mk = ((unsigned long long)mhi << 32) | mlo;
                const int my = rank_below(mk), gs = popc64(mk);                   // tuples of my voxel before me in this group / in this group
                if (valid) {
                    const uint32_t cur = (uint32_t)(s_pack[l] >> (16 * wid)) & 0xffffu;
                    if (my == 0) atomicAdd(&s_pack[l], (unsigned long long)gs << (16 * wid));      // (behind the read: LDS operations of a wave complete in order)
                    const float c = w * sd;                                       // the replay tuple: w and c = w * sd (:264)
                    const uint32_t pos = off + cur + (uint32_t)my;
                    tup[pos] = make_float2(w, c);
                    if (TEX) S.tup_ray[rb + pos] = ray;
                    unsafe = unsafe || !(fabsf(sd) <= 60.0f) || (c != 0.0f && fabsf(c) < 8.67e-19f);      // 2^-60: the residuals of the division-free quotient stay representable
                }
            }
        }
#else   // the first form: every step through the stash (developer A/B: -DTSL_SEQ_STASH; the stash arrays are then allocated)
        unsigned long long* const stash = S.stash + rb;
        const uint32_t Q = (((T + 3u) >> 2) + 63u) & ~63u;       // a quarter of the replay sequence, in whole groups of 64 tuples
        SQ_TICK(2)
        // ---- walk: every step a tuple at its replay position, counted per voxel and quarter ----
        for (int r = 0; r * SQ_NT < m; ++r) {
            const int idx = r * SQ_NT + ((r & 1) ? SQ_NT - 1 - tid : tid);
            if (idx >= m) continue;
            const int k = (int)g_perm[idx];
            const uint32_t sk = s_sr[k], pj = s_pre[k];
            const int ray = (int)(sk & 0x1fffffu), cnt = (int)(sk >> 21), j0 = (int)(pj >> 17);
            const uint4 rec = F.rayA[ray];
            const float pf0 = h2f((h16)(rec.x & 0xffffu)), pf1 = h2f((h16)(rec.x >> 16)), pf2 = h2f((h16)(rec.y & 0xffffu));
            const float d0 = h2f((h16)(rec.y >> 16)), d1 = h2f((h16)(rec.z & 0xffffu)), d2 = h2f((h16)(rec.z >> 16));
            const unsigned long long zz = (unsigned long long)seq_w_code(__uint_as_float(rec.w)) << SQ_TUP_Z_SHIFT;
            const float P0 = pf0 + P.T[0], P1 = pf1 + P.T[1], P2f = pf2 + P.T[2];                                     // :246
            const uint32_t at = pj & 0x1ffffu;
            for (int s = 0; s < cnt; ++s) {          // (starting the lanes at different steps, as the default path's walk does, changed nothing here: 6 110 against 6 080 frames/s)
                const float jf = (float)(j0 + s);
                const float x0 = (d0 * jf) * P.vs + P.T[0], x1 = (d1 * jf) * P.vs + P.T[1], x2 = (d2 * jf) * P.vs + P.T[2];     // :253
                const int i0 = rnd_i(div_vs(x0, P.vs, P.rvs, P.fastdiv)), i1 = rnd_i(div_vs(x1, P.vs, P.rvs, P.fastdiv)), i2 = rnd_i(div_vs(x2, P.vs, P.rvs, P.fastdiv));   // :254
                const int l = (((i0 + M.hN) & 15) << 8) | (((i1 + M.hN) & 15) << 4) | ((i2 + M.hNz) & 15);            // the segment lies inside this brick
                const float v0 = P0 - x0, v1 = P1 - x1, v2 = P2f - x2;                                                 // :258
                const float s2 = (v0 * v0 + v1 * v1) + v2 * v2;
                const float dist = s2 >= 1.2621774483536189e-29f ? sqrt_rn_norm(s2) : sqrt_rn(s2);                      // :259  (2^-96: below it sqrtf rescales)
                const float dot = (v0 * pf0 + v1 * pf1) + v2 * pf2;
                const float sd = dist * (float)sgn_f(dot);                                                              // :260
                const uint32_t pos = at + (uint32_t)s;
                stash[pos] = zz | ((unsigned long long)l << SQ_TUP_L_SHIFT) | (unsigned long long)__float_as_uint(sd);
#ifdef TSL_EXP_EXTRA_STORE      // developer experiment: 8 more bytes written per step (into the place the counting sort overwrites later): how sensitive is the mode to its memory traffic?
                reinterpret_cast<unsigned long long*>(S.tup + rb)[pos] = zz | (unsigned long long)__float_as_uint(sd);
#endif
                if (TEX) S.stash_ray[rb + pos] = (uint32_t)ray;
                const uint32_t qtr = (pos >= Q ? 1u : 0u) + (pos >= 2u * Q ? 1u : 0u) + (pos >= 3u * Q ? 1u : 0u);
                atomicAdd(&s_pack[l], 1ull << (16u * qtr));
            }
        }
        SQ_TICK(3)
        // The stash is GLOBAL memory written by one wave and read back by another behind the barrier below.  __syncthreads() only drains the LDS
        // counter (the compiler's workgroup-scope release waits for lgkmcnt(0) alone: it counts on the CU's L1 to keep its waves' accesses in order),
        // and on gfx950 a load of another wave did overtake a store still in flight: a few 128-byte lines of an item came back with the tuples the
        // working set held three batches earlier -- one (frame, brick) in a few hundred batches with some tuples in the wrong voxel's run (round 5;
        // tools/repro_r5.py benchlike).  Every storing wave therefore waits for its stores before it signals the barrier.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- run offsets of the brick's voxels for this item; a voxel's four counts become the quarters' first positions inside its run ----
        for (int i = tid; i < TSL_BRK3; i += SQ_NT) {
            const unsigned long long v = s_pack[i];
            const uint32_t c0 = (uint32_t)(v & 0xffffull), c1 = (uint32_t)((v >> 16) & 0xffffull), c2 = (uint32_t)((v >> 32) & 0xffffull), c3 = (uint32_t)(v >> 48);
            s_hist[i] = c0 + c1 + c2 + c3;
            s_pack[i] = ((unsigned long long)c0 << 16) | ((unsigned long long)(c0 + c1) << 32) | ((unsigned long long)(c0 + c1 + c2) << 48);
        }
        __syncthreads();
        (void)sq_scan_excl<TSL_BRK3 / SQ_NT>(s_hist, s_w);
        for (int i = tid; i < TSL_BRK3; i += SQ_NT) csr[i] = s_hist[i];
        if (tid == 0) { csr[TSL_BRK3] = T; csr[SQ_CSR_BASE] = (uint32_t)rb; }
        SQ_TICK(4)
        // ---- stable counting sort by voxel: wave w takes quarter w of the sequence, 64 tuples at a time, in replay order.  Tuples of one voxel
        //      inside a group of 64 find each other with twelve ballots; the voxel's cursor for this quarter is moved by the first of them ----
        float2* const tup = S.tup + rb;
        bool unsafe = false;
        {
            const uint32_t e = min(T, (uint32_t)(wid + 1) * Q);
            uint32_t t = (uint32_t)wid * Q + (uint32_t)lane;
            unsigned long long xn = t < e ? stash[t] : 0ull;
            for (uint32_t g0 = (uint32_t)wid * Q; g0 < e; g0 += 64u, t += 64u) {
                const unsigned long long x = xn;
                const bool valid = t < e;
                xn = t + 64u < e ? stash[t + 64u] : 0ull;                       // the next group's tuples are under way while this one is placed
                __builtin_amdgcn_sched_barrier(0);
                const int l = (int)((x >> SQ_TUP_L_SHIFT) & 4095ull);
                const unsigned long long vm = __ballot(valid);
                uint32_t mlo = (uint32_t)vm, mhi = (uint32_t)(vm >> 32);
#pragma unroll
                for (int bit = 0; bit < 12; ++bit) {                              // lanes whose bit equals mine stay: m &= ~(ballot ^ (mine ? ~0 : 0))
                    const uint32_t mine = (uint32_t)((int)((uint32_t)l << (31 - bit)) >> 31);
                    const unsigned long long bm = __ballot(mine != 0u);
                    mlo &= ~((uint32_t)bm ^ mine); mhi &= ~((uint32_t)(bm >> 32) ^ mine);
                }
                const unsigned long long mk = ((unsigned long long)mhi << 32) | mlo;
                const int my = rank_below(mk), gs = popc64(mk);                   // tuples of my voxel before me in this group / in this group
                if (valid) {
                    const uint32_t cur = (uint32_t)(s_pack[l] >> (16 * wid)) & 0xffffu;
                    if (my == 0) atomicAdd(&s_pack[l], (unsigned long long)gs << (16 * wid));      // (behind the read: LDS operations of a wave complete in order)
                    // the replay tuple: what an update needs that does not depend on the voxel -- w and c = w * sd (:264)
                    const float sd = __uint_as_float((uint32_t)x), w = seq_w_of((h16)(x >> SQ_TUP_Z_SHIFT)), c = w * sd;
                    const uint32_t pos = s_hist[l] + cur + (uint32_t)my;
                    tup[pos] = make_float2(w, c);
                    if (TEX) S.tup_ray[rb + pos] = S.stash_ray[rb + t];
                    unsafe = unsafe || !(fabsf(sd) <= 60.0f) || (c != 0.0f && fabsf(c) < 8.67e-19f);      // 2^-60: the residuals of the division-free quotient stay representable
                }
            }
        }
#endif
        SQ_TICK(5)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // (the placing pass orders its wave-private LDS row with a wavefront-scope fence: behind one the compiler
                                                                                    //  emits bare barriers -- the cursor adds must have landed before the next item clears the counters)
        const int any_unsafe = __syncthreads_or(unsafe ? 1 : 0);                  // (also: every wave is done with the LDS arrays before the next item clears them)
        if (tid == 0) csr[SQ_CSR_UNSAFE] = any_unsafe ? 1u : 0u;
#ifdef TSL_SEQ_TRACE
        if (tid == 0 && it < 1024) {
            uint32_t hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            long long* rec = F.dbg + 1024 + 4 * ((size_t)q * 1024 + it);
            rec[0] = _tr0; rec[1] = wall_clock64(); rec[2] = m; rec[3] = (long long)T | ((long long)hwid << 32);
        }
#endif
    }
    }      // frames
}

// one run of one voxel, applied by its lane.  The division-free update (seq_update_fast) needs D = W + w and its correctly rounded reciprocal:
// at Wmax both come with the tuple; below it -- and a weight can sit below Wmax for good: RN16(W + w) = W as soon as w is under half an f16 ulp
// of W, e.g. w < 0.25 from W = 512 on -- the reciprocal is one IEEE division per update.  Outside the range the form is proven for (the
// item's flag, |T| > 60) the literal expression.
__device__ __forceinline__ void seq_walk_run(const float2* __restrict__ tp, uint32_t t, uint32_t end, bool can_fast, h16& T0, h16& W0)
{
    if (!(can_fast && fabsf(h2f(T0)) <= 60.0f)) {
        for (; t < end; ++t) {
            const float2 x = tp[t];
            const h16 Tn = f2h((h2f(hmul(T0, W0)) + x.y) / (h2f(W0) + x.x));                                              // dense_tsdf.py:264
            float wn = h2f(W0) + x.x; if (TSL_WMAX < wn) wn = TSL_WMAX;                                                   // :267
            T0 = Tn; W0 = f2h(wn);
        }
        return;
    }
    uint32_t Tr = T0; h16 Wb = W0;
    for (; t < end; ++t) {
        const float2 x = tp[t];
        const float D = h2f(Wb) + x.x;                                                                                    // (= the tuple's Wmax + w at Wmax)
        Tr = seq_update_fast(Tr, Wb, x.y, D, 1.0f / D);                                                                   // :264  (a convex combination: |T| <= 60 stays)
        Wb = f2h(D > TSL_WMAX ? TSL_WMAX : D);                                                                            // :267
    }
    T0 = (h16)Tr; W0 = Wb;
}

// phase B: one thread per voxel of every brick the batch integrates into (k_plan's unit tables with every brick a unit: brick id, pool index,
// frames of the batch with segments in it); sixteen 256-voxel slices per brick.  A voxel whose longest frame of the batch has fewer than SQ_LONG
// updates is replayed by its lane, frame after frame, slot after slot of the (frame, brick); the others -- a few thousand voxels around the
// sensor, among them the one every ray of a frame passes through -- are only LISTED here and replayed by the long role of k_seq_replay, a wave each.
// The frames' distinct-voxel statistics (voxels with a run) are counted here, for every voxel.
__global__ void __launch_bounds__(256) k_seq_classify(BatchDev B, const SeqDev* __restrict__ SD, int4* __restrict__ long_list, unsigned long long* __restrict__ lmask)
{
    __shared__ int s_cum[PLAN_NCLS + 1];
    __shared__ int s_uq[TSL_NB];
    __shared__ unsigned s_mx[TSL_NB];                          // longest run of a voxel in every frame: the chain that bounds the replay
    uint32_t okmask = 0u;
#pragma unroll
    for (int q = 0; q < TSL_NB; ++q) if (q < B.n && B.f[q].counters[HDR_FAIL] == 0) okmask |= 1u << q;
    if (threadIdx.x < TSL_NB) { s_uq[threadIdx.x] = 0; s_mx[threadIdx.x] = 0u; }
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int c = 0; c < PLAN_NCLS; ++c) { s_cum[c] = acc; acc += min(B.f[0].counters[HDR_UNITS + c], B.f[0].unit_cap); }
        s_cum[PLAN_NCLS] = acc;
    }
    __syncthreads();
    const int total = s_cum[PLAN_NCLS] * 16;
    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        const int u = item >> 4, l = ((item & 15) << 8) | (int)threadIdx.x;
        int c = 0;
#pragma unroll
        for (int j = 1; j < PLAN_NCLS; ++j) c += u >= s_cum[j] ? 1 : 0;
        const int4 e = B.f[0].unit_tab[(size_t)c * B.f[0].unit_cap + (u - s_cum[c])];
        const int b = e.x, pool = e.z;
        const uint32_t tm = (uint32_t)e.w & okmask;
        if (pool < 0 || tm == 0u) continue;
        // the runs of this voxel in every frame of the batch: the offsets of every frame's first slot are requested before a run is walked.
        // (The per-frame code is instantiated eight times with the frame as a compile-time constant: the offsets stay in registers.)
        uint32_t o0[TSL_NB], o1[TSL_NB], word[TSL_NB];
        bool is_long = false, is_xlong = false;
        uint32_t has = 0u;
        auto load_frame = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            o0[q] = o1[q] = word[q] = 0u;
            if ((tm >> q) & 1u) {
                word[q] = (uint32_t)B.f[q].bslab[b];
                if (word[q] >> SQ_SLOT_BITS) {
                    const uint32_t* csr = SD[q].csr + (size_t)(word[q] & ((1u << SQ_SLOT_BITS) - 1u)) * SQ_CSR_STRIDE;
                    o0[q] = csr[l]; o1[q] = csr[l + 1];
                }
            }
        };
        auto length_of_frame = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            uint32_t len = o1[q] - o0[q];
            const uint32_t np = word[q] >> SQ_SLOT_BITS;
            for (uint32_t p0 = 1u; p0 < np; p0 += 8u) {      // a brick next to the sensor: further slots (chunks of its segments in rank order), eight at a time in flight
                uint32_t a[8], z[8];
#pragma unroll
                for (uint32_t k = 0u; k < 8u; ++k) {
                    a[k] = z[k] = 0u;
                    if (p0 + k < np) { const uint32_t* csr = SD[q].csr + (size_t)((word[q] & ((1u << SQ_SLOT_BITS) - 1u)) + p0 + k) * SQ_CSR_STRIDE; a[k] = csr[l]; z[k] = csr[l + 1]; }
                }
#pragma unroll
                for (uint32_t k = 0u; k < 8u; ++k) len += z[k] - a[k];
            }
            is_long = is_long || len >= (uint32_t)SQ_LONG;
            is_xlong = is_xlong || len >= (uint32_t)SQ_XLONG;
            has |= (len ? 1u : 0u) << q;
            const unsigned long long m = __ballot(len != 0u);           // dense_tsdf.py has no such counter; the frame statistics report the voxels a frame updated
            if (m && lane_id() == 0) atomicAdd(&s_uq[q], popc64(m));
            uint32_t mx = len;
            for (int d = 32; d > 0; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, d); mx = o > mx ? o : mx; }
            if (mx && lane_id() == 0) atomicMax(&s_mx[q], mx);
        };
#define SQ_ALL_FRAMES(F) F(std::integral_constant<int, 0>{}); F(std::integral_constant<int, 1>{}); F(std::integral_constant<int, 2>{}); F(std::integral_constant<int, 3>{}); \
                         F(std::integral_constant<int, 4>{}); F(std::integral_constant<int, 5>{}); F(std::integral_constant<int, 6>{}); F(std::integral_constant<int, 7>{});
        static_assert(TSL_NB == 8, "eight frames per batch");
        SQ_ALL_FRAMES(load_frame)
        SQ_ALL_FRAMES(length_of_frame)
        {   // hand the long ones over (one reservation per wave); a voxel that does not fit the list stays here.  The longest chains -- they set the length of
            // the whole replay -- go to a list of their own that the long role of k_seq_replay walks first.
            const unsigned long long xm = __ballot(is_xlong);
            if (xm) {
                int base = 0;
                if (lane_id() == (int)__builtin_ctzll(xm)) base = __hip_atomic_fetch_add(&B.f[0].counters[HDR_SEQ_XLONG], popc64(xm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                base = __shfl(base, (int)__builtin_ctzll(xm));
                const int pos = base + rank_below(xm);
                if (is_xlong) { if (pos < SQ_XLONG_CAP) long_list[pos] = make_int4(pool, b, l, (int)tm); else is_xlong = false; }
            }
            const bool plain = is_long && !is_xlong;
            const unsigned long long lm = __ballot(plain);
            if (lm) {
                int base = 0;
                if (lane_id() == (int)__builtin_ctzll(lm)) base = __hip_atomic_fetch_add(&B.f[0].counters[HDR_SEQ_LONG], popc64(lm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                base = __shfl(base, (int)__builtin_ctzll(lm));
                const int pos = SQ_XLONG_CAP + base + rank_below(lm);
                if (plain) { if (pos < SQ_LONG_CAP) long_list[pos] = make_int4(pool, b, l, (int)tm); else is_long = false; }
            }
        }
        {   // which lanes of this wave the long role of k_seq_replay takes: the replay kernel's short-run role reads the word instead of measuring again
            const unsigned long long lm = __ballot(is_long);
            if (lane_id() == 0) lmask[(size_t)item * 4 + (threadIdx.x >> 6)] = lm;
        }
    }
#undef SQ_ALL_FRAMES
    __syncthreads();
    if (threadIdx.x < TSL_NB && s_uq[threadIdx.x]) atomic_add_i64(&B.f[threadIdx.x].stats->unique, (long long)s_uq[threadIdx.x]);
    if (threadIdx.x < TSL_NB && threadIdx.x < (unsigned)B.n && s_mx[threadIdx.x]) atomicMax(reinterpret_cast<unsigned*>(&B.f[threadIdx.x].counters[HDR_SEQ_MAXRUN]), s_mx[threadIdx.x]);
}

template <bool TEX>
__device__ __forceinline__ void seq_role_short(const MapDev& M, const BatchDev& B, const SeqDev* __restrict__ SD, const unsigned long long* __restrict__ lmask, int first_block, int nblocks)
{
    __shared__ int s_cum[PLAN_NCLS + 1];
    uint32_t okmask = 0u;
#pragma unroll
    for (int q = 0; q < TSL_NB; ++q) if (q < B.n && B.f[q].counters[HDR_FAIL] == 0) okmask |= 1u << q;
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int c = 0; c < PLAN_NCLS; ++c) { s_cum[c] = acc; acc += min(B.f[0].counters[HDR_UNITS + c], B.f[0].unit_cap); }
        s_cum[PLAN_NCLS] = acc;
    }
    __syncthreads();
    const int total = s_cum[PLAN_NCLS] * 16;
    for (int item = (int)blockIdx.x - first_block; item < total; item += nblocks) {
        const int u = item >> 4, l = ((item & 15) << 8) | (int)threadIdx.x;
        int c = 0;
#pragma unroll
        for (int j = 1; j < PLAN_NCLS; ++j) c += u >= s_cum[j] ? 1 : 0;
        const int4 e = B.f[0].unit_tab[(size_t)c * B.f[0].unit_cap + (u - s_cum[c])];
        const int b = e.x, pool = e.z;
        const uint32_t tm = (uint32_t)e.w & okmask;
        if (pool < 0 || tm == 0u) continue;
        // the runs of this voxel in every frame of the batch: the offsets of every frame's first slot are requested before a run is walked.
        // (The per-frame code is instantiated eight times with the frame as a compile-time constant: the offsets stay in registers.)
        uint32_t o0[TSL_NB], o1[TSL_NB], rb[TSL_NB], unsafe[TSL_NB], word[TSL_NB], pmask[TSL_NB];      // pmask: which of the slots 1..32 hold a run of this voxel (33..: always visited)
        uint32_t has = 0u;
        auto load_frame = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            o0[q] = o1[q] = rb[q] = unsafe[q] = word[q] = 0u;
            if ((tm >> q) & 1u) {
                word[q] = (uint32_t)B.f[q].bslab[b];
                if (word[q] >> SQ_SLOT_BITS) {
                    const uint32_t* csr = SD[q].csr + (size_t)(word[q] & ((1u << SQ_SLOT_BITS) - 1u)) * SQ_CSR_STRIDE;
                    o0[q] = csr[l]; o1[q] = csr[l + 1]; rb[q] = csr[SQ_CSR_BASE]; unsafe[q] = csr[SQ_CSR_UNSAFE];
                }
            }
        };
        auto length_of_frame = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            uint32_t len = o1[q] - o0[q];
            const uint32_t np = word[q] >> SQ_SLOT_BITS;
            pmask[q] = 0u;
            for (uint32_t p0 = 1u; p0 < np; p0 += 8u) {      // a brick next to the sensor: further slots (chunks of its segments in rank order), eight at a time in flight
                uint32_t a[8], z[8];
#pragma unroll
                for (uint32_t k = 0u; k < 8u; ++k) {
                    a[k] = z[k] = 0u;
                    if (p0 + k < np) { const uint32_t* csr = SD[q].csr + (size_t)((word[q] & ((1u << SQ_SLOT_BITS) - 1u)) + p0 + k) * SQ_CSR_STRIDE; a[k] = csr[l]; z[k] = csr[l + 1]; }
                }
#pragma unroll
                for (uint32_t k = 0u; k < 8u; ++k) { len += z[k] - a[k]; if (z[k] > a[k] && p0 + k <= 32u) pmask[q] |= 1u << (p0 + k - 1u); }
            }
            has |= (len ? 1u : 0u) << q;
        };
#define SQ_ALL_FRAMES(F) F(std::integral_constant<int, 0>{}); F(std::integral_constant<int, 1>{}); F(std::integral_constant<int, 2>{}); F(std::integral_constant<int, 3>{}); \
                         F(std::integral_constant<int, 4>{}); F(std::integral_constant<int, 5>{}); F(std::integral_constant<int, 6>{}); F(std::integral_constant<int, 7>{});
        static_assert(TSL_NB == 8, "eight frames per batch");
        SQ_ALL_FRAMES(load_frame)
        SQ_ALL_FRAMES(length_of_frame)
        const bool is_long = (lmask[(size_t)item * 4 + (threadIdx.x >> 6)] >> (threadIdx.x & 63)) & 1ull;      // k_seq_classify's verdict: the long role takes it
        if (is_long || has == 0u) continue;
        const size_t v = (size_t)pool * TSL_BRK3 + (size_t)l;
        const uint32_t old = M.tw[v];
        h16 T0 = (h16)(old & 0xffffu), W0 = (h16)(old >> 16);
        auto walk_frame = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if (!((has >> q) & 1u)) return;
            const uint32_t np = word[q] >> SQ_SLOT_BITS, slot0 = word[q] & ((1u << SQ_SLOT_BITS) - 1u);
            uint32_t last_ray_at = 0u;
            if (o1[q] > o0[q]) { seq_walk_run(SD[q].tup + rb[q], o0[q], o1[q], unsafe[q] == 0u, T0, W0); last_ray_at = rb[q] + o1[q] - 1u; }
            for (uint32_t p = 1u; p < np; ++p) {
                if (p <= 32u && !((pmask[q] >> (p - 1u)) & 1u)) continue;              // nothing of this voxel in that chunk
                const uint32_t* csr = SD[q].csr + (size_t)(slot0 + p) * SQ_CSR_STRIDE;
                const uint32_t a = csr[l], z = csr[l + 1], base = csr[SQ_CSR_BASE], us = csr[SQ_CSR_UNSAFE];
                if (z <= a) continue;
                seq_walk_run(SD[q].tup + base, a, z, us == 0u, T0, W0);
                last_ray_at = base + z - 1u;
            }
            if (TEX) reinterpret_cast<uint2*>(M.col)[v] = B.f[q].colpix[B.f[q].rayFirst[SD[q].tup_ray[last_ray_at]]];          // :268-269: every step stores its ray's colour, the run's last ray stays
        };
        SQ_ALL_FRAMES(walk_frame)
#undef SQ_ALL_FRAMES
        M.tw[v] = (uint32_t)T0 | ((uint32_t)W0 << 16); M.obs[v] = 1;                                                      // :265
        M.touch[pool] = 1;
    }
}

// The long runs: ONE WAVE PER VOXEL, wave-uniform.  A wave issues one instruction per four cycles however many lanes are active, so what a long
// chain costs is instructions per update -- unless the 64 lanes can do 64 updates at once, which they can wherever the f16 state has stopped
// moving (see the chain below): the replay tuples stream from HBM 1 024 at a time and a round of 64 costs one evaluation (about twenty
// instructions) as long as none of them changes (T, W).
// The waves run at raised priority: phase A of the next batch (k_seq_group, eight waves per CU) shares the SIMDs, and a chain that gets
// every third issue slot is three times as long.
template <bool TEX>
__device__ __forceinline__ void seq_role_long(const MapDev& M, const BatchDev& B, const SeqDev* __restrict__ SD, const int4* __restrict__ long_list, int nblocks)
{
    __shared__ uint32_t s_lx[4][64], s_ls[4][64];    // per wave: the concatenated sequence of a voxel's runs in 64 slots (first update of each slot; tuple index - that)
    __builtin_amdgcn_s_setprio(3);
    const int wid = threadIdx.x >> 6, lane = lane_id();
    uint32_t* const s_ex = s_lx[wid]; uint32_t* const s_st = s_ls[wid];
#ifdef TSL_SEQ_DBG
    int dbg_e = 0, dbg_t = 0, dbg_w = 0;      // developer build: evaluations, commits that moved T, commits that moved W only
#endif
    uint32_t okmask = 0u;
#pragma unroll
    for (int q = 0; q < TSL_NB; ++q) if (q < B.n && B.f[q].counters[HDR_FAIL] == 0) okmask |= 1u << q;
    const int nx = min(B.f[0].counters[HDR_SEQ_XLONG], SQ_XLONG_CAP), n = nx + min(B.f[0].counters[HDR_SEQ_LONG], SQ_LONG_CAP - SQ_XLONG_CAP);
    for (int i = blockIdx.x * 4 + wid; i < n; i += nblocks * 4) {         // the longest chains first
        const int4 e = long_list[i < nx ? i : SQ_XLONG_CAP + (i - nx)];
        const int pool = uni_i(e.x), b = uni_i(e.y), l = uni_i(e.z);
        const uint32_t tm = (uint32_t)uni_i(e.w) & okmask;
        const size_t v = (size_t)pool * TSL_BRK3 + (size_t)l;
        const uint32_t old = (uint32_t)uni_i((int)M.tw[v]);
        uint32_t Tb = old & 0xffffu, Wb = old >> 16;
        for (int q = 0; q < TSL_NB; ++q) {
            if (!((tm >> q) & 1u)) continue;
            const uint32_t word = (uint32_t)uni_i(B.f[q].bslab[b]);
            const uint32_t np = word >> SQ_SLOT_BITS, slot0 = word & ((1u << SQ_SLOT_BITS) - 1u);
            uint32_t last_ray_at = 0xffffffffu;
            for (uint32_t pb = 0u; pb < np; pb += 64u) {
                // the voxel's run of this frame in 64 slots at a time, one slot per lane: one memory round trip for all of them
                uint4 meta = make_uint4(0u, 0u, 0u, 0u);
                if (pb + (uint32_t)lane < np) {
                    const uint32_t* csr = SD[q].csr + (size_t)(slot0 + pb + (uint32_t)lane) * SQ_CSR_STRIDE;
                    meta = make_uint4(csr[l], csr[l + 1], csr[SQ_CSR_BASE], csr[SQ_CSR_UNSAFE]);
                }
                const bool has = meta.y > meta.x;
                const unsigned long long pm = __ballot(has);
                if (pm == 0ull) continue;
                last_ray_at = (uint32_t)__builtin_amdgcn_readlane((int)(meta.z + meta.y - 1u), 63 - (int)__builtin_clzll(pm));
                if (__ballot(has && meta.w != 0u) != 0ull || !(fabsf(h2f((h16)Tb)) <= 60.0f)) {
                    // outside the division-free form's range: the literal expression, slot after slot (never seen in practice)
                    h16 T0 = (h16)Tb, W0 = (h16)Wb;
                    for (unsigned long long pq = pm; pq; pq &= pq - 1ull) {
                        const int src = (int)__builtin_ctzll(pq);
                        const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)meta.x, src), o1 = (uint32_t)__builtin_amdgcn_readlane((int)meta.y, src);
                        const float2* const tp = SD[q].tup + (uint32_t)__builtin_amdgcn_readlane((int)meta.z, src);
                        for (uint32_t t = o0; t < o1; ++t) {
                            const float2 x = tp[t];
                            const h16 Tn = f2h((h2f(hmul(T0, W0)) + x.y) / (h2f(W0) + x.x));                              // dense_tsdf.py:264
                            float wn = h2f(W0) + x.x; if (TSL_WMAX < wn) wn = TSL_WMAX;                                   // :267
                            T0 = Tn; W0 = f2h(wn);
                        }
                    }
                    Tb = (uint32_t)uni_i((int)T0); Wb = (uint32_t)uni_i((int)W0);
                    continue;
                }
                // The chain.  Deep into a voxel's life an update mostly leaves BOTH f16 values as they are: the running mean stalls (an increment of
                // w / W of the residual is far below half an f16 ulp of T), and so does the weight (RN16(W + w) = W once w is under half an ulp of W:
                // w < 0.25 from W = 512 on -- a voxel seen from two metres never reaches Wmax).  Whether update k changes the state can be decided
                // without the updates before it, as long as THEY did not: the 64 lanes evaluate 64 consecutive updates on the same (T, W); if none
                // changes it, 64 updates are done; otherwise the first lane that does holds the true next state (everything before it was a no-op)
                // and the lanes behind it go again.  Exact by construction; at worst -- a state that moves at every update, a voxel's first few
                // hundred -- one evaluation per update, as in a plain walk.
                // The 64 consecutive updates have to be FOUND first: next to the sensor a brick's segments are cut into tens of chunks by ray rank
                // (k_seq_split) and a voxel's run of a frame is scattered over their slots, a few tuples in each -- walked slot by slot the wave spent
                // ~150 instructions on every two or three updates (77 M wave-instructions per batch, twice the short runs' role).  So the runs of
                // the 64 slots are concatenated: a prefix sum of their lengths over the lanes, update k of the sequence found by a binary search
                // in it (wave-private LDS), the tuples gathered 512 per trip.
                const uint32_t len = has ? meta.y - meta.x : 0u;
                uint32_t incl = len;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if (lane >= d) incl += o; }
                const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                s_ex[lane] = incl - len;                                             // first update of the slot in the concatenated sequence
                s_st[lane] = meta.z + meta.x - (incl - len);                         // + k = the tuple of update k
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();      // (LDS operations of a wave complete in order)
                const float2* const tq = SD[q].tup;
                for (uint32_t k0 = 0u; k0 < n; k0 += 64u * SQ_LG) {
                    float2 x[SQ_LG];
#pragma unroll
                    for (int g = 0; g < SQ_LG; ++g) {
                        const uint32_t k = min(k0 + (uint32_t)(g * 64 + lane), n - 1u);
                        uint32_t lo = 0u;
#pragma unroll
                        for (uint32_t step = 32u; step; step >>= 1) lo += s_ex[lo + step] <= k ? step : 0u;      // the last slot that starts at or before k (it is not empty)
                        x[g] = tq[s_st[lo] + k];
                    }
                    uint32_t Tr = Tb;
#pragma unroll
                    for (int g = 0; g < SQ_LG; ++g) {
                        const uint32_t base = k0 + (uint32_t)(g * 64);
                        if (base >= n) continue;                                   // (uniform)
                        unsigned long long todo = __ballot(base + (uint32_t)lane < n);
                        while (todo) {
#ifdef TSL_SEQ_DBG
                            ++dbg_e;
#endif
                            // every lane of `todo` evaluates its update on a CANDIDATE state: T as it is, W as the updates before it in the group
                            // are expected to have left it; the lane is consistent if T comes out unchanged and W comes out as the next lane's candidate.
                            // Lanes before the first inconsistent one are thereby verified one after the other (each started from a true state), and so
                            // are that lane's inputs: its outputs are the true next state.  At Wmax, and where the weight has stalled below it, the
                            // candidate is W itself; while the weight still grows it is W + the increments rounded to W's f16 grid, summed over the lanes
                            // before (a guess: ties, a binade crossed on the way or an inexact f32 sum only make a lane inconsistent, never the result wrong:
                            // INVARIANT -- lane i + 1 evaluates on exactly the f16 weight lane i is checked against).
                            uint32_t Tn, Wn, bad;
                            if (Wb == SQ_W_SAT) {                                  // (uniform)
                                const float D = TSL_WMAX + x[g].x;
                                Tn = seq_update_fast(Tr, Wb, x[g].y, D, 1.0f / D);                                        // :264
                                Wn = SQ_W_SAT;                                                                            // :267
                                bad = (Tn ^ Tr) & 0xffffu;
                            } else {
                                const float Wf = h2f((h16)Wb);
                                const uint32_t eb = (Wb >> 10) & 31u;
                                const float grid = __uint_as_float((eb ? eb + 102u : 103u) << 23), ginv = __uint_as_float((eb ? 152u - eb : 151u) << 23);      // ulp of W in f16: 2^(e - 25), 2^-24 for a subnormal; and 1 / that
                                const float inc = ((todo >> lane) & 1ull) ? rintf(x[g].x * ginv) * grid : 0.0f;
                                float incl = inc;
#pragma unroll
                                for (int d = 1; d < 64; d <<= 1) { const float o = __shfl_up(incl, d); if (lane >= d) incl += o; }
                                // the weight a lane EXPECTS to leave behind is, literally, the weight the next lane starts from (the value travels one lane
                                // up; the first lane of `todo` starts from the true W): a lane whose result equals its expectation has thereby verified the
                                // next lane's input, whatever the f32 prefix sums rounded to -- exact by construction, not by the sums being exact
                                const uint32_t Wexp = (uint32_t)f2h(fminf(Wf + incl, TSL_WMAX));
                                const uint32_t Wprev = (uint32_t)__shfl_up((int)Wexp, 1);
                                const uint32_t Wc = lane == (int)__builtin_ctzll(todo) ? Wb : Wprev;
                                const float D = h2f((h16)Wc) + x[g].x;
                                Tn = seq_update_fast(Tr, Wc, x[g].y, D, 1.0f / D);                                        // :264
                                Wn = (uint32_t)f2h(D > TSL_WMAX ? TSL_WMAX : D);                                          // :267
                                bad = ((Tn ^ Tr) & 0xffffu) | (Wn ^ Wexp);
                            }
                            const unsigned long long cm = __ballot(bad != 0u) & todo;
                            if (cm == 0ull) { Wb = (uint32_t)__builtin_amdgcn_readlane((int)Wn, 63 - (int)__builtin_clzll(todo)); break; }      // every update of the group verified: W as the last one left it
                            const int j = (int)__builtin_ctzll(cm);
#ifdef TSL_SEQ_DBG
                            { const uint32_t tn = (uint32_t)__builtin_amdgcn_readlane((int)Tn, j) & 0xffffu; if (tn != Tr) ++dbg_t; else ++dbg_w; }
#endif
                            Tr = (uint32_t)__builtin_amdgcn_readlane((int)Tn, j) & 0xffffu;
                            Wb = (uint32_t)__builtin_amdgcn_readlane((int)Wn, j);
                            todo &= ~((2ull << j) - 1ull);                        // the lanes behind j (j = 63: none)
                        }
                    }
                    Tb = (uint32_t)uni_i((int)(Tr & 0xffffu));
                }
                __builtin_amdgcn_wave_barrier();                                   // (the rows are written again for the next 64 slots)
            }
            if (TEX && lane == 0 && last_ray_at != 0xffffffffu) reinterpret_cast<uint2*>(M.col)[v] = B.f[q].colpix[B.f[q].rayFirst[SD[q].tup_ray[last_ray_at]]];      // :268-269
        }
        if (lane == 0) { M.tw[v] = Tb | (Wb << 16); M.obs[v] = 1; M.touch[pool] = 1; }                                    // :265
#ifdef TSL_SEQ_DBG
        if (lane == 0) { atomicAdd(&B.f[0].counters[32], dbg_e); atomicAdd(&B.f[0].counters[33], dbg_t); atomicAdd(&B.f[0].counters[34], dbg_w); } dbg_e = dbg_t = dbg_w = 0;
#endif
    }
}

// phase B of a batch, ONE launch: the first `nlong` workgroups replay the long runs (a wave per voxel, the longest chains first), the others the
// short ones (a lane per voxel).  The two sets of voxels are disjoint (k_seq_classify), so nothing orders them against each other; in one
// launch they run side by side, and the launch is as long as the longest chain -- the voxel next to the sensor -- not chain + the rest.
template <bool TEX>
__global__ void __launch_bounds__(256) k_seq_replay(MapDev M, BatchDev B, const SeqDev* __restrict__ SD, const int4* __restrict__ long_list, const unsigned long long* __restrict__ lmask, int nlong)
{
    if ((int)blockIdx.x < nlong) seq_role_long<TEX>(M, B, SD, long_list, nlong);
    else seq_role_short<TEX>(M, B, SD, lmask, nlong, (int)gridDim.x - nlong);
}

// tsl_selftest(2): the division-free update against the literal expression, 2^32 operand tuples drawn from what the replay sees (any f16 value
// up to 60 in magnitude, any weight code, signed distances over 32 binades, the state's weight at Wmax or anywhere below it)
__global__ void __launch_bounds__(256) k_selftest_seqdiv(unsigned long long* bad)
{
    unsigned long long st = ((unsigned long long)blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    long long nbad = 0;
    for (int it = 0; it < 2048; ++it) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t a = (uint32_t)(st >> 32), bq = (uint32_t)st;
        // w: any positive finite f16 z^2 code (as the tuples carry it); T: any f16 value up to 60 in magnitude; sd: a float of up to 60 in
        // magnitude with a random exponent over 32 binades, either sign
        uint32_t zc = a & 0x7fffu; if (zc == 0u) zc = 1u; if (zc >= 0x7c00u) zc = 0x3c00u;
        const float w = seq_w_of((h16)zc);
        uint32_t tb = a >> 16; if (!(fabsf(h2f((h16)tb)) <= 60.0f)) tb &= 0xbfffu;
        float sd = __uint_as_float((bq & 0x80000000u) | ((132u - ((bq >> 23) & 31u)) << 23) | (bq & 0x7fffffu));
        if (fabsf(sd) > 60.0f) sd *= 0.5f;
        const float c = w * sd;
        if (c != 0.0f && fabsf(c) < 8.67e-19f) continue;
        // the state's weight: Wmax, or any f16 value below it
        const uint32_t wb = (it & 1) ? SQ_W_SAT : (uint32_t)(((bq >> 8) ^ a) & 0x7fffu) % (SQ_W_SAT + 1u);
        const float D = h2f((h16)wb) + w;
        const h16 want = f2h((h2f(hmul((h16)tb, (h16)wb)) + c) / D);                                                      // the literal expression, dense_tsdf.py:264
        const uint32_t got = seq_update_fast(tb | (bq << 16), wb | (a << 16), c, D, 1.0f / D);      // (garbage in the high halves: they must not matter)
        if (want != (h16)got) ++nbad;
    }
    nbad = wave_sum_ll(nbad);
    if (lane_id() == 0 && nbad) atomicAdd(bad, (unsigned long long)nbad);
}
int selftest_seqdiv(unsigned long long* bad_dev) { hipLaunchKernelGGL(k_selftest_seqdiv, dim3(8192), dim3(256), 0, 0, bad_dev); return TSL_OK; }

// Developer aid (TSL_SEQ_VERIFY=1; round 5's hunt for a one-brick-in-a-few-hundred-batches difference): an order-free checksum of every work item's run
// offsets and replay tuples, taken behind k_seq_group on the batch's stream (stage 0), in front of the replay on the main stream (1) and behind it
// (2).  Stages 1 and 2 compare with stage 0 and log what differs: { batch, frame | stage << 8, slot, 1: offsets 2: tuples }.
struct SeqVerify { unsigned long long* sum; int* log; int log_cap; };      // sum: [TSL_NB][slot_cap][2]; log: [0] = entries, then int4 records
__global__ void __launch_bounds__(256) k_seq_hash(BatchDev B, const SeqDev* __restrict__ SD, SeqVerify V, int stage, int batch_no)
{
    const int q = blockIdx.y;
    if (q >= B.n) return;
    const FrameDev& F = B.f[q];
    const SeqDev S = SD[q];
    if (F.counters[HDR_FAIL] != 0) return;
    const int nitems = min(F.counters[HDR_SEQ_SLOTS], S.slot_cap);
    __shared__ unsigned long long s_sum[2];
    for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
        const uint32_t* csr = S.csr + (size_t)it * SQ_CSR_STRIDE;
        if (threadIdx.x < 2) s_sum[threadIdx.x] = 0ull;
        __syncthreads();
        unsigned long long a = 0ull, b = 0ull;
        for (int i = threadIdx.x; i < SQ_CSR_UNSAFE + 1; i += 256) a += ((unsigned long long)csr[i] + 0x9E3779B97F4A7C15ull) * (unsigned long long)(2 * i + 1);
        const uint32_t T = csr[TSL_BRK3], rb = csr[SQ_CSR_BASE];
        const unsigned long long* tp = reinterpret_cast<const unsigned long long*>(S.tup + rb);
        for (uint32_t i = threadIdx.x; i < T; i += 256) b += (tp[i] ^ 0xD6E8FEB86659FD93ull) * (unsigned long long)(2u * i + 1u);
        a = (unsigned long long)wave_sum_ll((long long)a); b = (unsigned long long)wave_sum_ll((long long)b);
        if (lane_id() == 0) { atomicAdd(&s_sum[0], a); atomicAdd(&s_sum[1], b); }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long* rec = V.sum + ((size_t)q * S.slot_cap + it) * 2;
            if (stage == 0) { rec[0] = s_sum[0]; rec[1] = s_sum[1]; }
            else {
                const int what = (rec[0] != s_sum[0] ? 1 : 0) | (rec[1] != s_sum[1] ? 2 : 0);
                if (what) { const int e = atomicAdd(&V.log[0], 1); if (e < V.log_cap) reinterpret_cast<int4*>(V.log + 4)[e] = make_int4(batch_no, q | (stage << 8), it, what | ((int)T << 4)); }
            }
        }
        __syncthreads();
    }
}
// TSL_SEQ_VERIFY, second check: every item recomputed from its segment list by brute force -- per voxel the number of steps and an order-free sum over
// their replay tuples -- and compared with what k_seq_group left (run lengths from the offsets, the same sum over the voxel's run).  Log record:
// { batch, frame | 3 << 8, slot, voxel | (expected steps << 12) | (found << 22) }.
__global__ void __launch_bounds__(256) k_seq_check(MapDev M, BatchDev B, const SeqDev* __restrict__ SD, SeqVerify V, int batch_no)
{
    const int q = blockIdx.y;
    if (q >= B.n) return;
    const FrameDev& F = B.f[q];
    const FrameParams& P = *B.p[q];
    const SeqDev S = SD[q];
    if (F.counters[HDR_FAIL] != 0) return;
    const int nitems = min(F.counters[HDR_SEQ_SLOTS], S.slot_cap);
    __shared__ uint32_t s_cnt[TSL_BRK3];
    __shared__ uint32_t s_sum[TSL_BRK3];
    for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int4 item = S.items[it];
        const unsigned long long* const src = (item.w ? F.seg : F.seg_sorted) + item.x;
        const uint32_t* csr = S.csr + (size_t)item.z * SQ_CSR_STRIDE;
        for (int i = threadIdx.x; i < TSL_BRK3; i += 256) { s_cnt[i] = 0u; s_sum[i] = 0u; }
        __syncthreads();
        for (int k = threadIdx.x; k < item.y; k += 256) {
            const unsigned long long sg = src[k];
            const int cnt = (int)(sg & 63ull), j0 = (int)((sg >> SEG_CNT_BITS) & 0xfffull), ray = (int)((sg >> (SEG_CNT_BITS + SEG_J_BITS)) & ((1u << STG_RAY_BITS) - 1));
            const uint4 rec = F.rayA[ray];
            const float pf0 = h2f((h16)(rec.x & 0xffffu)), pf1 = h2f((h16)(rec.x >> 16)), pf2 = h2f((h16)(rec.y & 0xffffu));
            const float d0 = h2f((h16)(rec.y >> 16)), d1 = h2f((h16)(rec.z & 0xffffu)), d2 = h2f((h16)(rec.z >> 16));
            const float w = seq_w_of(seq_w_code(__uint_as_float(rec.w)));
            const float P0 = pf0 + P.T[0], P1 = pf1 + P.T[1], P2f = pf2 + P.T[2];
            for (int st = 0; st < cnt; ++st) {
                const float jf = (float)(j0 + st);
                const float x0 = (d0 * jf) * P.vs + P.T[0], x1 = (d1 * jf) * P.vs + P.T[1], x2 = (d2 * jf) * P.vs + P.T[2];
                const int i0 = rnd_i(div_vs(x0, P.vs, P.rvs, P.fastdiv)), i1 = rnd_i(div_vs(x1, P.vs, P.rvs, P.fastdiv)), i2 = rnd_i(div_vs(x2, P.vs, P.rvs, P.fastdiv));
                const int l = (((i0 + M.hN) & 15) << 8) | (((i1 + M.hN) & 15) << 4) | ((i2 + M.hNz) & 15);
                const float v0 = P0 - x0, v1 = P1 - x1, v2 = P2f - x2;
                const float s2 = (v0 * v0 + v1 * v1) + v2 * v2;
                const float dist = s2 >= 1.2621774483536189e-29f ? sqrt_rn_norm(s2) : sqrt_rn(s2);
                const float dot = (v0 * pf0 + v1 * pf1) + v2 * pf2;
                const float sd = dist * (float)sgn_f(dot);
                const float c = w * sd;
                atomicAdd(&s_cnt[l], 1u);
                atomicAdd(&s_sum[l], __float_as_uint(w) * 2654435761u + __float_as_uint(c) * 40503u + 1u);
            }
        }
        __syncthreads();
        const uint32_t rb = csr[SQ_CSR_BASE];
        for (int l = threadIdx.x; l < TSL_BRK3; l += 256) {
            const uint32_t a = csr[l], z = csr[l + 1];
            uint32_t sum = 0u;
            for (uint32_t t = a; t < z && t < a + 100000u; ++t) { const float2 x = S.tup[rb + t]; sum += __float_as_uint(x.x) * 2654435761u + __float_as_uint(x.y) * 40503u + 1u; }
            if (z - a != s_cnt[l] || sum != s_sum[l]) {
                const int e = atomicAdd(&V.log[0], 1);
                if (e < V.log_cap) reinterpret_cast<int4*>(V.log + 4)[e] = make_int4(batch_no, q | (3 << 8), item.z, l | ((int)min(s_cnt[l], 1023u) << 12) | ((int)min(z - a, 1023u) << 22));
            }
        }
        __syncthreads();
    }
}
static bool seq_verify_on() { static const bool on = std::getenv("TSL_SEQ_VERIFY") != nullptr; return on; }
static int launch_seq_hash(tsl_tsdf* m, const BatchDev& B, int bi, int stage, hipStream_t st)
{
    SeqVerify V = { static_cast<unsigned long long*>(m->seqv_sum[bi]), m->seqv_log, 4096 };
    hipLaunchKernelGGL(k_seq_hash, dim3(512, B.n), dim3(256), 0, st, B, (const SeqDev*)(m->seq_d + bi * TSL_NB), V, stage, (int)m->batch_seq);
    return TSL_OK;
}
int seq_verify_report(tsl_tsdf* m, int* out, int cap)      // out: [0] = mismatches logged, then 4 ints each
{
    if (!m->seqv_log) { out[0] = -1; return TSL_OK; }
    TSL_HIP(hipMemcpy(out, m->seqv_log, sizeof(int) * (size_t)cap, hipMemcpyDeviceToHost));
    return TSL_OK;
}

static int seq_ensure(tsl_tsdf* m)
{
    if (m->seq_ready) return TSL_OK;
    int rc;
    const size_t np = (size_t)m->F.max_points;
    if (m->seq_tuple_cap <= 0) m->seq_tuple_cap = 1ll << 23;          // a frame yields at most rays x steps tuples; 2^23 is 640 x 480 at recast_step 2 twice over (a frame beyond it fails loudly; option "seq_tuple_cap")
    const bool tex = m->cfg.texture_enabled != 0;
    for (int si = 0; si < TSL_NSETS; ++si) {
        SeqDev& S = m->seq_h[si];
        S.cap = m->seq_tuple_cap; S.stash_ray = nullptr; S.tup_ray = nullptr; S.items = nullptr;
        S.stash = nullptr;
#ifdef TSL_SEQ_STASH
        if ((rc = dev_alloc(m, (void**)&S.stash, 8 * (size_t)S.cap, 0))) return rc;
#endif
        if ((rc = dev_alloc(m, (void**)&S.tup, 8 * ((size_t)S.cap + 16), 0))) return rc;
        S.slot_cap = m->F.max_frame_bricks + 1024;          // one slot per (frame, brick) + the further chunks of the few bricks next to the sensor
        if ((rc = dev_alloc(m, (void**)&S.csr, 4 * (size_t)S.slot_cap * SQ_CSR_STRIDE, 0))) return rc;
        if ((rc = dev_alloc(m, (void**)&S.items, sizeof(int4) * (size_t)S.slot_cap, 0))) return rc;
        if (tex) {
#ifdef TSL_SEQ_STASH
            if ((rc = dev_alloc(m, (void**)&S.stash_ray, 4 * (size_t)S.cap, 0))) return rc;
#endif
            if ((rc = dev_alloc(m, (void**)&S.tup_ray, 4 * (size_t)S.cap, 0))) return rc;
        }
    }
    if ((rc = dev_alloc(m, (void**)&m->seq_d, sizeof(SeqDev) * TSL_NSETS, 0))) return rc;
    TSL_HIP(hipMemcpyAsync(m->seq_d, m->seq_h, sizeof(SeqDev) * TSL_NSETS, hipMemcpyHostToDevice, m->stream_));
    size_t tb = 0;
    TSL_HIP(rocprim::radix_sort_pairs(nullptr, tb, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, np * TSL_NB, 0u, 64u, m->stream_));
    m->seqb_temp_bytes = tb + 256;
    for (int bi = 0; bi < TSL_NBATCH; ++bi) {
        for (int k = 0; k < 2; ++k) {
            if ((rc = dev_alloc(m, &m->seqb_keys[bi][k], 8 * np * TSL_NB, 0))) return rc;
            if ((rc = dev_alloc(m, &m->seqb_vals[bi][k], 4 * np * TSL_NB, 0))) return rc;
        }
        if ((rc = dev_alloc(m, &m->seqb_temp[bi], m->seqb_temp_bytes, 0))) return rc;
        if ((rc = dev_alloc(m, &m->seqb_long[bi], sizeof(int4) * (size_t)SQ_LONG_CAP, 0))) return rc;
        if ((rc = dev_alloc(m, &m->seqb_perm[bi], sizeof(unsigned short) * SQ_SORTCAP * 3 * (size_t)m->ncu, 0))) return rc;          // k_seq_group: a row per workgroup of its launch
        if ((rc = dev_alloc(m, &m->seqb_lmask[bi], 8 * 64 * (size_t)PLAN_NCLS * TSL_NB * m->F.max_frame_bricks, 0))) return rc;          // a word per wave of 64 voxels of every brick a batch can list (unit_cap bricks per class)
    }
    if (seq_verify_on()) {
        for (int bi = 0; bi < TSL_NBATCH; ++bi) if ((rc = dev_alloc(m, &m->seqv_sum[bi], 16 * (size_t)TSL_NB * m->seq_h[0].slot_cap, 0))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->seqv_log, sizeof(int) * (4 + 4 * 4096), 0))) return rc;
    }
    TSL_HIP(hipStreamSynchronize(m->stream_));          // the fills ran on the main stream; the kernels below use the batch streams
    m->seq_ready = true;
    return TSL_OK;
}
// the literal mode's scratch, allocated when the mode is switched on (tsl_tsdf_set_option: a failure is reported there, with nothing queued that could be
// lost, and what was allocated is handed back -- ADVICE r4); the first sequential batch finds it ready
int seq_prepare(tsl_tsdf* m)
{
    const int64_t before = m->bytes;
    const int rc = seq_ensure(m);
    m->seq_bytes0 += m->bytes - before;          // what the literal scratch added to the handle's account (also of a failed attempt: released below)
    if (rc) seq_release(m);
    return rc;
}
void seq_release(tsl_tsdf* m)
{
    for (auto& S : m->seq_h) { void* p[] = { S.stash, S.tup, S.csr, S.stash_ray, S.tup_ray, S.items }; for (void* x : p) if (x) (void)hipFree(x); S = SeqDev(); }
    if (m->seq_d) (void)hipFree(m->seq_d);
    m->seq_d = nullptr;
    for (int bi = 0; bi < TSL_NBATCH; ++bi) {
        for (int k = 0; k < 2; ++k) { if (m->seqb_keys[bi][k]) (void)hipFree(m->seqb_keys[bi][k]); if (m->seqb_vals[bi][k]) (void)hipFree(m->seqb_vals[bi][k]); m->seqb_keys[bi][k] = m->seqb_vals[bi][k] = nullptr; }
        if (m->seqb_temp[bi]) (void)hipFree(m->seqb_temp[bi]);
        if (m->seqb_long[bi]) (void)hipFree(m->seqb_long[bi]);
        if (m->seqb_lmask[bi]) (void)hipFree(m->seqb_lmask[bi]);
        if (m->seqb_perm[bi]) (void)hipFree(m->seqb_perm[bi]);
        m->seqb_perm[bi] = nullptr;
        m->seqb_temp[bi] = nullptr; m->seqb_long[bi] = nullptr; m->seqb_lmask[bi] = nullptr;
        if (m->seqv_sum[bi]) (void)hipFree(m->seqv_sum[bi]);
        m->seqv_sum[bi] = nullptr;
    }
    if (m->seqv_log) (void)hipFree(m->seqv_log);
    m->seqv_log = nullptr;
    m->seq_ready = false;
    m->bytes -= m->seq_bytes0; m->seq_bytes0 = 0;      // tsl_tsdf_memory_bytes counted the scratch once and must not count it again after a rebuild
}

// behind phase A of a batch, on its stream: struct-for ranks of the batch's rays (one sort), then every (frame, brick)'s replay runs
int launch_seq_group(tsl_tsdf* m, const BatchDev& B, const FrameParams* hp, int bi, hipStream_t st)
{
    TSL_REQUIRE(hp[0].group && hp[0].variant == 2, "sequential semantics: the hash-grouped brick path (variant 2, group 1)");
    int rc = seq_prepare(m); if (rc) return rc;
    int stride = 0;
    for (int q = 0; q < B.n; ++q) stride = hp[q].total > stride ? hp[q].total : stride;
    if (stride <= 0) return TSL_OK;
    const int keybits = 3 * m->pcl_bits;                     // the struct-for key counts the cells of the sensor grid: < ext^3 <= 2^(3 bits)
    TSL_REQUIRE(keybits + 4 <= 64, "sequential semantics: sensor grid too large for the rank key");
    const int total = B.n * stride;
    unsigned long long* k0 = (unsigned long long*)m->seqb_keys[bi][0]; unsigned long long* k1 = (unsigned long long*)m->seqb_keys[bi][1];
    uint32_t* v0 = (uint32_t*)m->seqb_vals[bi][0]; uint32_t* v1 = (uint32_t*)m->seqb_vals[bi][1];
    prof_begin(m, TSL_K_SORT, st);
    hipLaunchKernelGGL(k_seq_keys, dim3((stride + 255) / 256, B.n), dim3(256), 0, st, B, k0, v0, stride, keybits);
    size_t tb = m->seqb_temp_bytes;
    TSL_HIP(rocprim::radix_sort_pairs(m->seqb_temp[bi], tb, k0, k1, v0, v1, (size_t)total, 0u, (unsigned)(keybits + 4), st));
    hipLaunchKernelGGL(k_seq_ranks, dim3((total + 255) / 256), dim3(256), 0, st, B, (const uint32_t*)v1, total);
    prof_end(m, st);
    prof_begin(m, TSL_K_RAYS, st);
    const int gx = m->F.max_frame_bricks < 1024 ? m->F.max_frame_bricks : 1024;
    hipLaunchKernelGGL(k_seq_split, dim3(gx, B.n), dim3(SQ_NT), 0, st, m->M, B, (const SeqDev*)(m->seq_d + bi * TSL_NB));
    if (hp[0].tex) hipLaunchKernelGGL(k_seq_group<true>, dim3(3 * m->ncu), dim3(SQ_NT), 0, st, m->M, B, (const SeqDev*)(m->seq_d + bi * TSL_NB), static_cast<unsigned short*>(m->seqb_perm[bi]));
    else hipLaunchKernelGGL(k_seq_group<false>, dim3(3 * m->ncu), dim3(SQ_NT), 0, st, m->M, B, (const SeqDev*)(m->seq_d + bi * TSL_NB), static_cast<unsigned short*>(m->seqb_perm[bi]));
    if (seq_verify_on()) {
        launch_seq_hash(m, B, bi, 0, st);
        SeqVerify V = { static_cast<unsigned long long*>(m->seqv_sum[bi]), m->seqv_log, 4096 };
        hipLaunchKernelGGL(k_seq_check, dim3(512, B.n), dim3(256), 0, st, m->M, B, (const SeqDev*)(m->seq_d + bi * TSL_NB), V, (int)m->batch_seq);
    }
    // which voxels get a wave of their own in the replay, the frames' distinct-voxel counts: run lengths only, nothing of the map
    hipLaunchKernelGGL(k_seq_classify, dim3(16 * m->ncu), dim3(256), 0, st, B, (const SeqDev*)(m->seq_d + bi * TSL_NB), static_cast<int4*>(m->seqb_long[bi]),
                       static_cast<unsigned long long*>(m->seqb_lmask[bi]));
    prof_end(m, st);
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}
int launch_seq_apply(tsl_tsdf* m, const BatchDev& B, const FrameParams& P, int bi)
{
    const SeqDev* sd = m->seq_d + bi * TSL_NB;
    const int4* ll = static_cast<const int4*>(m->seqb_long[bi]);
    const unsigned long long* lm = static_cast<const unsigned long long*>(m->seqb_lmask[bi]);
    const int nlong = 4 * m->ncu, nshort = 12 * m->ncu;
#ifdef TSL_TEST_HOOKS      // developer build only (include/taichislam_hip.h, "environment switches")
    static const bool split_roles = std::getenv("TSL_SEQ_SPLIT_ROLES") != nullptr;
#else
    constexpr bool split_roles = false;
#endif
    if (split_roles && !P.tex) {                   // developer timing aid: the two roles as two launches (same result: the voxel sets are disjoint)
        hipLaunchKernelGGL(k_seq_replay<false>, dim3(nshort), dim3(256), 0, m->stream_, m->M, B, sd, ll, lm, 0);
        hipLaunchKernelGGL(k_seq_replay<false>, dim3(nlong), dim3(256), 0, m->stream_, m->M, B, sd, ll, lm, nlong);
        return TSL_OK;
    }
    if (seq_verify_on()) launch_seq_hash(m, B, bi, 1, m->stream_);
    if (P.tex) hipLaunchKernelGGL(k_seq_replay<true>, dim3(nlong + nshort), dim3(256), 0, m->stream_, m->M, B, sd, ll, lm, nlong);
    else hipLaunchKernelGGL(k_seq_replay<false>, dim3(nlong + nshort), dim3(256), 0, m->stream_, m->M, B, sd, ll, lm, nlong);
    if (seq_verify_on()) launch_seq_hash(m, B, bi, 2, m->stream_);
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}

// =====================================================================================================================================
// Option "semantics" = 1 on a GLOBAL map: the reference-literal SEQUENTIAL fusion, fuse_submaps_kernel / fuse_with_interploation
// (dense_tsdf.py:272-318).  The reference walks every cell of every submap and, for seven of the eight surrounding global voxels, does an
// unsynchronised f16 read-modify-write of the running weighted average (:274-280); splats race.  The sequential schedule -- submap cells in
// struct-for order (submap, block of num_voxel_per_blk_axis^3 cells lexicographic, cell row-major inside the block), the seven corners in loop
// order -- is what the CPU checker's FAITHFUL fusion and tools/ti_seq execute.  Here: every splat becomes a tuple
//   key = global brick | global voxel | place of the source cell in that order | corner,   value = { w_tsdf, tsdf, occupancy };
// radix sort; one thread per global voxel applies its run in order.  (The default fusion, tsl_fuse.hip, sums the same terms exactly and
// divides once: order-free, the multi-GPU merge rests on it.)
// The place of a source cell: the map is stored in 16^3 bricks whatever num_voxel_per_blk_axis is (the reference's own configuration uses 10),
// so the struct-for blocks a brick overlaps are listed for every brick (at most (15 / blk + 2)^3 of them), the list is sorted, and a cell's
// place is (position of its block in the sorted list, cell inside the block) -- positions in a sorted superset of the active blocks order the
// cells exactly as the active blocks would.  For blocks of 16 this is the brick order by owner.
// =====================================================================================================================================
struct PoseTabS { const float* p; };
#define FSEQ_GP_SHIFT 44          // key: global pool brick (17 bits) | voxel (12) | block place and cell (29) | corner (3)
#define FSEQ_GL_SHIFT 32
struct FseqGeo { int blk, nca, cand, cellbits, nrx, nrz; };      // block edge, candidate blocks per axis and per brick, bits of a cell inside a block, blocks per axis of the field

__device__ __forceinline__ unsigned long long fseq_block_key(const FseqGeo& Q, int s, int ub, int vb, int wb)
{ return (((unsigned long long)s * Q.nrx + ub) * Q.nrx + vb) * Q.nrz + wb; }

// the struct-for blocks every source brick overlaps (~0 where the candidate lies beyond the brick or the field)
__global__ void __launch_bounds__(128) k_fseq_cand(MapDev S, FseqGeo Q, int nused, unsigned long long* __restrict__ cand)
{
    const int p = blockIdx.x;
    if (p >= nused) return;
    const int owner = S.owner[p];
    const int s = owner / S.nb3, b = owner - s * S.nb3;
    const int bk = b % S.nbz, bj = (b / S.nbz) % S.nbx, bi = b / (S.nbz * S.nbx);
    for (int t = threadIdx.x; t < Q.cand; t += 128) {
        const int cc = t % Q.nca, cb = (t / Q.nca) % Q.nca, ca = t / (Q.nca * Q.nca);
        const int ub = (bi * 16) / Q.blk + ca, vb = (bj * 16) / Q.blk + cb, wb = (bk * 16) / Q.blk + cc;
        const bool on = ub * Q.blk <= bi * 16 + 15 && vb * Q.blk <= bj * 16 + 15 && wb * Q.blk <= bk * 16 + 15 && ub < Q.nrx && vb < Q.nrx && wb < Q.nrz;
        cand[(size_t)p * Q.cand + t] = on ? fseq_block_key(Q, s, ub, vb, wb) : ~0ull;
    }
}
__device__ __forceinline__ int fseq_lower_bound(const unsigned long long* __restrict__ a, int n, unsigned long long key)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
// observed cells of the source bricks: sizes the tuple arrays (seven splats each at most)
__global__ void __launch_bounds__(256) k_fseq_count(MapDev S, int nused, unsigned long long* __restrict__ counter)
{
    long long n = 0;
    const uint32_t* o32 = reinterpret_cast<const uint32_t*>(S.obs);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)nused * (TSL_BRK3 / 4); i += (size_t)gridDim.x * 256) {
        const uint32_t v = o32[i];
        n += ((int8_t)v > 0) + ((int8_t)(v >> 8) > 0) + ((int8_t)(v >> 16) > 0) + ((int8_t)(v >> 24) > 0);
    }
    n = wave_sum_ll(n);
    if (lane_id() == 0 && n) __hip_atomic_fetch_add(counter + 1, (unsigned long long)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256) k_fseq_expand(MapDev S, MapDev G, PoseTabS poses, float vs, int nused, int npose, FseqGeo Q, const unsigned long long* __restrict__ csort, int ncand,
                                                     unsigned long long* tkeys, unsigned long long* tvals, unsigned long long cap, unsigned long long* counter)
{
    __shared__ int s_place[125];
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        const int owner = S.owner[p];
        const int s = owner / S.nb3, b = owner - s * S.nb3;
        if (s >= npose) continue;
        const float* Rp = poses.p + (size_t)s * 12;
        float R[9], T[3];
        for (int a = 0; a < 9; ++a) R[a] = Rp[a];
        for (int a = 0; a < 3; ++a) T[a] = Rp[9 + a];
        const int bk = b % S.nbz, bj = (b / S.nbz) % S.nbx, bi = b / (S.nbz * S.nbx);
        const int ub0 = (bi * 16) / Q.blk, vb0 = (bj * 16) / Q.blk, wb0 = (bk * 16) / Q.blk;
        __syncthreads();
        for (int t = threadIdx.x; t < Q.cand; t += 256) {
            const int cc = t % Q.nca, cb = (t / Q.nca) % Q.nca, ca = t / (Q.nca * Q.nca);
            s_place[t] = fseq_lower_bound(csort, ncand, fseq_block_key(Q, s, ub0 + ca, vb0 + cb, wb0 + cc));
        }
        __syncthreads();
        for (int l0 = 0; l0 < TSL_BRK3; l0 += 256) {
            const int l = l0 + (int)threadIdx.x;
            const size_t v = (size_t)p * TSL_BRK3 + l;
            const bool on = S.obs[v] > 0;                                                       // :292
            unsigned long long key[7], val[7]; int n = 0;
            if (on) {
                const int u = bi * 16 + (l >> 8), vv = bj * 16 + ((l >> 4) & 15), w = bk * 16 + (l & 15);      // 0-based cell of the field
                const int i = u - S.hN, j = vv - S.hN, k = w - S.hNz;
                const int ub = u / Q.blk, vb = vv / Q.blk, wb = w / Q.blk;
                const unsigned long long cell = (unsigned long long)(((u - ub * Q.blk) * Q.blk + (vv - vb * Q.blk)) * Q.blk + (w - wb * Q.blk));
                const unsigned long long place = ((unsigned long long)s_place[((ub - ub0) * Q.nca + (vb - vb0)) * Q.nca + (wb - wb0)] << Q.cellbits) | cell;
                const float p0 = (float)i * vs, p1 = (float)j * vs, p2 = (float)k * vs;
                float f[3]; int lo[3];
                for (int a = 0; a < 3; ++a) {
                    const float x = ((R[a * 3] * p0 + R[a * 3 + 1] * p1) + R[a * 3 + 2] * p2) + T[a];      // :293
                    f[a] = x / vs; lo[a] = (int)floorf(f[a]);                                     // :294-296
                }
                const uint32_t tw = S.tw[v];
                const float wsrc = h2f((h16)(tw >> 16));
                const uint32_t occ = (uint32_t)(uint8_t)S.occ[v];
                for (int c = 1; c < 8; ++c) {                                                    // :297-300 (corner 0 skipped), di, dj, dk in loop order
                    const int ci = lo[0] + ((c >> 2) & 1), cj = lo[1] + ((c >> 1) & 1), ck = lo[2] + (c & 1);
                    const float wt = ((1.0f - fabsf((float)ci - f[0])) * (1.0f - fabsf((float)cj - f[1]))) * (1.0f - fabsf((float)ck - f[2]));   // :303
                    const float w_tsdf = wsrc * wt;                                              // :307
                    if (!in_volume(G, ci, cj, ck)) continue;
                    int gl; const int gb = brick_of(G, ci, cj, ck, &gl);
                    const int gp = pool_claim<false>(G, 0, gb);
                    if (gp < 0) continue;
                    key[n] = ((unsigned long long)gp << FSEQ_GP_SHIFT) | ((unsigned long long)gl << FSEQ_GL_SHIFT) | (place << 3) | (unsigned long long)c;
                    val[n] = ((unsigned long long)__float_as_uint(w_tsdf) << 32) | ((unsigned long long)(tw & 0xffffu) << 8) | (unsigned long long)occ;
                    ++n;
                }
            }
            // wave-aggregated append (the order of the tuples in memory does not matter: the sort puts them in replay order)
            int inc = n;
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane_id() >= d) inc += o; }
            const int tot = __shfl(inc, 63);
            unsigned long long base = 0ull;
            if (tot) {
                if (lane_id() == 63) base = __hip_atomic_fetch_add(counter, (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                base = __shfl(base, 63);
            }
            const unsigned long long at = base + (unsigned long long)(inc - n);
            for (int t = 0; t < n; ++t) if (at + t < cap) { tkeys[at + t] = key[t]; tvals[at + t] = val[t]; }
        }
    }
}

// one thread per tuple; the head of a global voxel's run applies the whole run in order  (fuse_with_interploation :272-280)
template <bool TEX>
__global__ void __launch_bounds__(256) k_fseq_apply(MapDev S, MapDev G, const unsigned long long* __restrict__ tkeys, const unsigned long long* __restrict__ tvals,
                                                    FseqGeo Q, const unsigned long long* __restrict__ csort, long long total)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const unsigned long long k = tkeys[i];
    if (i != 0 && (tkeys[i - 1] >> FSEQ_GL_SHIFT) == (k >> FSEQ_GL_SHIFT)) return;
    const unsigned long long run = k >> FSEQ_GL_SHIFT;
    const size_t v = (size_t)(k >> FSEQ_GP_SHIFT) * TSL_BRK3 + (size_t)((k >> FSEQ_GL_SHIFT) & 4095ull);
    const uint32_t old = G.tw[v];
    h16 T0 = (h16)(old & 0xffffu), W0 = (h16)(old >> 16);
    int8_t occ = G.occ[v];
    h16 col[3] = { 0, 0, 0 };
    if (TEX) { const uint2 c = reinterpret_cast<const uint2*>(G.col)[v]; col[0] = (h16)(c.x & 0xffffu); col[1] = (h16)(c.x >> 16); col[2] = (h16)(c.y & 0xffffu); }
    for (long long q = i; q < total; ++q) {
        const unsigned long long kq = tkeys[q];
        if ((kq >> FSEQ_GL_SHIFT) != run) break;
        const unsigned long long tv = tvals[q];
        const float w_tsdf = __uint_as_float((uint32_t)(tv >> 32)), tsdf = h2f((h16)((tv >> 8) & 0xffffull));
        const float w_new = w_tsdf + h2f(W0);                                                                              // :273
        if (TEX) {                                                                                                          // :276-277 (with the old W)
            // the source cell, back from its place: block -> (submap, block coordinates), cell -> coordinates inside the block
            const unsigned long long place = (kq >> 3) & ((1ull << 29) - 1ull);
            unsigned long long bkey = csort[place >> Q.cellbits];
            const int cell = (int)(place & ((1ull << Q.cellbits) - 1ull));
            const int wb = (int)(bkey % Q.nrz); bkey /= Q.nrz;
            const int vb = (int)(bkey % Q.nrx); bkey /= Q.nrx;
            const int ub = (int)(bkey % Q.nrx); const int s = (int)(bkey / Q.nrx);
            const int u = ub * Q.blk + cell / (Q.blk * Q.blk), vv = vb * Q.blk + (cell / Q.blk) % Q.blk, w = wb * Q.blk + cell % Q.blk;
            const int sp = pool_lookup_ro(S, s, ((u >> 4) * S.nbx + (vv >> 4)) * S.nbz + (w >> 4));
            const size_t sv = (size_t)sp * TSL_BRK3 + (size_t)(((u & 15) << 8) | ((vv & 15) << 4) | (w & 15));
            const uint2 sc = reinterpret_cast<const uint2*>(S.col)[sv];
            const h16 c1[3] = { (h16)(sc.x & 0xffffu), (h16)(sc.x >> 16), (h16)(sc.y & 0xffffu) };
            for (int a = 0; a < 3; ++a) col[a] = f2h((h2f(hmul(W0, col[a])) + w_tsdf * h2f(c1[a])) / w_new);
        }
        T0 = f2h((h2f(hmul(W0, T0)) + w_tsdf * tsdf) / w_new);                                                             // :274
        W0 = f2h(w_new);                                                                                                    // :278
        occ = (int8_t)(occ + (int8_t)(tv & 0xffull));                                                                       // :280
    }
    G.tw[v] = (uint32_t)T0 | ((uint32_t)W0 << 16);
    G.obs[v] = 1;                                                                                                           // :279
    G.occ[v] = occ;
    if (TEX) reinterpret_cast<uint2*>(G.col)[v] = make_uint2((uint32_t)col[0] | ((uint32_t)col[1] << 16), (uint32_t)col[2]);
    G.touch[k >> FSEQ_GP_SHIFT] = 1;
}

// tsl_tsdf_fuse_submaps with semantics = 1 on the global map (called after the reset and the pose upload)
int fuse_submaps_sequential(tsl_tsdf* g, tsl_tsdf* sub, const float* pose_dev, int nsrc)
{
    TSL_REQUIRE(nsrc <= (1 << 17) && g->M.max_bricks <= (1 << 17), "sequential fusion: at most 2^17 bricks on either side");
    FseqGeo Q;
    Q.blk = sub->cfg.num_voxel_per_blk_axis;
    TSL_REQUIRE(Q.blk >= 4 && Q.blk <= 32, "sequential fusion: num_voxel_per_blk_axis of the submaps must be 4..32");
    Q.nca = 15 / Q.blk + 2; Q.cand = Q.nca * Q.nca * Q.nca;
    Q.cellbits = 0; while ((1 << Q.cellbits) < Q.blk * Q.blk * Q.blk) ++Q.cellbits;
    Q.nrx = sub->N / Q.blk; Q.nrz = sub->Nz / Q.blk;
    const long long ncand = (long long)nsrc * Q.cand;
    TSL_REQUIRE(ncand < (1ll << (29 - Q.cellbits)), "sequential fusion: too many source bricks for the replay key at this num_voxel_per_blk_axis");
    hipStream_t q = ms(g);
    (void)hipGetLastError();                        // (rocPRIM returns the thread's last error: a stale one is not this call's)
    int rc;
    if (!g->fseq_ctr) { if ((rc = dev_alloc(g, (void**)&g->fseq_ctr, 64, 0))) return rc; }
    // 0. how many cells splat: sizes the tuple arrays (a worst-case allocation was ~0.9 GB per 1000 source bricks: ADVICE r3)
    TSL_HIP(hipMemsetAsync(g->fseq_ctr, 0, 64, q));
    hipLaunchKernelGGL(k_fseq_count, dim3(1024), dim3(256), 0, q, sub->M, nsrc, (unsigned long long*)g->fseq_ctr);
    unsigned long long nobs = 0;
    TSL_HIP(hipMemcpyAsync(&nobs, (unsigned long long*)g->fseq_ctr + 1, 8, hipMemcpyDeviceToHost, q));
    TSL_HIP(hipStreamSynchronize(q));
    const unsigned long long cap = nobs * 7ull;
    const size_t cbytes = 8 * (size_t)ncand + 64;
    for (int k = 0; k < 2; ++k) {
        if ((rc = grow(&g->fseq_keys[k], &g->fseq_bytes[k], 8 * (size_t)cap + cbytes))) return rc;
        if ((rc = grow(&g->fseq_vals[k], &g->fseq_vbytes[k], 8 * (size_t)cap + cbytes))) return rc;
    }
    size_t ta = 0, tb = 0;
    TSL_HIP(rocprim::radix_sort_pairs(nullptr, ta, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)(cap ? cap : 1), 0u, 64u, q));
    TSL_HIP(rocprim::radix_sort_keys(nullptr, tb, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)ncand, 0u, 64u, q));
    if ((rc = grow(&g->fseq_temp, &g->fseq_tbytes, (ta > tb ? ta : tb) + 256))) return rc;
    // 1. the struct-for blocks the source bricks overlap, sorted: a block's position in the list is its place in the replay order.  The
    //    list must outlive the tuple buffers it shares memory with: it is kept behind them
    unsigned long long* cin = (unsigned long long*)((char*)g->fseq_keys[0] + 8 * (size_t)cap);
    unsigned long long* csort = (unsigned long long*)((char*)g->fseq_vals[0] + 8 * (size_t)cap);
    hipLaunchKernelGGL(k_fseq_cand, dim3(nsrc), dim3(128), 0, q, sub->M, Q, nsrc, cin);
    size_t tmp = g->fseq_tbytes;
    TSL_HIP(rocprim::radix_sort_keys(g->fseq_temp, tmp, cin, csort, (size_t)ncand, 0u, 64u, q));
    // 2. every splat as a tuple, 3. replay order per global voxel, 4. apply
    PoseTabS pt = { pose_dev };
    prof_begin(g, TSL_K_FUSE);
    hipLaunchKernelGGL(k_fseq_expand, dim3(nsrc < 8192 ? nsrc : 8192), dim3(256), 0, q, sub->M, g->M, pt, g->P.vs, nsrc, g->npose, Q, (const unsigned long long*)csort, (int)ncand,
                       (unsigned long long*)g->fseq_keys[0], (unsigned long long*)g->fseq_vals[0], cap, (unsigned long long*)g->fseq_ctr);
    unsigned long long count = 0;
    TSL_HIP(hipMemcpyAsync(&count, g->fseq_ctr, 8, hipMemcpyDeviceToHost, q));
    TSL_HIP(hipStreamSynchronize(q));
    TSL_REQUIRE(count <= cap, "sequential fusion: tuple buffer overflow");
    if (count) {
        tmp = g->fseq_tbytes;
        TSL_HIP(rocprim::radix_sort_pairs(g->fseq_temp, tmp, (unsigned long long*)g->fseq_keys[0], (unsigned long long*)g->fseq_keys[1], (unsigned long long*)g->fseq_vals[0],
                                          (unsigned long long*)g->fseq_vals[1], (size_t)count, 0u, 64u, q));
        const unsigned blocks = (unsigned)((count + 255) / 256);
        if (g->M.col && sub->M.col) hipLaunchKernelGGL(k_fseq_apply<true>, dim3(blocks), dim3(256), 0, q, sub->M, g->M, (const unsigned long long*)g->fseq_keys[1],
                                                        (const unsigned long long*)g->fseq_vals[1], Q, (const unsigned long long*)csort, (long long)count);
        else hipLaunchKernelGGL(k_fseq_apply<false>, dim3(blocks), dim3(256), 0, q, sub->M, g->M, (const unsigned long long*)g->fseq_keys[1],
                                (const unsigned long long*)g->fseq_vals[1], Q, (const unsigned long long*)csort, (long long)count);
    }
    prof_end(g);
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}

}  // namespace tsl
