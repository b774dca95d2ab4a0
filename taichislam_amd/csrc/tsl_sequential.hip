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

// =====================================================================================================================================
// Option "semantics" = 1 on a GLOBAL map: the reference-literal SEQUENTIAL fusion, fuse_submaps_kernel / fuse_with_interploation
// (dense_tsdf.py:272-318).  The reference walks every cell of every submap and, for seven of the eight surrounding global voxels, does an
// unsynchronised f16 read-modify-write of the running weighted average (:274-280); splats race.  The sequential schedule -- submap cells in
// struct-for order (submap, brick lexicographic, cell row-major), the seven corners in loop order -- is what the CPU checker's FAITHFUL fusion
// and tools/ti_seq execute.  Here: every splat becomes a tuple  key = global brick | global voxel | sequence number (rank of the source brick
// among the submaps' bricks by owner, cell, corner),  value = { w_tsdf, tsdf, occupancy };  radix sort;  one thread per global voxel applies
// its run in order.  (The default fusion, tsl_fuse.hip, sums the same terms exactly and divides once: order-free, the multi-GPU merge rests on it.)
// =====================================================================================================================================
struct PoseTabS { const float* p; };
#define FSEQ_GP_SHIFT 44          // key: global pool brick (17 bits) | voxel (12) | source brick rank (17) | source cell (12) | corner (3)
#define FSEQ_GL_SHIFT 32

__global__ void __launch_bounds__(256) k_fseq_order(MapDev S, int nused, unsigned long long* keys, uint32_t* vals)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < nused) { keys[p] = (unsigned long long)(uint32_t)S.owner[p]; vals[p] = (uint32_t)p; }
}

__global__ void __launch_bounds__(256) k_fseq_expand(MapDev S, MapDev G, PoseTabS poses, float vs, int nused, int npose, const uint32_t* __restrict__ brick_of_rank,
                                                     unsigned long long* tkeys, unsigned long long* tvals, unsigned long long cap, unsigned long long* counter)
{
    for (int r = blockIdx.x; r < nused; r += gridDim.x) {
        const int p = (int)brick_of_rank[r];
        const int owner = S.owner[p];
        const int s = owner / S.nb3, b = owner - s * S.nb3;
        if (s >= npose) continue;
        const float* Rp = poses.p + (size_t)s * 12;
        float R[9], T[3];
        for (int a = 0; a < 9; ++a) R[a] = Rp[a];
        for (int a = 0; a < 3; ++a) T[a] = Rp[9 + a];
        const int bk = b % S.nbz, bj = (b / S.nbz) % S.nbx, bi = b / (S.nbz * S.nbx);
        for (int l0 = 0; l0 < TSL_BRK3; l0 += 256) {
            const int l = l0 + (int)threadIdx.x;
            const size_t v = (size_t)p * TSL_BRK3 + l;
            const bool on = S.obs[v] > 0;                                                       // :292
            unsigned long long key[7], val[7]; int n = 0;
            if (on) {
                const int i = bi * 16 + (l >> 8) - S.hN, j = bj * 16 + ((l >> 4) & 15) - S.hN, k = bk * 16 + (l & 15) - S.hNz;
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
                    key[n] = ((unsigned long long)gp << FSEQ_GP_SHIFT) | ((unsigned long long)gl << FSEQ_GL_SHIFT) | ((unsigned long long)r << 15) | ((unsigned long long)l << 3) | (unsigned long long)c;
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
                                                    const uint32_t* __restrict__ brick_of_rank, long long total)
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
            const size_t sv = (size_t)brick_of_rank[(kq >> 15) & 0x1ffffull] * TSL_BRK3 + (size_t)((kq >> 3) & 4095ull);
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
    hipStream_t q = ms(g);
    int rc;
    const unsigned long long cap = (unsigned long long)nsrc * TSL_BRK3 * 7ull;
    for (int k = 0; k < 2; ++k) {
        if ((rc = grow(&g->fseq_keys[k], &g->fseq_bytes[k], 8 * (size_t)cap + 8 * (size_t)nsrc + 64))) return rc;
        if ((rc = grow(&g->fseq_vals[k], &g->fseq_vbytes[k], 8 * (size_t)cap + 8 * (size_t)nsrc + 64))) return rc;
    }
    size_t ta = 0, tb = 0;
    TSL_HIP(rocprim::radix_sort_pairs(nullptr, ta, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (size_t)cap, 0u, 64u, q));
    TSL_HIP(rocprim::radix_sort_pairs(nullptr, tb, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)nsrc, 0u, 32u, q));
    if ((rc = grow(&g->fseq_temp, &g->fseq_tbytes, (ta > tb ? ta : tb) + 256))) return rc;
    if (!g->fseq_ctr) { if ((rc = dev_alloc(g, (void**)&g->fseq_ctr, 64, 0))) return rc; }
    // 1. the submaps' bricks in struct-for order: by owner = submap * bricks per submap + brick index
    unsigned long long* bk = (unsigned long long*)g->fseq_keys[0]; unsigned long long* bk_s = (unsigned long long*)g->fseq_keys[1];
    uint32_t* bv = (uint32_t*)g->fseq_vals[0]; uint32_t* bv_s = (uint32_t*)g->fseq_vals[1];
    hipLaunchKernelGGL(k_fseq_order, dim3((nsrc + 255) / 256), dim3(256), 0, q, sub->M, nsrc, bk, bv);
    size_t tmp = g->fseq_tbytes;
    TSL_HIP(rocprim::radix_sort_pairs(g->fseq_temp, tmp, bk, bk_s, bv, bv_s, (size_t)nsrc, 0u, 32u, q));
    // the rank -> brick table must outlive the tuple buffers it shares memory with: it is kept behind them
    uint32_t* rank_tab = (uint32_t*)((char*)g->fseq_vals[0] + 8 * (size_t)cap);
    TSL_HIP(hipMemcpyAsync(rank_tab, bv_s, 4 * (size_t)nsrc, hipMemcpyDeviceToDevice, q));
    // 2. every splat as a tuple, 3. replay order per global voxel, 4. apply
    TSL_HIP(hipMemsetAsync(g->fseq_ctr, 0, 64, q));
    PoseTabS pt = { pose_dev };
    prof_begin(g, TSL_K_FUSE);
    hipLaunchKernelGGL(k_fseq_expand, dim3(nsrc < 8192 ? nsrc : 8192), dim3(256), 0, q, sub->M, g->M, pt, g->P.vs, nsrc, g->npose, (const uint32_t*)rank_tab,
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
                                                        (const unsigned long long*)g->fseq_vals[1], (const uint32_t*)rank_tab, (long long)count);
        else hipLaunchKernelGGL(k_fseq_apply<false>, dim3(blocks), dim3(256), 0, q, sub->M, g->M, (const unsigned long long*)g->fseq_keys[1],
                                (const unsigned long long*)g->fseq_vals[1], (const uint32_t*)rank_tab, (long long)count);
    }
    prof_end(g);
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}

}  // namespace tsl
