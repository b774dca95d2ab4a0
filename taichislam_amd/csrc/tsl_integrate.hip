// tsl_integrate.hip -- the dominant kernel family: per-voxel weighted TSDF update of one frame's rays.
// Replaces process_new_pcl (taichi_slam/mapping/dense_tsdf.py:236-270, reference root).
//
// Two interchangeable strategies, bit-identical results (exact int64 fixed-point sums):
//
//   variant 0/1  "global atomics": every ray-step adds {w*sd, w} with two no-return int64 atomics into a
//                per-frame brick scratch in HBM/L2, k_finalize applies the sums.  Priced on MI355X at
//                ~11.7 G pairs/s scattered and ~12 ns per lane for same-address hits (tools/ubench/atomics.hip):
//                ~1 ms/frame at BASELINE configs[1].  Kept as the simple cross-check.
//
//   variant 2    "brick-binned LDS" (default): all updates of one 16^3 brick happen in LDS.
//                phase A (per frame, on the stream of its working set, independent of the map contents):
//                k_segments   : every ray is cut into runs of consecutive steps inside one brick ("segments", one u64
//                               each) by searching the brick-boundary crossings per axis; with the hash grouping of
//                               pixels the rays themselves are built here too.  Segments go to private slots of the ray,
//                               per-brick counts to an LDS hash that is flushed once per workgroup.
//                k_plan       : segment range and integrate parts of every active brick (block-aggregated reservations).
//                k_scatter    : counting-sort the segments by brick (LDS hash per 4096-slot tile, one global
//                               reservation per (tile, brick)).
//                phase B (frame order, main stream):
//                k_integrate_bricks : one workgroup per brick part (<= 1024 segments); the brick's 4096 {num,den}
//                               int64 accumulators live in 64 KiB of LDS (ds_add_u64, >400 G pairs/s chip-wide), then
//                               the brick is finalised in place with all row loads in flight.  Bricks split over several
//                               workgroups add their partial sums to an HBM slab; the last workgroup to arrive (ticket)
//                               finalises from there.
#include "tsl_tsdf.hpp"
#include <hip/hip_ext.h>

namespace tsl {

#ifdef TSL_TIMING
// developer timing: every wave stores raw timestamps (plain stores, no atomics) at dbg[wave*16 + k]
#define TSL_T0() const int _wv = (blockIdx.x * (int)blockDim.x + (int)threadIdx.x) >> 6
#define TSL_TICK(F, k) do { long long _n = wall_clock64(); if (lane_id() == 0 && _wv < 16384) (F).dbg[_wv * 16 + (k)] = _n; } while (0)
#else
#define TSL_T0() do {} while (0)
#define TSL_TICK(F, k) do {} while (0)
#endif

// (the segment key layout, the header words and the work-list classes are in tsl_tsdf.hpp: tsl_sequential.hip reads the same tables)
#define SCATTER_TILE 4096

struct RayRegs { float pf0, pf1, pf2, d0, d1, d2, P0, P1, P2, w; long long qden; int n; };

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    for (int d = 32; d > 0; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, d); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ RayRegs make_ray(const uint4 rec, int n, const FrameParams& P)
{
    RayRegs R;
    R.n = n;
    R.pf0 = h2f((h16)(rec.x & 0xffffu)); R.pf1 = h2f((h16)(rec.x >> 16)); R.pf2 = h2f((h16)(rec.y & 0xffffu));
    R.d0 = h2f((h16)(rec.y >> 16)); R.d1 = h2f((h16)(rec.z & 0xffffu)); R.d2 = h2f((h16)(rec.z >> 16));
    R.w = __uint_as_float(rec.w);
    R.qden = to_fix(R.w);
    R.P0 = R.pf0 + P.T[0]; R.P1 = R.pf1 + P.T[1]; R.P2 = R.pf2 + P.T[2];                      // dense_tsdf.py:246
    return R;
}
template <bool WITH_N = true>
__device__ __forceinline__ RayRegs load_ray(const FrameDev& F, const FrameParams& P, int r) { return make_ray(F.rayA[r], WITH_N ? F.rayN[r] : 0, P); }
// voxel visited at step j  (dense_tsdf.py:253-254)
__device__ __forceinline__ void step_voxel(const RayRegs& R, const FrameParams& P, int j, float* x, int* xi)
{
    const float jf = (float)j;
    x[0] = (R.d0 * jf) * P.vs + P.T[0]; x[1] = (R.d1 * jf) * P.vs + P.T[1]; x[2] = (R.d2 * jf) * P.vs + P.T[2];
    xi[0] = rnd_i(div_vs(x[0], P.vs, P.rvs, P.fastdiv)); xi[1] = rnd_i(div_vs(x[1], P.vs, P.rvs, P.fastdiv)); xi[2] = rnd_i(div_vs(x[2], P.vs, P.rvs, P.fastdiv));
}
// numerator term of the running average  (dense_tsdf.py:258-264)
__device__ __forceinline__ long long step_term(const RayRegs& R, const float* x)
{
    const float v0 = R.P0 - x[0], v1 = R.P1 - x[1], v2 = R.P2 - x[2];
    const float s2 = (v0 * v0 + v1 * v1) + v2 * v2;
    const float dist = s2 >= 1.2621774483536189e-29f ? sqrt_rn_norm(s2) : sqrt_rn(s2);        // 2^-96: below it sqrtf rescales
    const float dot = (v0 * R.pf0 + v1 * R.pf1) + v2 * R.pf2;
    const float sd = dist * (float)sgn_f(dot);
    return to_fix_wave(R.w * sd);
}
// occupy[pos_p] = 1  (dense_tsdf.py:248)
template <bool COOP = true>
__device__ __forceinline__ void mark_occupied(const MapDev& M, const FrameParams& P, const RayRegs& R)
{
    const int oi = rnd_i(div_vs(R.P0, P.vs, P.rvs, P.fastdiv)), oj = rnd_i(div_vs(R.P1, P.vs, P.rvs, P.fastdiv)), ok = rnd_i(div_vs(R.P2, P.vs, P.rvs, P.fastdiv));
    if (in_volume(M, oi, oj, ok)) {
        int l; const int b = brick_of(M, oi, oj, ok, &l);
        const int p = pool_claim<COOP>(M, P.slot, b);
        if (p >= 0) M.occ[(size_t)p * TSL_BRK3 + l] = 1;
    }
}
// frame scratch slot of a brick (allocating the brick and the slot on first touch); < 0 when out of capacity.
// Fast path: one plain, L1-cacheable load of the per-frame table (entries only go EMPTY -> slot inside a frame
// and are reset by k_finalize, so a non-negative value is never stale); slow path: wave-cooperative claim.
template <bool COOP = true>
__device__ __forceinline__ int frame_slot_slow(const MapDev& M, const FrameDev& F, int s, int b)
{
    const int sl = COOP ? claim_index(F.slot_tab + b, &F.counters[1], F.max_frame_bricks) : claim_index_1(F.slot_tab + b, &F.counters[1], F.max_frame_bricks);
    if (sl >= 0) {
        const int p = pool_claim<COOP>(M, s, b);
        if (p >= 0) { F.touched[sl] = p; F.touched_b[sl] = b; }      // idempotent: every claimer writes the same values
        else return -1;
    } else frame_fail(M, F, 2);
    return sl;
}
__device__ __forceinline__ int frame_slot(const MapDev& M, const FrameDev& F, int s, int b)
{
    const int sl = F.slot_tab[b];
    return sl >= 0 ? sl : frame_slot_slow(M, F, s, b);
}

// apply one frame's sums to a voxel  (dense_tsdf.py:264-267 with the frame's total weight)
__device__ __forceinline__ uint32_t apply_update(uint32_t old, long long qnum, long long qden)
{
    const float num = from_fix(qnum), den = from_fix(qden);
    const h16 T0 = (h16)(old & 0xffffu), W0 = (h16)(old >> 16);
    const h16 Tn = f2h((h2f(hmul(T0, W0)) + num) / (h2f(W0) + den));
    float wn = h2f(W0) + den; if (TSL_WMAX < wn) wn = TSL_WMAX;
    return (uint32_t)Tn | ((uint32_t)f2h(wn) << 16);
}
// the same with the sums already converted (from_fix / from_fix32)
__device__ __forceinline__ uint32_t apply_update_f(uint32_t old, float num, float den)
{
    const h16 T0 = (h16)(old & 0xffffu), W0 = (h16)(old >> 16);
    const h16 Tn = f2h((h2f(hmul(T0, W0)) + num) / (h2f(W0) + den));
    float wn = h2f(W0) + den; if (TSL_WMAX < wn) wn = TSL_WMAX;
    return (uint32_t)Tn | ((uint32_t)f2h(wn) << 16);
}

// =====================================================================================================
// variant 0/1: global int64 atomics
// =====================================================================================================
template <int VARIANT>
__global__ void __launch_bounds__(256) k_integrate(MapDev M, FrameDev F, const FrameParams* __restrict__ Pp)
{
    const FrameParams& P = *Pp;
    const int split = P.split;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int r = gid / split, sub = gid - r * split;
    const int nrays = *F.nrays;
    const bool live = r < nrays;
    long long n_ok = 0, n_oob = 0;
    RayRegs R; R.n = 0; R.qden = 0;
    if (live) { R = load_ray(F, P, r); if (sub == 0) mark_occupied(M, P, R); }
    int cur_b = -1; unsigned long long* cur_acc = nullptr;
    int nmax = R.n;
    if (VARIANT == 1) { for (int d = 32; d > 0; d >>= 1) { int o = __shfl_xor(nmax, d); nmax = o > nmax ? o : nmax; } }
    const int iters = (nmax + split - 1) / split;          // wave-uniform trip count when VARIANT == 1 (cross-lane ops inside)
    for (int it = 0; it < iters; ++it) {
        const int j = 1 + sub + it * split;
        const bool act = live && j <= R.n;
        unsigned long long* dst = nullptr;
        long long qn = 0;
        if (act) {
            float x[3]; int xi[3];
            step_voxel(R, P, j, x, xi);
            if (in_volume(M, xi[0], xi[1], xi[2])) {
                qn = step_term(R, x);
                int l; const int b = brick_of(M, xi[0], xi[1], xi[2], &l);
                if (b != cur_b) { cur_b = b; const int sl = frame_slot(M, F, P.slot, b); cur_acc = sl >= 0 ? F.acc + (size_t)sl * (TSL_BRK3 * 2) : nullptr; }
                if (cur_acc) { dst = cur_acc + (size_t)l * 2; ++n_ok; }
            } else ++n_oob;
        }
        if (VARIANT == 1) {     // wave-uniform fast path: all live lanes hit one voxel -> reduce in-wave, one atomic pair
            const unsigned long long m = __ballot(dst != nullptr);
            if (m) {
                const int leader = (int)__builtin_ctzll(m);
                const unsigned long long lead = __shfl((unsigned long long)dst, leader);
                const bool same = (dst == nullptr) || ((unsigned long long)dst == lead);
                if (__all(same)) {
                    const long long sn = wave_sum_ll(dst ? qn : 0), sdn = wave_sum_ll(dst ? R.qden : 0);
                    if (lane_id() == leader) {
                        __hip_atomic_fetch_add(dst, (unsigned long long)sn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_add(dst + 1, (unsigned long long)sdn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    continue;
                }
            }
        }
        if (dst) {
            __hip_atomic_fetch_add(dst, (unsigned long long)qn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(dst + 1, (unsigned long long)R.qden, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    n_ok = wave_sum_ll(n_ok); n_oob = wave_sum_ll(n_oob);
    if (lane_id() == 0) {
        if (n_ok) atomic_add_i64(&F.stats->steps, n_ok);
        if (n_oob) atomic_add_i64(&F.stats->steps_oob, n_oob);
    }
}

// finalise bricks from the HBM scratch.  list == nullptr: every touched slot; else the `*nlist` slots in list.
// Always releases the frame slots of all touched bricks (slot_of_pool) for the next frame.
__global__ void __launch_bounds__(256) k_finalize(MapDev M, FrameDev F, const int* list, const int* nlist)
{
    const int ntouched = min(F.counters[1], F.max_frame_bricks);
    const int nwork = list ? min(*nlist, F.max_frame_bricks) : ntouched;
    long long uniq = 0;
    for (int q = blockIdx.x; q < nwork; q += gridDim.x) {
        const int sl = list ? list[q] : q;
        const int p = F.touched[sl];
        if (threadIdx.x == 0) M.touch[p] = 1;
        ulonglong2* acc = reinterpret_cast<ulonglong2*>(F.acc + (size_t)sl * (TSL_BRK3 * 2));
        uint32_t* tw = M.tw + (size_t)p * TSL_BRK3;
        int8_t* obs = M.obs + (size_t)p * TSL_BRK3;
        ulonglong2 a[TSL_BRK3 / 256]; uint32_t old[TSL_BRK3 / 256];
#pragma unroll
        for (int q = 0; q < TSL_BRK3 / 256; ++q) { a[q] = acc[q * 256 + threadIdx.x]; old[q] = tw[q * 256 + threadIdx.x]; }
#pragma unroll
        for (int q = 0; q < TSL_BRK3 / 256; ++q) {
            const int l = q * 256 + threadIdx.x;
            if (a[q].y != 0ull) {
                tw[l] = apply_update(old[q], (long long)a[q].x, (long long)a[q].y);
                obs[l] = 1;                                                                      // dense_tsdf.py:265
                acc[l] = make_ulonglong2(0ull, 0ull);
                ++uniq;
            }
        }
    }
    for (int sl = blockIdx.x * 256 + threadIdx.x; sl < ntouched; sl += gridDim.x * 256) F.slot_tab[F.touched_b[sl]] = TSL_EMPTY;
    uniq = wave_sum_ll(uniq);
    if (lane_id() == 0 && uniq) atomic_add_i64(&F.stats->unique, uniq);
    if (!list && blockIdx.x == 0 && threadIdx.x == 0) F.stats->bricks = ntouched;
}

// =====================================================================================================
// variant 2: brick-binned segments, LDS accumulation
// =====================================================================================================
// exact voxel coordinate of step j along one axis  (dense_tsdf.py:253-254, one component)
__device__ __forceinline__ int axis_coord(float d, float T, const FrameParams& P, int j) { return rnd_i(div_vs((d * (float)j) * P.vs + T, P.vs, P.rvs, P.fastdiv)); }
// "cell" of a coordinate along one axis: -1 below the volume, nb above it, else the brick coordinate.  Monotone in c.
__device__ __forceinline__ int axis_cell(int c, int h, int N, int nb) { const int u = c + h; return u < 0 ? -1 : (u >= N ? nb : (u >> 4)); }

// Smallest step e in (j, jb] at which the ray's cell along one axis differs from `cell` (the cell at step j), or jb+1.
// The coordinate is a monotone function of the step (every operation in axis_coord is monotone), so the event is
// located from a real-arithmetic estimate and then fixed up with exact evaluations -- typically two.
__device__ __forceinline__ int next_axis_event(float d, float inv_d, float T, const FrameParams& P, float t_over_vs, int j, int jb, int cell, int h, int N, int nb)
{
    if (j >= jb) return jb + 1;
    int bound;                       // first coordinate that belongs to the next cell in the direction of travel
    bool up;
    if (d > 0.0f) { if (cell >= nb) return jb + 1; up = true; bound = (cell < 0 ? 0 : min((cell + 1) * 16, N)) - h; }
    else if (d < 0.0f) { if (cell < 0) return jb + 1; up = false; bound = (cell >= nb ? N - 1 : cell * 16 - 1) - h; }
    else return jb + 1;
    const float est = ((float)bound + (up ? -0.5f : 0.5f) - t_over_vs) * inv_d;  // real-valued crossing step; only an estimate (inv_d ~ 1/d): the exact evaluations below decide
    int e = (int)fminf(fmaxf(ceilf(est), (float)(j + 1)), (float)jb);
    #define TSL_PRED(q) (up ? (axis_coord(d, T, P, (q)) >= bound) : (axis_coord(d, T, P, (q)) <= bound))
    if (TSL_PRED(e)) { while (e - 1 > j && TSL_PRED(e - 1)) --e; }
    else { ++e; while (e <= jb && !TSL_PRED(e)) ++e; }
    #undef TSL_PRED
    return e;
}

// ---- small open-addressing hash in LDS: brick id -> counter (a block / tile only touches ~100 distinct bricks) ----
#define LH_SIZE 1024
#define LH_EMPTY (-1)
template <int LOG2 = 10>
__device__ __forceinline__ int lh_slot(int* keys, int b)         // find-or-insert; returns the table index or -1 when full
{
    unsigned h = ((unsigned)b * 2654435761u) >> (32 - LOG2);
    for (int probe = 0; probe < (1 << LOG2); ++probe) {
        const int k = keys[h];
        if (k == b) return (int)h;
        if (k == LH_EMPTY) { const int old = atomicCAS(&keys[h], LH_EMPTY, b); if (old == LH_EMPTY || old == b) return (int)h; }
        h = (h + 1) & ((1u << LOG2) - 1u);
    }
    return -1;
}

// K4a: cut rays into per-brick segments without visiting every step: per axis the next brick/volume boundary
// crossing is searched directly (next_axis_event), so a ray costs ~12 crossings x ~2 exact coordinate evaluations
// instead of ~135 full voxel evaluations.  Segments are keyed by BRICK id (no per-frame slot claims anywhere: the
// dense renumbering of the frame's bricks is the order in which they are listed); per-brick counts are kept in an LDS hash and flushed
// once per block.
// segment: [0,6) count [6,18) first step [18,40) ray [40,64) brick id
#define SEG_RAY_SLOTS 16        // private segment slots per ray (split evenly over its lanes); further segments are appended behind them
#define SEG_LH_LOG2 9
#define SEG_LH (1 << SEG_LH_LOG2)
// FUSED: the rays are built here as well -- lane 0 of every ray replays the pixels of its sensor voxel in raster order
// (dense_tsdf.py:230-249; crowded voxels by the whole wave) and hands the ray to the other lanes; ray id = position of the
// sensor voxel in the frame's list.
template <bool FUSED>
__global__ void __launch_bounds__(256) k_segments(MapDev M, BatchDev B)
{
    if ((int)blockIdx.y >= B.n) return;
    const FrameDev& F = B.f[blockIdx.y];
    const FrameParams& P = *B.p[blockIdx.y];
    __shared__ int s_key[SEG_LH];
    __shared__ int s_cnt[SEG_LH];
    TSL_T0();
    TSL_TICK(F, 8);
    for (int i = threadIdx.x; i < SEG_LH; i += 256) { s_key[i] = LH_EMPTY; s_cnt[i] = 0; }
    __syncthreads();
    TSL_TICK(F, 0);

    const int split = P.split;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int r = gid / split, sub = gid - r * split;
    const int nrays = FUSED ? F.counters[6] : *F.nrays;
    long long n_ok = 0, n_oob = 0;
    const bool working = __any(r < nrays);
    const int spl = SEG_RAY_SLOTS / split;                              // private slots of this lane in the frame's segment array
    unsigned long long* myseg = F.seg + (size_t)gid * spl;
    int nslot = 0;
    bool ok = r < nrays;
    uint4 rec = make_uint4(0, 0, 0, 0);
    int nsteps = 0;
    if (FUSED) {
        const int lane = threadIdx.x & 63;
        bool big = false;
        int gn = 0, sl = 0;
        uint32_t first = 0;
        unsigned long long vkey = 0ull;
        ok = false;
        if (r < nrays && sub == 0) {
            sl = F.act[r];
            // the voxel's slot: key, count and the ids of its first H_INL pixels -- one 64-byte line
            const uint4* sp = reinterpret_cast<const uint4*>(F.htab + sl);
            const uint4 w0 = sp[0], w1 = sp[1], w2 = sp[2], w3 = sp[3];
            vkey = ((unsigned long long)w0.x | ((unsigned long long)w0.y << 32)) - 1ull;      // the slot holds h_key(voxel key, 0)
            gn = (int)w0.z;
            if (gn > H_INL) big = true;
            else {
                uint32_t id[H_INL] = { w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w };
                uint2 pl[H_INL];
#pragma unroll
                for (int q = 0; q < H_INL; ++q) {                                // all payloads are requested together, then replayed from registers
                    if (q >= gn) id[q] = 0xffffffffu;
                    pl[q] = q < gn ? F.pix[id[q]] : make_uint2(0u, 0u);
                }
                PixAcc A = {};
                for (int k = 0; k < gn; ++k) {                                   // next pixel in raster order = smallest id not yet taken
                    uint32_t best = 0xffffffffu; uint2 bp = make_uint2(0u, 0u);
#pragma unroll
                    for (int q = 0; q < H_INL; ++q) { const bool t = id[q] < best; best = t ? id[q] : best; bp.x = t ? pl[q].x : bp.x; bp.y = t ? pl[q].y : bp.y; }
#pragma unroll
                    for (int q = 0; q < H_INL; ++q) id[q] = id[q] == best ? 0xffffffffu : id[q];
                    if (k == 0) first = best;
                    if (P.tex) acc_colour(P, F, best, A);
                    acc_payload(bp, A);
                }
                ok = finish_ray(P, F, A, first, &rec, &nsteps);
            }
        }
        // crowded voxels (more than H_INL pixels): the whole wave replays them.  Pixel ranks b * H_INL .. of the voxel live in the slot keyed
        // (voxel, block b); lane b resolves block b (b, b + 64, ...), every selection step takes the smallest id above the last one.
        for (unsigned long long bm = __ballot(big); bm; bm &= bm - 1ull) {
            const int src = (int)__builtin_ctzll(bm);
            const int n = __shfl(gn, src), sl0 = __shfl(sl, src);
            const unsigned long long vk = ((unsigned long long)(uint32_t)__shfl((int)(vkey >> 32), src) << 32) | (unsigned long long)(uint32_t)__shfl((int)(uint32_t)vkey, src);
            const int nblk = (n + H_INL - 1) / H_INL;
            const uint32_t hmask = (1u << P.hlog2) - 1u;
            auto find_block = [&](int bq) -> int {
                if (bq == 0) return sl0;
                const unsigned long long k2 = h_key(vk, bq);
                uint32_t h = h_hash64(k2, P.hlog2);
                for (uint32_t probe = 0; probe <= hmask; ++probe) {
                    const unsigned long long cur = F.htab[h].key;
                    if (cur == k2) return (int)h;
                    if (cur == H_EMPTY) return -1;
                    h = (h + 1u) & hmask;
                }
                return -1;
            };
            const int mine = lane < nblk ? find_block(lane) : -1;                // the block this lane reads in every selection step
            const bool too_many = n > GROUP_BIG_CAP;
            if (too_many && lane == 0) atomicOr(M.err, 8);
            PixAcc A = {};
            long long last = -1; uint32_t f0 = 0;
            for (int k = 0; k < n && !too_many; ++k) {
                uint32_t best = 0xffffffffu;
                for (int bq = lane; bq < nblk; bq += 64) {
                    const int hs = bq == lane ? mine : find_block(bq);
                    if (hs < 0) continue;
                    const int cb = min(H_INL, n - bq * H_INL);
                    const uint32_t* ids = F.htab[hs].pix;
                    for (int q = 0; q < cb; ++q) { const uint32_t v = ids[q]; if ((long long)v > last && v < best) best = v; }
                }
                best = wave_min_u32(best);
                if (best == 0xffffffffu) break;                                  // (a block went missing: cannot happen, the ray is dropped below)
                if (k == 0) f0 = best;
                acc_pixel(P, F, best, A);
                last = (long long)best;
            }
            uint4 rc; int ns = 0;
            const bool k2 = A.cnt == n && finish_ray(P, F, A, f0, &rc, &ns, lane == src);
            if (lane == src) { rec = rc; nsteps = ns; ok = k2; first = f0; }
        }
        if (r < nrays && sub == 0) {
            F.rayA[r] = rec; F.rayFirst[r] = first;
            if (P.seq) {      // sequential semantics: the ray's step count and its place in Taichi's struct-for order over the sensor grid
                // (pointer block lexicographic, then dense cell: assumption A6 of SURVEY.md section 8c) -- the Morton key is taken apart again
                uint32_t u[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    unsigned long long x = (vkey >> (2 - a)) & 0x1249249249249249ull;
                    x = (x | (x >> 2)) & 0x10c30c30c30c30c3ull; x = (x | (x >> 4)) & 0x100f00f00f00f00full; x = (x | (x >> 8)) & 0x1f0000ff0000ffull;
                    x = (x | (x >> 16)) & 0x1f00000000ffffull; x = (x | (x >> 32)) & 0x1fffffull;
                    u[a] = (uint32_t)x;
                }
                const unsigned long long nblk = (unsigned long long)(P.pcl_ext / P.pcl_blk), blk = (unsigned long long)P.pcl_blk;
                const unsigned long long kb = ((u[0] / P.pcl_blk) * nblk + (u[1] / P.pcl_blk)) * nblk + (u[2] / P.pcl_blk);
                const unsigned long long kl = ((u[0] % P.pcl_blk) * blk + (u[1] % P.pcl_blk)) * blk + (u[2] % P.pcl_blk);
                reinterpret_cast<unsigned long long*>(F.keys)[r] = kb * blk * blk * blk + kl;
                F.rayN[r] = ok ? nsteps : 0;
            }
        }
        block_count_add(&F.stats->v_pcl, r < nrays && sub == 0);
        block_count_add(&F.stats->v_skipped, r < nrays && sub == 0 && !ok);
        if (gid == 0) *F.nrays = nrays;
        // hand the ray to the other lanes of its group (a group never straddles a wave: split divides 64)
        const int src = lane - sub;
        rec.x = (uint32_t)__shfl((int)rec.x, src); rec.y = (uint32_t)__shfl((int)rec.y, src); rec.z = (uint32_t)__shfl((int)rec.z, src); rec.w = (uint32_t)__shfl((int)rec.w, src);
        nsteps = __shfl(nsteps, src);
        ok = __shfl((int)ok, src) != 0;
    }
    if (ok) {
        const RayRegs R = FUSED ? make_ray(rec, nsteps, P) : load_ray(F, P, r);
        if (sub == 0) mark_occupied(M, P, R);
        TSL_TICK(F, 1);
        const int len = (R.n + split - 1) / split;
        const int ja = 1 + sub * len, jb = min(R.n, ja + len - 1);
        const float d[3] = { R.d0, R.d1, R.d2 };
        const float id[3] = { __builtin_amdgcn_rcpf(R.d0), __builtin_amdgcn_rcpf(R.d1), __builtin_amdgcn_rcpf(R.d2) };
        const float tv[3] = { P.T[0] / P.vs, P.T[1] / P.vs, P.T[2] / P.vs };
        const int hh[3] = { M.hN, M.hN, M.hNz }, NN[3] = { M.N, M.N, M.Nz }, nb[3] = { M.nbx, M.nbx, M.nbz };
        if (ja <= jb) {
            int cell[3], ev[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                cell[a] = axis_cell(axis_coord(d[a], P.T[a], P, ja), hh[a], NN[a], nb[a]);
                ev[a] = next_axis_event(d[a], id[a], P.T[a], P, tv[a], ja, jb, cell[a], hh[a], NN[a], nb[a]);
            }
            int j = ja;
            while (j <= jb) {
                const int e = min(min(ev[0], ev[1]), ev[2]);                 // first step of the next run (or jb+1)
                const bool inside = cell[0] >= 0 && cell[0] < nb[0] && cell[1] >= 0 && cell[1] < nb[1] && cell[2] >= 0 && cell[2] < nb[2];
                if (inside) {
                    const int b = (cell[0] * M.nbx + cell[1]) * M.nbz + cell[2];
                    n_ok += e - j;
                    for (int j0 = j; j0 < e; j0 += SEG_MAX_CNT) {
                        const int cnt = min(e - j0, SEG_MAX_CNT);
                        const unsigned long long en = ((unsigned long long)b << STG_B_SHIFT) | ((unsigned long long)r << (SEG_CNT_BITS + SEG_J_BITS)) |
                                                      ((unsigned long long)j0 << SEG_CNT_BITS) | (unsigned long long)cnt;
                        bool stored = true;
                        if (nslot < spl) myseg[nslot++] = en;
                        else {      // more brick crossings than private slots: append behind the per-ray slots
                            const long long pos = (long long)nrays * SEG_RAY_SLOTS + __hip_atomic_fetch_add(&F.counters[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (pos < F.seg_cap) F.seg[pos] = en; else { stored = false; frame_fail(M, F, 4); }
                        }
                        if (stored) {
                            const int hs = lh_slot<SEG_LH_LOG2>(s_key, b);
                            if (hs >= 0) atomicAdd(&s_cnt[hs], 1);
                            else if (__hip_atomic_fetch_add(&F.bhist[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {      // hash full (incoherent rays)
                                const int q = __hip_atomic_fetch_add(&F.counters[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (q < F.max_frame_bricks) F.act_b[q] = b; else frame_fail(M, F, 2);
                            }
                        }
                    }
                } else n_oob += e - j;
                j = e;
                if (j > jb) break;
#pragma unroll
                for (int a = 0; a < 3; ++a) if (ev[a] == j) {
                    // step j is the first one in another cell along this axis: the coordinate is monotone and moves by at most ~1 voxel per step
                    // (|d| <= 1), so that cell is the neighbour in the direction of travel (-1 / nb = outside the volume) -- no evaluation needed
                    cell[a] += d[a] > 0.0f ? 1 : -1;
                    ev[a] = next_axis_event(d[a], id[a], P.T[a], P, tv[a], j, jb, cell[a], hh[a], NN[a], nb[a]);
                }
            }
        }
    }
    if (r < nrays) for (int q = nslot; q < spl; ++q) myseg[q] = ~0ull;        // unused private slots
    TSL_TICK(F, 2);
    __syncthreads();
    TSL_TICK(F, 3);
    TSL_TICK(F, 5);
    for (int i = threadIdx.x; i < SEG_LH; i += 256) {        // SEG_LH is a multiple of 256: no lane leaves the loop early
        const int c = s_cnt[i];
        const int b = c ? s_key[i] : 0;
        const bool first = c && __hip_atomic_fetch_add(&F.bhist[b], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
        const int q = wave_reserve(&F.counters[1], first);      // the first block to count a brick this frame lists it as active
        if (first) { if (q < F.max_frame_bricks) F.act_b[q] = b; else frame_fail(M, F, 2); }
    }
    TSL_TICK(F, 6);
#ifdef TSL_TIMING
    if (lane_id() == 0 && _wv < 16384) F.dbg[_wv * 16 + 15] = working;
#endif
    (void)working;
    n_ok = wave_sum_ll(n_ok); n_oob = wave_sum_ll(n_oob);
    {   // one atomic pair per workgroup (not per wave): the counters of a frame are single addresses
        __shared__ long long s_steps[2];
        if (threadIdx.x == 0) { s_steps[0] = 0; s_steps[1] = 0; }
        __syncthreads();
        if (lane_id() == 0) {
            if (n_ok) atomicAdd(reinterpret_cast<unsigned long long*>(&s_steps[0]), (unsigned long long)n_ok);
            if (n_oob) atomicAdd(reinterpret_cast<unsigned long long*>(&s_steps[1]), (unsigned long long)n_oob);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (s_steps[0]) atomic_add_i64(&F.stats->steps, s_steps[0]);
            if (s_steps[1]) atomic_add_i64(&F.stats->steps_oob, s_steps[1]);
        }
    }
}

// K4b: lay out the frame's active bricks (listed by k_segments) -- a range of the sorted segment array per brick -- and the batch's
// integrate work list.  A brick whose segments of ALL frames of the batch together number at most `unit_max` becomes one UNIT: one
// workgroup walks its frames in order and keeps the voxels in registers in between (one read and one write of the brick per batch,
// no ordering traffic).  A heavier brick is walked in PARTS of at most `psegs` segments of one frame; every part owns ONE slot of the
// batch's merge slab and stores its 4096 {num, den} sums there (plain coalesced stores: no atomics, no tickets), and k_apply_slab,
// launched behind the brick kernel, sums the parts of every (frame, brick) and applies the frames in order.
// Items are binned into PLAN_NCLS classes by the segments they walk (long first).  Nothing has to be in any particular order inside a
// class, so block-aggregated reservations replace a prefix scan, and no thread waits for another: the slots of a (frame, brick) are
// recorded in the FRAME's own per-brick word (bslab), the brick's first frame of the batch lists it as heavy.
// unit  = { brick id, segments of the batch, pool index (claimed here, on the brick's first touch ever), frames of the batch with segments }
// part  = { first segment, segments | parts of the (frame, brick) << 16, pool index, slab slot }
// heavy = { brick id, pool index, frames of the batch with segments, - }
__device__ __forceinline__ int plan_class(int w) { return w >= 2560 ? 0 : (w >= 1280 ? 1 : (w >= 512 ? 2 : 3)); }
// Three tiers by the brick's segments of the whole batch (wb):
//   wb <= unit_half            UNIT over all its frames;
//   unit_half < wb <= unit_max the brick's FIRST frames (about half of its segments) are a unit, its later frames are walked as parts and
//                              applied by k_apply_slab on top of what the unit wrote (the apply kernel is the next launch on the stream, and
//                              the split follows the frame order, so the frames are still applied in order): the longest chain of a launch
//                              -- a unit near the limit, walked by one workgroup -- halves for four instead of eight slab slots per brick;
//   wb > unit_max              parts for every frame.
__global__ void __launch_bounds__(256) k_plan(MapDev M, BatchDev B, int psegs, int unit_max, int unit_half)
{
    if ((int)blockIdx.y >= B.n) return;
    const int y = blockIdx.y;
    const FrameDev& F = B.f[y];
    const int listed = F.counters[1];
    const int nact = min(listed, F.max_frame_bricks);
    if ((int)blockIdx.x * 256 >= nact && blockIdx.x) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int v = 0, b = 0, np = 0, pcls = -1, ucls = -1, tm = 0, wb = 0, tmu = 0, tmp = 0, wu = 0;
    if (i < nact) {
        b = F.act_b[i];
        v = F.bhist[b];
        int vq[TSL_NB];
#pragma unroll
        for (int q = 0; q < TSL_NB; ++q) { vq[q] = q < B.n ? B.f[q].bhist[b] : 0; if (vq[q] > 0) tm |= 1 << q; wb += vq[q]; }
        // frames the unit walks (tmu) / frames walked as parts (tmp): every frame's thread of the brick derives the same split
        if (wb > unit_max) { tmp = tm; }
        else if (wb > unit_half) {
            int acc = 0;
#pragma unroll
            for (int q = 0; q < TSL_NB; ++q) if ((tm >> q) & 1) { if (2 * acc < wb) { tmu |= 1 << q; wu += vq[q]; } else tmp |= 1 << q; acc += vq[q]; }
        } else { tmu = tm; wu = wb; }
        if (tmu && (tmu & -tmu) == (1 << y)) ucls = plan_class(wu);                       // the first frame of the unit's frames lists the unit
        if (((tmp >> y) & 1) && v > 0) { np = (v + psegs - 1) / psegs; pcls = plan_class((v + np - 1) / np); }
    }
    const int off = block_reserve_n(&F.counters[3], v);
    const int s0 = block_reserve_n(&B.f[0].counters[HDR_SLAB], np);                       // one slab slot per part, consecutive per (frame, brick)
    int p0 = 0, u0 = 0;
    for (int c = 0; c < PLAN_NCLS; ++c) {
        const int q = block_reserve_n(&F.counters[HDR_PARTS + c], pcls == c ? np : 0); if (pcls == c) p0 = q;
        const int u = block_reserve_n(&B.f[0].counters[HDR_UNITS + c], ucls == c ? 1 : 0); if (ucls == c) u0 = u;
    }
    if (i < nact) {
        F.boffset[b] = off;
        F.bnseg[b] = v;
        const int pool = (np || ucls >= 0) ? pool_claim<false>(M, B.p[y]->slot, b) : -1;      // < 0: pool exhausted (reported through M.err), the item is skipped
        if (ucls >= 0) {
            if (u0 < B.f[0].unit_cap) B.f[0].unit_tab[(size_t)ucls * B.f[0].unit_cap + u0] = make_int4(b, wu, pool, tmu);
            else for (int q = 0; q < B.n; ++q) if ((tm >> q) & 1) frame_fail(M, B.f[q], 2);
        }
        if (np && pool >= 0) {
            if ((tmp & -tmp) == (1 << y)) {                                               // the first of the brick's part frames lists it for k_apply_slab
                const int hi = __hip_atomic_fetch_add(&B.f[0].counters[HDR_HEAVY], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (hi < B.f[0].max_frame_bricks) B.f[0].heavy_tab[hi] = make_int4(b, pool, tmp, 0);
                else for (int q = 0; q < B.n; ++q) if ((tm >> q) & 1) frame_fail(M, B.f[q], 2);
            }
            if (s0 + np > F.max_frame_bricks || np >= (1 << PART_NP_BITS) || p0 + np > F.part_cap) frame_fail(M, F, 2);      // this frame is not integrated at all
            else {
                F.bslab[b] = s0 | (np << SLAB_SLOT_BITS);
                const int per = (v + np - 1) / np;
                int4* tab = F.part_tab + (size_t)pcls * F.part_cap;
                for (int k = 0; k < np; ++k) {
                    const int pos = k * per, n = min(v, pos + per) - pos;
                    tab[p0 + k] = make_int4(off + pos, (int)((uint32_t)n | ((uint32_t)np << 16)), pool, s0 + k);
                }
            }
        }
    }
    if (i == 0) { F.stats->bricks = listed; if (listed > F.max_frame_bricks) frame_fail(M, F, 2); }
}

// K4c: counting sort by brick (LDS hash of the bricks seen in each 4096-segment tile, one global reservation per (tile, brick))
__global__ void __launch_bounds__(256) k_scatter(BatchDev B)
{
    if ((int)blockIdx.y >= B.n) return;
    const FrameDev& F = B.f[blockIdx.y];
    __shared__ int s_key[LH_SIZE];
    __shared__ int s_cnt[LH_SIZE];
    __shared__ int s_base[LH_SIZE];
    if (B.p[blockIdx.y]->group) {       // k_segments has read the frame's sensor-voxel table: the slots the frame opened are emptied for the set's next frame
        const int na = F.counters[6], nx = F.counters[7];
        for (int i = blockIdx.x * 256 + threadIdx.x; i < na + nx; i += gridDim.x * 256) {
            HSlot* const hs = F.htab + (i < na ? F.act[i] : F.actx[i - na]);
            hs->key = H_EMPTY; hs->cnt = 0;
        }
    }
    const int total = (int)min((long long)*F.nrays * SEG_RAY_SLOTS + F.counters[2], (long long)F.seg_cap);
    const int ntiles = (total + SCATTER_TILE - 1) / SCATTER_TILE;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int i = threadIdx.x; i < LH_SIZE; i += 256) { s_key[i] = LH_EMPTY; s_cnt[i] = 0; }
        __syncthreads();
        const int t0 = tile * SCATTER_TILE;
        unsigned long long key[SCATTER_TILE / 256]; int rank[SCATTER_TILE / 256]; int hs[SCATTER_TILE / 256];
#pragma unroll
        for (int q = 0; q < SCATTER_TILE / 256; ++q) {
            const int i = t0 + q * 256 + threadIdx.x;
            rank[q] = -1; hs[q] = -1;
            if (i < total) {
                key[q] = F.seg[i];
                if (key[q] != ~0ull) {
                    const int b = (int)(key[q] >> STG_B_SHIFT);
                    hs[q] = lh_slot<10>(s_key, b);
                    if (hs[q] >= 0) rank[q] = atomicAdd(&s_cnt[hs[q]], 1);
                    else rank[q] = F.boffset[b] + __hip_atomic_fetch_add(&F.bcursor[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // hash full (rare)
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < LH_SIZE; i += 256) {
            const int c = s_cnt[i];
            if (c) { const int b = s_key[i]; s_base[i] = F.boffset[b] + __hip_atomic_fetch_add(&F.bcursor[b], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SCATTER_TILE / 256; ++q)
            if (rank[q] >= 0) F.seg_sorted[(hs[q] >= 0 ? s_base[hs[q]] : 0) + rank[q]] = key[q];
        __syncthreads();
    }
}

// =====================================================================================================
// Brick kernel, round 2 -- building blocks: the same parts and the same exact sums as round 1, reshaped for the way the SIMDs issue.
//   * the ray step is branch-free (the division form is a template parameter, the square root takes max(s2, 2^-96) --
//     below that the fixed-point term rounds to zero either way -- and the 64-bit conversion is decided once per pair
//     of steps for the whole wave), and two steps are evaluated per iteration: two independent dependency chains per
//     wave where the r01 kernel had one, at the same two waves per SIMD;
//   * {num} and {den} are separate 32 KiB planes (an 8-byte entry spans two of the 64 banks instead of four) with a
//     5-bit XOR swizzle of the x / y digits into the bank bits;
//   * the flush computes all 16 voxels of a thread without branches (stores are predicated), and converts the sums
//     through the 32-bit path when every sum of the wave fits.
// =====================================================================================================
#ifdef TSL_NOSWZ      // developer A/B: the accumulator planes without the bank swizzle
__device__ __forceinline__ int acc_swz5(int l) { return l; }
#else
__device__ __forceinline__ int acc_swz5(int l) { return l ^ (((l >> 8) ^ (l >> 5)) & 31); }
#endif
struct StepK { float vs, rvs, T0, T1, T2; int hN, hNz; };

template <bool FASTDIV>
__device__ __forceinline__ void step_eval(const RayRegs& R, const StepK& K, int j, int* slot, float* qf)
{
    const float jf = (float)j;
    const float x0 = (R.d0 * jf) * K.vs + K.T0, x1 = (R.d1 * jf) * K.vs + K.T1, x2 = (R.d2 * jf) * K.vs + K.T2;      // dense_tsdf.py:253
    const int i0 = rnd_i(div_vs(x0, K.vs, K.rvs, FASTDIV ? 1 : 0)), i1 = rnd_i(div_vs(x1, K.vs, K.rvs, FASTDIV ? 1 : 0)), i2 = rnd_i(div_vs(x2, K.vs, K.rvs, FASTDIV ? 1 : 0));   // :254
    *slot = acc_swz5((((i0 + K.hN) & 15) << 8) | (((i1 + K.hN) & 15) << 4) | ((i2 + K.hNz) & 15));
    const float v0 = R.P0 - x0, v1 = R.P1 - x1, v2 = R.P2 - x2;                                                      // :258
    const float s2 = (v0 * v0 + v1 * v1) + v2 * v2;
    // s2 < 2^-96: dist < 2^-48 and w <= 2^16, so |w * sd| * 2^24 < 2^-8 rounds to 0 whatever the root is
    const float dist = sqrt_rn_norm(fmaxf(s2, 1.2621774483536189e-29f));                                             // :259
    const float dot = (v0 * R.pf0 + v1 * R.pf1) + v2 * R.pf2;
    const float sd = dot == 0.0f ? 0.0f : copysignf(dist, dot);                                                      // :260  dist * sign(dot)
    *qf = rintf((R.w * sd) * TSL_FIX_SCALE);                                                                         // integer-valued (to_fix)
}

// sums -> f32, exactly rounded once: through one v_cvt_f32_i32 when the value fits 32 bits
__device__ __forceinline__ bool fits_i32(long long q) { return q == (long long)(int)q; }
__device__ __forceinline__ float from_fix32(long long q) { return (float)(int)q * (float)TSL_FIX_INV; }

#define FLUSH_CHUNK 4
template <int N>
__device__ __forceinline__ void apply_chunk(const uint32_t* old, const long long* qn, const long long* qd, uint32_t* nv, bool all_small)
{
    if (all_small) {
#pragma unroll
        for (int q = 0; q < N; ++q) nv[q] = apply_update_f(old[q], from_fix32(qn[q]), from_fix32(qd[q]));
    } else {
#pragma unroll
        for (int q = 0; q < N; ++q) nv[q] = apply_update_f(old[q], from_fix(qn[q]), from_fix(qd[q]));
    }
}

// =====================================================================================================
// k_integrate_batch: phase B of a whole batch of frames (up to TSL_NB) as ONE persistent, software-pipelined launch.
//
// The work list (k_plan's tables, PLAN_NCLS length classes each): all UNITS of the batch, long to short, then all PARTS, long to short
// (by class, frame 0, 1, ... inside a class).  The resident workgroups claim items from it in rank order through one counter, two
// items ahead of the one they walk.
//   UNIT  a brick whose whole batch fits one workgroup: its frames are walked in order, the brick's voxels stay in registers in
//         between (each frame's sums are applied to them exactly as a per-frame launch would apply them to memory), the brick is
//         read once and written once per batch and nothing has to be ordered against other workgroups.
//   PART  (a slice of) one frame's segments in a heavier brick.  A part's walk depends only on its frame's rays, never on the map, so
//         the parts of all frames of the batch are walked side by side, in any order; each stores its 4096 {num, den} sums into a
//         slot of its own in the batch's merge slab in HBM (plain coalesced stores: no atomics, no tickets, nothing to clear).
//         k_apply_slab, the next launch on the stream, adds the parts of every (frame, brick) and applies the frames in frame order
//         exactly as per-frame launches would; the brick is read once and written once.  Nobody ever waits for another workgroup.
//
// Inside a workgroup the pipeline is the one of round 2's per-frame kernel, over "steps" (one chunk of CSEGS segments of one frame of
// one item): while a step is walked, the keys of the next step -- next chunk, next frame of the unit, or the first step of the next
// claimed item -- are in flight; after the walk they are length-sorted in their own 8 KiB of LDS and their ray records requested,
// and only then the finished frame is applied.  The items' descriptions live in a three-slot LDS ring (current, next, being fetched).
// One launch per batch removes three of four kernel tails (half of the workgroups of a per-frame launch were idle for the second
// half of it), and with longest-first dynamic claims over an eight-frame work list the heavy bricks next to the sensor no longer
// set the length of every frame.
// =====================================================================================================
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }      // a value every lane holds (read from LDS): keep it in an SGPR
// item record in LDS
#define IT_UNIT 0       // 1: unit
#define IT_POOL 1       // pool index of the brick (< 0: skip)
#define IT_SLAB 2       // parts: the part's slab slot
#define IT_NP   3       // parts: parts of the (frame, brick)
#define IT_TMALL 4      // units: frames of the batch that integrate into the brick
#define IT_FMASK 5      // frames this item walks
#define IT_OFF  6                     // [IT_OFF + q] first segment of frame q
#define IT_N    (IT_OFF + TSL_NB)     // [IT_N + q] segments of frame q
#define IT_WORDS (IT_N + TSL_NB + 2)
#define NRANGE (PLAN_NCLS * (TSL_NB + 1))

template <bool TEX, bool FASTDIV, int NT, int SPT, int WPE>
__global__ void __launch_bounds__(NT, WPE) k_integrate_batch(MapDev M, BatchDev B, int kind)
{
    // kind: 0 = the whole work list; split launches: 1 = the UNITS only (main stream, behind the previous batch's phase B: they read and write
    // the map), 2 = the PARTS only (the batch's phase-A stream: a part reads its frame's rays and writes its own slab slot, never the map)
    // NT threads walk a step of up to CSEGS = SPT * NT segments (SPT per thread), WPE = waves per SIMD the register budget allows.
    //   NT 256, SPT 4, WPE 2: two workgroups per CU (74 KiB of LDS each), 8 waves per CU;
    //   NT 512, SPT 4, WPE 2: one workgroup per CU (82 KiB), 8 waves on one brick: half the walk latency per brick, half as many bricks in flight;
    //   NT 512, SPT 2, WPE 4: steps of 1024 segments, 74 KiB of LDS and at most 128 VGPRs: TWO 512-thread workgroups per CU = 16 waves per CU,
    //                         a workgroup's barriers and LDS round trips are covered by the other's walk.
    constexpr int CSEGS = SPT * NT, VPT = TSL_BRK3 / NT, CH = VPT < FLUSH_CHUNK ? VPT : FLUSH_CHUNK;
    static_assert(VPT <= 16, "the written / first-touch masks hold 16 voxels per thread");
#ifdef TSL_EXP_SLOT16      // developer A/B: {num, den} of a voxel side by side in one 16-byte slot instead of two 32 KiB planes
    __shared__ ulonglong2 s_acc[TSL_BRK3];                      // 64 KiB
#define S_NUM(i) s_acc[i].x
#define S_DEN(i) s_acc[i].y
#else
    __shared__ unsigned long long s_num[TSL_BRK3];              // 32 KiB
    __shared__ unsigned long long s_den[TSL_BRK3];              // 32 KiB
#define S_NUM(i) s_num[i]
#define S_DEN(i) s_den[i]
#endif
    __shared__ unsigned long long s_keys[CSEGS];                // 8 / 16 KiB: length sort of the NEXT step's keys while the planes hold the current sums
    __shared__ uint32_t s_win[TEX ? TSL_BRK3 : 1];              // texture: colour winner per voxel (first pixel of the ray + 1)
    __shared__ int s_bin[64];
    __shared__ int s_claim[2];
    __shared__ int s_uq[TSL_NB];                                // distinct voxels this workgroup updated, per frame of the batch
    __shared__ int s_cum[NRANGE + 1];                           // first rank of every range of the work list (units by class, then parts by class and frame)
    __shared__ int s_it[3][IT_WORDS];
    uint32_t okmask = 0u;
#pragma unroll
    for (int q = 0; q < TSL_NB; ++q) if (q < B.n) {
        const FrameDev& Fq = B.f[q];
        if (Fq.counters[HDR_FAIL] == 0) okmask |= 1u << q;                 // nothing of a frame that overflowed its scratch is integrated
        // restore the "all zero between uses" invariant of the set's per-brick histogram / cursor
        const int nact = min(Fq.counters[1], Fq.max_frame_bricks);
        for (int i = blockIdx.x * NT + threadIdx.x; i < nact; i += gridDim.x * NT) { const int b = Fq.act_b[i]; Fq.bhist[b] = 0; Fq.bcursor[b] = 0; }
    }
    if (threadIdx.x < TSL_NB) s_uq[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        int acc = 0, k = 0;
#ifdef TSL_CLASS_MAJOR        // developer A/B: the first order of the list
        for (int c = 0; c < PLAN_NCLS; ++c) {
            s_cum[k++] = acc; acc += min(B.f[0].counters[HDR_UNITS + c], B.f[0].unit_cap);
            for (int q = 0; q < TSL_NB; ++q) { s_cum[k++] = acc; if ((okmask >> q) & 1u) acc += min(B.f[q].counters[HDR_PARTS + c], B.f[q].part_cap); }
        }
#else
        // ALL units first (long to short), then all parts (long to short): a unit is a chain of up to eight frame steps with ~4 us of fixed cost
        // each -- a unit of 600 segments takes 45 us, a part of 3 000 segments 26 -- so ranking the items by segments alone left the light units
        // for the end of the launch, where each of them added its whole chain to the span
        for (int c = 0; c < PLAN_NCLS; ++c) { s_cum[k++] = acc; if (kind != 2) acc += min(B.f[0].counters[HDR_UNITS + c], B.f[0].unit_cap); }
        for (int c = 0; c < PLAN_NCLS; ++c)
            for (int q = 0; q < TSL_NB; ++q) { s_cum[k++] = acc; if (kind != 1 && ((okmask >> q) & 1u)) acc += min(B.f[q].counters[HDR_PARTS + c], B.f[q].part_cap); }
#endif
        s_cum[NRANGE] = acc;
        s_claim[0] = __hip_atomic_fetch_add(&B.f[0].counters[kind == 2 ? HDR_CLAIM2 : HDR_CLAIM], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int total = uni(s_cum[NRANGE]);
    int* const claim = &B.f[0].counters[kind == 2 ? HDR_CLAIM2 : HDR_CLAIM];
    // rank -> table entry; *kq = -1 for a unit, else the frame of the part
    auto entry_of = [&](int r, int* kq) -> int4 {
        int k = 0;
#pragma unroll
        for (int j = 1; j < NRANGE; ++j) k += r >= uni(s_cum[j]) ? 1 : 0;
#ifdef TSL_CLASS_MAJOR
        const int local = r - uni(s_cum[k]), cls = k / (TSL_NB + 1), sub = k - cls * (TSL_NB + 1);
#else
        const int local = r - uni(s_cum[k]), cls = k < PLAN_NCLS ? k : (k - PLAN_NCLS) / TSL_NB, sub = k < PLAN_NCLS ? 0 : 1 + (k - PLAN_NCLS) % TSL_NB;
#endif
        *kq = sub - 1;
        if (sub == 0) return B.f[0].unit_tab[(size_t)cls * B.f[0].unit_cap + local];
        const FrameDev& Fq = B.f[sub - 1];
        return Fq.part_tab[(size_t)cls * Fq.part_cap + local];
    };
    // per-frame segment ranges of a unit (thread q < TSL_NB loads frame q's); parts carry theirs in the entry
    auto info_of = [&](const int4 e, int kq, int* io, int* in) {
        *io = 0; *in = 0;
        const int q = threadIdx.x;
        if (q < TSL_NB) {
            if (kq < 0) { if ((((uint32_t)e.w) & okmask) >> q & 1u) { *io = B.f[q].boffset[e.x]; *in = B.f[q].bnseg[e.x]; } }
            else if (q == kq) { *io = e.x; *in = e.y & 0xffff; }
        }
    };
    auto commit = [&](int slot, const int4 e, int kq, int io, int in) {
        int* it = s_it[slot];
        if (threadIdx.x < TSL_NB) { it[IT_OFF + threadIdx.x] = io; it[IT_N + threadIdx.x] = in; }
        if (threadIdx.x == 0) {
            const uint32_t tm = kq < 0 ? (uint32_t)e.w & okmask : 1u << kq;
            uint32_t fm = tm;
            int pool = e.z;
            if (fm == 0u) { fm = 1u; pool = -1; }                  // a unit whose frames all overflowed: nothing to walk, nothing to apply
            it[IT_UNIT] = kq < 0 ? 1 : 0; it[IT_POOL] = pool; it[IT_SLAB] = kq < 0 ? 0 : e.w; it[IT_NP] = kq < 0 ? 1 : (e.y >> 16) & ((1 << PART_NP_BITS) - 1);
            it[IT_TMALL] = (int)tm; it[IT_FMASK] = (int)fm;
        }
    };
    const int r0 = uni(s_claim[0]);
    if (r0 >= total) return;
    { int kq, io, in; const int4 e = entry_of(r0, &kq); info_of(e, kq, &io, &in); commit(0, e, kq, io, in); }
    __syncthreads();
    // steps (chunks of CSEGS segments) of the frames `fm` of the item in ring slot `sl`
    auto steps_of = [&](int sl, uint32_t fm) -> int {
        int n = 0;
#pragma unroll
        for (int q = 0; q < TSL_NB; ++q) if ((fm >> q) & 1u) n += max(1, (uni(s_it[sl][IT_N + q]) + CSEGS - 1) / CSEGS);
        return n;
    };
    long long uniq = 0;
    int t = 0, slot = 0, c = 0; (void)t;                       // t-th item of this workgroup (in ring slot `slot`), chunk c of frame f of it
    int n_known = 0;                                           // items already fetched behind the current one (ring slots slot+1, slot+2)
    bool exhausted = false;                                    // a claim came back beyond the end of the list
    int f = __builtin_ctz((uint32_t)uni(s_it[0][IT_FMASK]));

    unsigned long long kk[SPT]; uint4 recs[SPT]; uint32_t wids[SPT];
    // length-sort `nseg` keys (this thread holds k[q] = key q*NT+tid) through s_keys / s_bin, deal them out in alternating directions
    // and request the ray records of the dealt keys from frame FD.  Counting sort by step count, descending: the lanes of a wave walk
    // segments of (almost) equal length and every thread gets about the same number of steps.  Contains 3 barriers; s_bin must be zero on entry.
#define TSL_SORT_DEAL(KIN, NSEG, FD)                                                                                    \
    {                                                                                                                   \
        int rr[SPT];                                                                                                    \
        _Pragma("unroll") for (int q = 0; q < SPT; ++q) {                                                               \
            const int i = q * NT + (int)threadIdx.x; rr[q] = -1;                                                        \
            if (i < (NSEG)) rr[q] = atomicAdd(&s_bin[63 - (int)(KIN[q] & 63ull)], 1);                                   \
        }                                                                                                               \
        __syncthreads();                                                                                                \
        if (threadIdx.x < 64) {                                                                                         \
            const int cb = s_bin[threadIdx.x]; int inc = cb;                                                            \
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if ((int)threadIdx.x >= d) inc += o; }  \
            s_bin[threadIdx.x] = inc - cb;                                                                              \
        }                                                                                                               \
        __syncthreads();                                                                                                \
        _Pragma("unroll") for (int q = 0; q < SPT; ++q) if (rr[q] >= 0) s_keys[s_bin[63 - (int)(KIN[q] & 63ull)] + rr[q]] = KIN[q]; \
        __syncthreads();                                                                                                \
        _Pragma("unroll") for (int q = 0; q < SPT; ++q) {                                                               \
            const int i = q * NT + ((q & 1) ? NT - 1 - (int)threadIdx.x : (int)threadIdx.x);                            \
            if (i < (NSEG)) {                                                                                           \
                kk[q] = s_keys[i];                                                                                      \
                const int r = (int)((kk[q] >> (SEG_CNT_BITS + SEG_J_BITS)) & ((1u << STG_RAY_BITS) - 1));               \
                recs[q] = (FD).rayA[r];                                                                                 \
                wids[q] = TEX ? (FD).rayFirst[r] + 1u : 0u;                                                             \
            }                                                                                                           \
        }                                                                                                               \
        if (threadIdx.x < 64) s_bin[threadIdx.x] = 0;              /* ordered against the next use by the barriers in between */ \
    }
    // the pipeline is empty (launch start, or the workgroup ran out of fetched items): request the first step of the item in `SL` now
#define TSL_PRIME(SL, FQ)                                                                                               \
    {                                                                                                                   \
        const FrameDev& F0 = B.f[FQ];                                                                                   \
        const int off0 = uni(s_it[SL][IT_OFF + (FQ)]), nseg0 = min(uni(s_it[SL][IT_N + (FQ)]), CSEGS);                  \
        unsigned long long k0[SPT];                                                                                     \
        _Pragma("unroll") for (int q = 0; q < SPT; ++q) { const int i = q * NT + threadIdx.x; k0[q] = i < nseg0 ? F0.seg_sorted[off0 + i] : 0ull; } \
        TSL_SORT_DEAL(k0, nseg0, F0)                                                                                    \
    }

    {   // the sums are zero between items: cleared here once, afterwards by whoever reads them
#ifdef TSL_EXP_SLOT16
        for (int i = threadIdx.x; i < TSL_BRK3; i += NT) s_acc[i] = make_ulonglong2(0ull, 0ull);
#else
        ulonglong2* zn = reinterpret_cast<ulonglong2*>(s_num); ulonglong2* zd = reinterpret_cast<ulonglong2*>(s_den);
        for (int i = threadIdx.x; i < TSL_BRK3 / 2; i += NT) { zn[i] = make_ulonglong2(0ull, 0ull); zd[i] = make_ulonglong2(0ull, 0ull); }
#endif
        if (TEX) for (int i = threadIdx.x; i < TSL_BRK3; i += NT) s_win[i] = 0u;
        if (threadIdx.x < 64) s_bin[threadIdx.x] = 0;
        __syncthreads();
        TSL_PRIME(0, f)
    }
    uint32_t old[VPT];
    uint32_t vmask = 0u;                                       // bits 0..15: voxel written by this item, 16..31: its weight was zero when the item began
    bool have_old = false;
    for (;;) {
        const int* const it = s_it[slot];
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot == 0 ? 2 : slot - 1;
        const bool unit = uni(it[IT_UNIT]) != 0;
        const int p = uni(it[IT_POOL]), rk = uni(it[IT_SLAB]);
        const uint32_t fmask = (uint32_t)uni(it[IT_FMASK]);
        const int n_f = uni(it[IT_N + f]);
        const FrameDev& F = B.f[f];
        const FrameParams& P = *B.p[f];
        const StepK K = { P.vs, P.rvs, P.T[0], P.T[1], P.T[2], M.hN, M.hNz };
        const uint32_t later = fmask >> (f + 1);
        const bool first_frame = (fmask & ((1u << f) - 1u)) == 0u, last_frame = later == 0u;
        const bool first_step = first_frame && c == 0;
        const int nseg = min(CSEGS, n_f - c * CSEGS);
        const bool last_chunk = (c + 1) * CSEGS >= n_f;               // last chunk of the frame: apply after the walk
        const bool last_step = last_chunk && last_frame;
        // claim the next item late: when at most one more step is known behind this one (two steps of lookahead keep the pipeline
        // full; an early claim would take an item away from a workgroup that could start it sooner)
        int known = max(1, (n_f + CSEGS - 1) / CSEGS) - 1 - c + steps_of(slot, later << (f + 1));
        if (n_known >= 1) known += steps_of(slot1, (uint32_t)uni(s_it[slot1][IT_FMASK]));
        const bool do_claim = !exhausted && n_known < 2 && known <= 1;
        if (do_claim && threadIdx.x == 0) s_claim[0] = __hip_atomic_fetch_add(claim, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef TSL_TIMING
        long long* const _rec = F.dbg + 131072 + (size_t)blockIdx.x * 128 + (size_t)(t < 16 ? t : 15) * 8;
        if (threadIdx.x == 0 && first_step) { _rec[0] = wall_clock64(); _rec[1] = (unit ? 1 : 0) | (uni(it[IT_NP]) << 8) | ((long long)fmask << 32); _rec[2] = 0; _rec[5] = 0; }
        if (threadIdx.x == 0 && c == 0) _rec[2] += n_f;
        long long* const _ph = F.dbg + 196608 + (size_t)blockIdx.x * 128 + (size_t)(t < 16 ? t : 15) * 8;
        if (threadIdx.x == 0 && first_step) { _ph[0] = _ph[1] = _ph[2] = _ph[3] = _ph[4] = 0; }
        const long long _ta = wall_clock64();
#endif
        // the next step: the next chunk of this frame, the first chunk of the unit's next frame, or the first step of the next item
        const bool has_next = !last_step || n_known >= 1;
        int fx = f, cx = c + 1, slotx = slot;
        if (last_chunk) {
            cx = 0;
            if (!last_frame) fx = f + 1 + __builtin_ctz(later);
            else if (has_next) { slotx = slot1; fx = __builtin_ctz((uint32_t)uni(s_it[slotx][IT_FMASK])); }
        }
        const FrameDev& Fx = B.f[fx];
        // requests that ride under the walk: the next step's keys, the unit's voxels
        unsigned long long kn[SPT];
        const int offx = uni(s_it[slotx][IT_OFF + fx]) + cx * CSEGS;
        const int nsegn = has_next ? min(CSEGS, uni(s_it[slotx][IT_N + fx]) - cx * CSEGS) : 0;
#pragma unroll
        for (int q = 0; q < SPT; ++q) { const int i = q * NT + threadIdx.x; kn[q] = i < nsegn ? Fx.seg_sorted[offx + i] : 0ull; }
        const uint32_t* const twr = M.tw + (size_t)(p >= 0 ? p : 0) * TSL_BRK3;
        if (first_step) { have_old = false; vmask = 0u; }
        if (!TEX && unit && p >= 0 && first_step) {                  // the unit's voxels are requested under its first walk
#pragma unroll
            for (int q = 0; q < VPT; ++q) old[q] = twr[q * NT + threadIdx.x];
            have_old = true;
        }
        // ---- walk ----
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int i = q * NT + ((q & 1) ? NT - 1 - (int)threadIdx.x : (int)threadIdx.x);
            if (i >= nseg) continue;
            const unsigned long long key = kk[q];
            const int cnt = (int)(key & ((1u << SEG_CNT_BITS) - 1)), j0 = (int)((key >> SEG_CNT_BITS) & ((1u << SEG_J_BITS) - 1));
            const RayRegs R = make_ray(recs[q], 0, P);
            const uint32_t wid = wids[q];
            // lanes start at different offsets inside their (equally long) segments: rays that enter a brick together -- all
            // of them next to the sensor -- would otherwise hit the same few voxels in the same iteration
            // ... and NEIGHBOURING lanes far apart: the lanes of a wave hold segments of equal length, so neighbours in the sorted order are often
            // neighbours in space, and 1 / 64 of a segment apart (the first form: lane * cnt / 64) they still added to the same voxel in the same
            // instruction -- three quarters of the LDS conflict cycles were same-address ones.  27 / 64 of a segment apart: SQ_LDS_ADDR_CONFLICT
            // 8.9 M -> 2.4 M per launch, conflict cycles 53 % -> 45 % of the active LDS cycles (profiles/r04_brick_kernel_experiments.txt)
#ifndef TSL_EXP_STAGGER
#define TSL_EXP_STAGGER 27  // (developer A/B: 1 = the first form)
#endif
            int off = ((int)((threadIdx.x * (unsigned)TSL_EXP_STAGGER) & 63u) * cnt) >> 6;
            for (int s = 0; s < cnt; s += 2) {
                const int ja = j0 + off; off = (off + 1 == cnt) ? 0 : off + 1;
                const int jb = j0 + off; off = (off + 1 == cnt) ? 0 : off + 1;
                int la, lb; float qa, qb;
                step_eval<FASTDIV>(R, K, ja, &la, &qa);
                step_eval<FASTDIV>(R, K, jb, &lb, &qb);
                long long na = (long long)(int)qa, nb = (long long)(int)qb;
                if (__builtin_expect(__any(!(fabsf(qa) < 2147483648.0f) || !(fabsf(qb) < 2147483648.0f)), 0)) { na = __float2ll_rn(qa); nb = __float2ll_rn(qb); }
                atomicAdd(&S_NUM(la), (unsigned long long)na);
                atomicAdd(&S_DEN(la), (unsigned long long)R.qden);
                if (TEX) atomicMax(&s_win[la], wid);                                           // dense_tsdf.py:268-269, order-free winner
                const bool vb = s + 1 < cnt;
#ifdef TSL_EXP_ADDZERO      // developer A/B: the first form -- the missing second step of an odd segment's last pair adds zeros (no branch)
                atomicAdd(&S_NUM(lb), (unsigned long long)(vb ? nb : 0ll));
                atomicAdd(&S_DEN(lb), (unsigned long long)(vb ? R.qden : 0ll));
                if (TEX) atomicMax(&s_win[lb], vb ? wid : 0u);
#else                       // ... or is skipped: 5 % fewer LDS conflict cycles, the launch 4 % shorter (profiles/r04_brick_kernel_experiments.txt)
                if (vb) { atomicAdd(&S_NUM(lb), (unsigned long long)nb); atomicAdd(&S_DEN(lb), (unsigned long long)R.qden); if (TEX) atomicMax(&s_win[lb], wid); }
#endif
            }
        }
        __syncthreads();
#ifdef TSL_TIMING
        if (threadIdx.x == 0 && last_step) _rec[3] = wall_clock64();
        const long long _tb = wall_clock64();
#endif
        // the claimed item: its table entry now, its per-frame ranges after the sort (both land in the ring at the end of this step)
        bool fetching = false;
        int4 ne = make_int4(0, 0, -1, 0); int nkq = 0, nio = 0, nin = 0;
        if (do_claim) {
            const int r = uni(s_claim[0]);
            if (r < total) { fetching = true; ne = entry_of(r, &nkq); } else exhausted = true;
        }
        // ---- the next step's keys are here: sort them and request its ray records, then apply under that latency ----
        if (has_next) TSL_SORT_DEAL(kn, nsegn, Fx)
        if (fetching) info_of(ne, nkq, &nio, &nin);
#ifdef TSL_TIMING
        const long long _tc = wall_clock64();
        if (threadIdx.x == 0) { _ph[0] += _tb - _ta; _ph[1] += _tc - _tb; _ph[4] += 1; }
#endif
        if (!last_chunk) {
            if (fetching) { commit(n_known == 0 ? slot1 : slot2, ne, nkq, nio, nin); ++n_known; __syncthreads(); }
            ++c; continue;
        }
        if (p >= 0 && threadIdx.x == 0) M.touch[p] = 1;                   // the brick's TSDF changes in this batch (incremental ESDF)
        if (p >= 0 && unit) {
            uint32_t* tw = M.tw + (size_t)p * TSL_BRK3;
            if (!have_old) {
#pragma unroll
                for (int q = 0; q < VPT; ++q) old[q] = tw[q * NT + threadIdx.x];
                have_old = true;
            }
            if (first_frame) {
#pragma unroll
                for (int q = 0; q < VPT; ++q) vmask |= ((old[q] >> 16) == 0u ? 1u : 0u) << (16 + q);      // W == 0 <=> never integrated
            }
#pragma unroll
            for (int h = 0; h < VPT; h += CH) {                  // CH voxels at a time, no branch between them; the sums are left zero
                long long qn[CH], qd[CH]; uint32_t nv[CH]; bool small = true, anyu = false;
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const int ls = acc_swz5((h + q) * NT + threadIdx.x);
                    qn[q] = (long long)S_NUM(ls); qd[q] = (long long)S_DEN(ls); S_NUM(ls) = 0ull; S_DEN(ls) = 0ull;
                    small = small && fits_i32(qn[q]) && fits_i32(qd[q]); anyu = anyu || qd[q] != 0;
                }
                if (!__any(anyu)) continue;                      // none of this wave's 64 x CH voxels was touched by the frame
                apply_chunk<CH>(old + h, qn, qd, nv, __all(small));
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const bool upd = qd[q] != 0;
                    old[h + q] = upd ? nv[q] : old[h + q];
                    vmask |= (upd ? 1u : 0u) << (h + q);
                    uniq += upd ? 1 : 0;
                    if (TEX && upd) {
                        const int l = (h + q) * NT + threadIdx.x;
                        reinterpret_cast<uint2*>(M.col)[(size_t)p * TSL_BRK3 + l] = F.colpix[s_win[acc_swz5(l)] - 1u];
                    }
                }
            }
            if (TEX) for (int i = threadIdx.x; i < TSL_BRK3; i += NT) s_win[i] = 0u;
            if (last_frame) {                                    // the unit's last frame: the voxels go back to the map
                int8_t* obs = M.obs + (size_t)p * TSL_BRK3;
#pragma unroll
                for (int q = 0; q < VPT; ++q) {
                    const int l = q * NT + threadIdx.x;
                    if ((vmask >> q) & 1u) {
                        tw[l] = old[q];
                        if ((vmask >> (16 + q)) & 1u) obs[l] = 1;       // imported voxels already carry observed = 1
                    }
                }
            }
        } else if (p >= 0) {
            // part of a heavy brick: its sums go to the part's own slot of the batch's slab -- every entry, zeros included, as plain 16-byte
            // stores (a slot is written by exactly one workgroup, so nothing has to be cleared or ordered); k_apply_slab, the next launch on
            // this stream, adds the parts of each (frame, brick) and applies the frames in order
            ulonglong2* acc = reinterpret_cast<ulonglong2*>(F.acc) + (size_t)rk * TSL_BRK3;
#pragma unroll
            for (int q = 0; q < VPT; ++q) {
                const int l = q * NT + threadIdx.x, ls = acc_swz5(l);
                acc[l] = make_ulonglong2(S_NUM(ls), S_DEN(ls));
                S_NUM(ls) = 0ull; S_DEN(ls) = 0ull;
                if (TEX) { F.accw[(size_t)rk * TSL_BRK3 + l] = s_win[ls]; s_win[ls] = 0u; }
            }
        } else {                                                          // brick pool exhausted (reported by k_plan): drop the sums
#ifdef TSL_EXP_SLOT16
            for (int i = threadIdx.x; i < TSL_BRK3; i += NT) s_acc[i] = make_ulonglong2(0ull, 0ull);
#else
            ulonglong2* zn = reinterpret_cast<ulonglong2*>(s_num); ulonglong2* zd = reinterpret_cast<ulonglong2*>(s_den);
            for (int i = threadIdx.x; i < TSL_BRK3 / 2; i += NT) { zn[i] = make_ulonglong2(0ull, 0ull); zd[i] = make_ulonglong2(0ull, 0ull); }
#endif
            if (TEX) for (int i = threadIdx.x; i < TSL_BRK3; i += NT) s_win[i] = 0u;
        }
#ifdef TSL_TIMING
        const long long _td = wall_clock64();
#endif
        {   // this frame's distinct-voxel count: collected in LDS, one global atomic per workgroup and frame at the end of the launch
            // (same-address atomics serialise at ~12 ns each at the L2: per wave and frame they were ~3 600 per counter and batch)
            const long long u = wave_sum_ll(uniq);
            if (lane_id() == 0 && u) atomicAdd(&s_uq[f], (int)u);
            uniq = 0;
        }
        if (fetching) { commit(n_known == 0 ? slot1 : slot2, ne, nkq, nio, nin); ++n_known; }
#ifdef TSL_TIMING
        if (threadIdx.x == 0 && last_step) _rec[4] = wall_clock64();
        if (threadIdx.x == 0) { _ph[2] += _td - _tc; _ph[3] += wall_clock64() - _td; }
#endif
        if (!has_next && n_known == 0) break;                             // the list is exhausted and nothing is left in the ring
        __syncthreads();                                                  // the sums are zero again, the ring slot is written
        c = 0;
        if (last_frame) { ++t; slot = slot1; --n_known; f = __builtin_ctz((uint32_t)uni(s_it[slot][IT_FMASK])); }
        else f = fx;
        if (!has_next) TSL_PRIME(slot, f)                                 // nothing was in flight for this step
    }
#undef TSL_PRIME
#undef TSL_SORT_DEAL
    __syncthreads();
    if (threadIdx.x < TSL_NB && s_uq[threadIdx.x]) atomic_add_i64(&B.f[threadIdx.x].stats->unique, (long long)s_uq[threadIdx.x]);
}

// =====================================================================================================
// k_apply_slab: the heavy bricks of a batch (k_plan's list), behind the brick kernel on the same stream.  Every part of a heavy brick
// left its 4096 {num, den} sums in a slab slot of its own; here the parts of each (frame, brick) are added (int64: exact, order-free)
// and the frames are applied in frame order, exactly as per-frame launches would apply them -- the brick is read once and written once
// per batch.  One workgroup takes an eighth of a brick (two voxels per thread): the loads of ALL frames are independent of the apply chain,
// so they are in flight together, and the heavy bricks of a batch spread over eight workgroups each instead of serialising behind the last
// part to arrive (round 2's ticket scheme: 25-33 us per brick on the critical path of the persistent kernel, plus an L2 atomic pair per
// voxel and part).
// =====================================================================================================
#ifndef APPLY_SPLIT
#define APPLY_SPLIT 8
#endif
#ifndef APPLY_ROUND
#define APPLY_ROUND 2
#endif
template <bool TEX>
__global__ void __launch_bounds__(256) k_apply_slab(MapDev M, BatchDev B)
{
    __shared__ int s_uq[TSL_NB];                                // distinct voxels this workgroup updated, per frame
    constexpr int VP = TSL_BRK3 / APPLY_SPLIT / 256;           // voxels per thread
    uint32_t okmask = 0u;
#pragma unroll
    for (int q = 0; q < TSL_NB; ++q) if (q < B.n && B.f[q].counters[HDR_FAIL] == 0) okmask |= 1u << q;
    const int nheavy = min(B.f[0].counters[HDR_HEAVY], B.f[0].max_frame_bricks);
    const ulonglong2* const slab = reinterpret_cast<const ulonglong2*>(B.f[0].acc);
    if (threadIdx.x < TSL_NB) s_uq[threadIdx.x] = 0;
    __syncthreads();
    for (int item = blockIdx.x; item < nheavy * APPLY_SPLIT; item += gridDim.x) {
        const int4 e = B.f[0].heavy_tab[item / APPLY_SPLIT];
        const int b = e.x, p = e.y;
        const uint32_t tm = (uint32_t)e.z & okmask;
        if (p < 0 || tm == 0u) continue;
        const int l0 = (item % APPLY_SPLIT) * (TSL_BRK3 / APPLY_SPLIT) + (int)threadIdx.x;
        uint32_t* tw = M.tw + (size_t)p * TSL_BRK3;
        // three rounds of independent loads: the frames' slot words and the voxels, the first part of every frame, then the further
        // parts of the few (frame, brick)s that have them; only then the apply chain, which is pure arithmetic
        int w[TSL_NB];
#pragma unroll
        for (int q = 0; q < TSL_NB; ++q) w[q] = ((tm >> q) & 1u) ? B.f[q].bslab[b] : 0;
        uint32_t cur[VP], first[VP];
#pragma unroll
        for (int k = 0; k < VP; ++k) { cur[k] = tw[l0 + k * 256]; first[k] = cur[k]; }
        ulonglong2 a[TSL_NB][VP]; uint32_t win[TSL_NB][VP];
#pragma unroll
        for (int q = 0; q < TSL_NB; ++q) {
            const bool on = (tm >> q) & 1u;
            const size_t base = (size_t)(w[q] & ((1 << SLAB_SLOT_BITS) - 1)) * TSL_BRK3 + l0;
#pragma unroll
            for (int k = 0; k < VP; ++k) {
                a[q][k] = on ? slab[base + k * 256] : make_ulonglong2(0ull, 0ull);
                win[q][k] = (TEX && on) ? B.f[q].accw[base + k * 256] : 0u;
            }
        }
        // further parts (only the bricks next to the sensor have more than one or two per frame): APPLY_ROUND slots per round, all frames of a
        // round in flight together -- the rounds, not the frames, are the dependent chain
        int npmax = 0;
#pragma unroll
        for (int q = 0; q < TSL_NB; ++q) npmax = max(npmax, w[q] >> SLAB_SLOT_BITS);
        for (int j0 = 1; j0 < npmax; j0 += APPLY_ROUND) {
            ulonglong2 x[TSL_NB][APPLY_ROUND][VP];
#pragma unroll
            for (int q = 0; q < TSL_NB; ++q) {
                const int np = w[q] >> SLAB_SLOT_BITS, s0 = w[q] & ((1 << SLAB_SLOT_BITS) - 1);
#pragma unroll
                for (int t = 0; t < APPLY_ROUND; ++t) {
                    const bool on = j0 + t < np;
                    const size_t base = (size_t)(s0 + (on ? j0 + t : 0)) * TSL_BRK3 + l0;
#pragma unroll
                    for (int k = 0; k < VP; ++k) {
                        x[q][t][k] = on ? slab[base + k * 256] : make_ulonglong2(0ull, 0ull);
                        if (TEX && on) win[q][k] = max(win[q][k], B.f[q].accw[base + k * 256]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < TSL_NB; ++q)
#pragma unroll
                for (int t = 0; t < APPLY_ROUND; ++t)
#pragma unroll
                    for (int k = 0; k < VP; ++k) { a[q][k].x += x[q][t][k].x; a[q][k].y += x[q][t][k].y; }
        }
        bool changed[VP];
#pragma unroll
        for (int k = 0; k < VP; ++k) changed[k] = false;
#pragma unroll
        for (int q = 0; q < TSL_NB; ++q) {
            if (!((tm >> q) & 1u)) continue;                     // (uniform)
            long long ug = 0;
#pragma unroll
            for (int k = 0; k < VP; ++k) {
                if (a[q][k].y != 0ull) {
                    cur[k] = apply_update(cur[k], (long long)a[q][k].x, (long long)a[q][k].y);
                    changed[k] = true; ++ug;
                    if (TEX) reinterpret_cast<uint2*>(M.col)[(size_t)p * TSL_BRK3 + l0 + k * 256] = B.f[q].colpix[win[q][k] - 1u];
                }
            }
            // (one same-address atomic costs ~12 ns at the L2 and they serialise: the counts go through LDS, one atomic per workgroup and frame)
            ug = wave_sum_ll(ug);
            if (lane_id() == 0 && ug) atomicAdd(&s_uq[q], (int)ug);
        }
        __syncthreads();
        if (threadIdx.x < TSL_NB && s_uq[threadIdx.x]) { atomic_add_i64(&B.f[threadIdx.x].stats->unique, (long long)s_uq[threadIdx.x]); s_uq[threadIdx.x] = 0; }
        int8_t* obs = M.obs + (size_t)p * TSL_BRK3;
#pragma unroll
        for (int k = 0; k < VP; ++k) if (changed[k]) {
            tw[l0 + k * 256] = cur[k];
            if ((first[k] >> 16) == 0u) obs[l0 + k * 256] = 1;          // W == 0 <=> never integrated (imported voxels already carry observed = 1)
        }
    }
}

int check_variant2(tsl_tsdf* m)
{
    TSL_REQUIRE(m->F.max_frame_bricks <= 4096 && m->P.max_steps_f < (float)(1 << SEG_J_BITS) && m->F.max_points < (1 << STG_RAY_BITS) && m->nb3 < (1 << 24) && 3 * m->pcl_bits <= H_BLOCK_SHIFT,
                "variant 2: ray too long / too many points / too many bricks for the segment key (use variant 1)");
    return TSL_OK;
}

int launch_segments(tsl_tsdf* m, const BatchDev& B, const FrameParams* hp, int total, hipStream_t st)
{
    const FrameParams& P = hp[0];
    if (P.variant != 2) return TSL_OK;
    const int iblocks = (int)(((int64_t)total * P.split + 255) / 256);
    prof_begin(m, TSL_K_SEGMENTS, st);
    if (P.group) hipLaunchKernelGGL(k_segments<true>, dim3(iblocks, B.n), dim3(256), 0, st, m->M, B);
    else hipLaunchKernelGGL(k_segments<false>, dim3(iblocks, B.n), dim3(256), 0, st, m->M, B);
    prof_end(m, st);
    // the unit limit is quoted for a full batch; a shorter batch (one frame when something reads the map after every frame) scales it:
    // a unit is walked by one workgroup frame after frame, and with few frames a long unit is just a long serial item
    int unit_max = m->unit_max <= m->unit_floor ? m->unit_max : std::max(m->unit_floor, (int)((long long)m->unit_max * B.n / TSL_NB));
    int unit_half = std::min(unit_max, m->unit_half <= 2048 ? m->unit_half : std::max(2048, (int)((long long)m->unit_half * B.n / TSL_NB)));
    if (P.seq && m->seq_impl) unit_max = unit_half = 1 << 30;      // sequential semantics: every brick of the batch is listed once, as a unit (k_seq_replay walks that list); no parts, no slab
    prof_begin(m, TSL_K_BIN, st);
    hipLaunchKernelGGL(k_plan, dim3((B.f[0].max_frame_bricks + 255) / 256, B.n), dim3(256), 0, st, m->M, B, m->chunks * m->wg * m->spt, unit_max, unit_half);
    hipLaunchKernelGGL(k_scatter, dim3(256, B.n), dim3(256), 0, st, B);
    prof_end(m, st);
    return TSL_OK;
}

// phase B of a batch, variant 2: the brick kernel.  kind 0 = one launch over the whole work list on the main stream; split launches:
// kind 2 = the parts on the batch's phase-A stream `st` (grid: `pgrid` percent of the slots), kind 1 = the units on the main stream.
int launch_brick(tsl_tsdf* m, const BatchDev& B, const FrameParams& P, int kind, hipStream_t st, hipEvent_t start, hipEvent_t stop)
{
    // resident workgroups: one 512-thread one per CU (82 KiB of LDS, textured 98 KiB), or two 256-thread ones (74 KiB each; textured 90 KiB: one).
    // With start / stop events the launch goes through hipExtLaunchKernelGGL, which records them in the dispatch itself.
    // A batch of one or two frames (the per-frame ESDF hook flushes after every frame) has ~500 items: half the grid, one workgroup per CU,
    // leaves LDS for what runs beside it (the ESDF rounds of the frame before: +2 % in bench.py --config 4) and costs the batch nothing.
    const int bg = kind == 2 ? m->pgrid : (B.n <= 2 ? (m->bgrid + 1) / 2 : (kind == 1 ? m->ugrid : m->bgrid));
#define TSL_LAUNCH_IB(TEXV, FD) do { \
        if (m->wg == 512 && m->spt == 2 && !TEXV) hipExtLaunchKernelGGL((k_integrate_batch<false, FD, 512, 2, 4>), dim3((2 * m->ncu * bg + 99) / 100), dim3(512), 0, st, start, stop, 0, m->M, B, kind); \
        else if (m->wg == 512) hipExtLaunchKernelGGL((k_integrate_batch<TEXV, FD, 512, 4, 2>), dim3((m->ncu * bg + 99) / 100), dim3(512), 0, st, start, stop, 0, m->M, B, kind); \
        else hipExtLaunchKernelGGL((k_integrate_batch<TEXV, FD, 256, 4, (TEXV ? 1 : 2)>), dim3(((TEXV ? 1 : 2) * m->ncu * bg + 99) / 100), dim3(256), 0, st, start, stop, 0, m->M, B, kind); } while (0)
    if (P.tex) { if (P.fastdiv) TSL_LAUNCH_IB(true, true); else TSL_LAUNCH_IB(true, false); }
    else { if (P.fastdiv) TSL_LAUNCH_IB(false, true); else TSL_LAUNCH_IB(false, false); }
#undef TSL_LAUNCH_IB
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}
// the batch's heavy bricks: parts -> map (the count lives on the device: the grid covers what a batch can list, idle workgroups leave at once)
int launch_slab_apply(tsl_tsdf* m, const BatchDev& B, const FrameParams& P)
{
    prof_begin(m, TSL_K_FINALIZE);
    if (P.tex) hipLaunchKernelGGL(k_apply_slab<true>, dim3(16 * m->ncu), dim3(256), 0, m->stream_, m->M, B);
    else hipLaunchKernelGGL(k_apply_slab<false>, dim3(16 * m->ncu), dim3(256), 0, m->stream_, m->M, B);
    prof_end(m);
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}
int launch_apply_batch(tsl_tsdf* m, const BatchDev& B, const FrameParams& P, hipEvent_t start, hipEvent_t stop)
{
    const int rc = launch_brick(m, B, P, 0, m->stream_, start, stop);
    return rc ? rc : launch_slab_apply(m, B, P);
}

int launch_apply(tsl_tsdf* m, FSet& S, int total)
{
    FrameParams& P = m->P;
    FrameDev& F = S.F;
    const int iblocks = (int)(((int64_t)total * P.split + 255) / 256);
    prof_begin(m, TSL_K_INTEGRATE);
    if (P.variant == 1) hipLaunchKernelGGL(k_integrate<1>, dim3(iblocks), dim3(256), 0, m->stream_, m->M, F, (const FrameParams*)S.Pd);
    else hipLaunchKernelGGL(k_integrate<0>, dim3(iblocks), dim3(256), 0, m->stream_, m->M, F, (const FrameParams*)S.Pd);
    prof_end(m);
    prof_begin(m, TSL_K_FINALIZE);
    hipLaunchKernelGGL(k_finalize, dim3(1024), dim3(256), 0, m->stream_, m->M, F, (const int*)nullptr, (const int*)nullptr);
    prof_end(m);
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}

}  // namespace tsl
