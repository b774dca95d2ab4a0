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

namespace tsl {

#ifdef TSL_TIMING
// developer timing: every wave stores raw timestamps (plain stores, no atomics) at dbg[wave*16 + k]
#define TSL_T0() const int _wv = (blockIdx.x * (int)blockDim.x + (int)threadIdx.x) >> 6
#define TSL_TICK(F, k) do { long long _n = wall_clock64(); if (lane_id() == 0 && _wv < 16384) (F).dbg[_wv * 16 + (k)] = _n; } while (0)
#else
#define TSL_T0() do {} while (0)
#define TSL_TICK(F, k) do {} while (0)
#endif

// segment key (variants 0/1 staging): [0,6) step count  [6,18) first step  [18,42) ray id  [42,58) frame slot of the brick
#define SEG_CNT_BITS 6
#define SEG_J_BITS   12
#define SEG_RAY_BITS 24
#define SEG_SLOT_SHIFT (SEG_CNT_BITS + SEG_J_BITS + SEG_RAY_BITS)
#define SEG_MAX_CNT 63
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
__device__ __forceinline__ int next_axis_event(float d, float T, const FrameParams& P, float t_over_vs, int j, int jb, int cell, int h, int N, int nb)
{
    if (j >= jb) return jb + 1;
    int bound;                       // first coordinate that belongs to the next cell in the direction of travel
    bool up;
    if (d > 0.0f) { if (cell >= nb) return jb + 1; up = true; bound = (cell < 0 ? 0 : min((cell + 1) * 16, N)) - h; }
    else if (d < 0.0f) { if (cell < 0) return jb + 1; up = false; bound = (cell >= nb ? N - 1 : cell * 16 - 1) - h; }
    else return jb + 1;
    const float est = ((float)bound + (up ? -0.5f : 0.5f) - t_over_vs) / d;      // real-valued crossing step
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
#define STG_RAY_BITS 22
#define SEG_RAY_SLOTS 16        // private segment slots per ray (split evenly over its lanes); further segments are appended behind them
#define SEG_LH_LOG2 9
#define SEG_LH (1 << SEG_LH_LOG2)
#define STG_B_SHIFT (SEG_CNT_BITS + SEG_J_BITS + STG_RAY_BITS)
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
        ok = false;
        if (r < nrays && sub == 0) {
            sl = F.act[r];
            gn = F.hcnt[sl];
            if (gn > GROUP_SMALL) big = true;
            else {
                const uint32_t* ids = F.plist + F.hoff[sl];
                PixAcc A = {};
                long long last = -1;
                for (int k = 0; k < gn; ++k) {                                   // next pixel in raster order
                    uint32_t best = 0xffffffffu;
                    for (int q = 0; q < gn; ++q) { const uint32_t v = ids[q]; if ((long long)v > last && v < best) best = v; }
                    if (k == 0) first = best;
                    acc_pixel(P, F, best, A);
                    last = (long long)best;
                }
                ok = finish_ray(P, F, A, first, &rec, &nsteps);
            }
        }
        for (unsigned long long bm = __ballot(big); bm; bm &= bm - 1ull) {       // crowded voxels: the whole wave selects the next id
            const int src = (int)__builtin_ctzll(bm);
            const int n = __shfl(gn, src);
            const uint32_t* gid_list = F.plist + F.hoff[__shfl(sl, src)];
            PixAcc A = {};
            long long last = -1; uint32_t f0 = 0;
            for (int k = 0; k < n; ++k) {
                uint32_t best = 0xffffffffu;
                for (int q = lane; q < n; q += 64) { const uint32_t v = gid_list[q]; if ((long long)v > last && v < best) best = v; }
                best = wave_min_u32(best);
                if (k == 0) f0 = best;
                acc_pixel(P, F, best, A);
                last = (long long)best;
            }
            uint4 rc; int ns = 0;
            const bool k2 = finish_ray(P, F, A, f0, &rc, &ns, lane == src);
            if (lane == src) { rec = rc; nsteps = ns; ok = k2; first = f0; }
        }
        if (r < nrays && sub == 0) {
            F.rayA[r] = rec; F.rayFirst[r] = first;
            if (F.hwide) reinterpret_cast<unsigned long long*>(F.hkey)[sl] = ~0ull; else reinterpret_cast<uint32_t*>(F.hkey)[sl] = ~0u;
            F.hcnt[sl] = 0; F.hfill[sl] = 0;                                     // the table is empty again for the next frame of this set
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
        const float tv[3] = { P.T[0] / P.vs, P.T[1] / P.vs, P.T[2] / P.vs };
        const int hh[3] = { M.hN, M.hN, M.hNz }, NN[3] = { M.N, M.N, M.Nz }, nb[3] = { M.nbx, M.nbx, M.nbz };
        if (ja <= jb) {
            int cell[3], ev[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                cell[a] = axis_cell(axis_coord(d[a], P.T[a], P, ja), hh[a], NN[a], nb[a]);
                ev[a] = next_axis_event(d[a], P.T[a], P, tv[a], ja, jb, cell[a], hh[a], NN[a], nb[a]);
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
                    cell[a] = axis_cell(axis_coord(d[a], P.T[a], P, j), hh[a], NN[a], nb[a]);
                    ev[a] = next_axis_event(d[a], P.T[a], P, tv[a], j, jb, cell[a], hh[a], NN[a], nb[a]);
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
    if (lane_id() == 0) {
        if (n_ok) atomic_add_i64(&F.stats->steps, n_ok);
        if (n_oob) atomic_add_i64(&F.stats->steps_oob, n_oob);
    }
}

// K4b: lay out the frame's active bricks (listed by k_segments): a range of the sorted segment array per brick and its
// integrate parts (a brick with more than PART_SEGS segments is integrated by several workgroups).  Neither has to be in
// any particular order, so block-aggregated reservations replace a prefix scan.  Parts go to three tables by length, so
// k_integrate_bricks starts the long ones first and its tail is made of short ones.
// part = { first segment, segments | parts of the brick << 16, pool index of the brick (claimed here, on its first touch ever), active rank }
__device__ __forceinline__ int part_class(int per, int psegs) { return per * 8 >= psegs * 5 ? 0 : (per * 4 >= psegs ? 1 : 2); }
__global__ void __launch_bounds__(256) k_plan(MapDev M, BatchDev B, int psegs)
{
    if ((int)blockIdx.y >= B.n) return;
    const FrameDev& F = B.f[blockIdx.y];
    const int listed = F.counters[1];
    const int nact = min(listed, F.max_frame_bricks);
    if ((int)blockIdx.x * 256 >= nact && blockIdx.x) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int v = 0, b = 0, np = 0, cls = 2;
    if (i < nact) {
        b = F.act_b[i];
        v = F.bhist[b];
        np = (v + psegs - 1) / psegs;
        cls = np ? part_class((v + np - 1) / np, psegs) : 2;
    }
    const int off = block_reserve_n(&F.counters[3], v);
    int p0 = 0;
    for (int c = 0; c < 3; ++c) { const int q = block_reserve_n(&F.counters[8 + c], cls == c ? np : 0); if (cls == c) p0 = q; }
    if (i < nact) {
        F.boffset[b] = off;
        const int per = np ? (v + np - 1) / np : 0;
        int4* tab = F.part_tab + (size_t)cls * F.part_cap;
        const int pool = np ? pool_claim<false>(M, B.p[blockIdx.y]->slot, b) : -1;      // < 0: pool exhausted (reported through M.err), the parts are skipped
        for (int k = 0; k < np; ++k) {
            const int pos = k * per, n = min(v, pos + per) - pos;
            if (p0 + k < F.part_cap) tab[p0 + k] = make_int4(off + pos, n | (np << 16), pool, i); else frame_fail(M, F, 2);
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
__device__ __forceinline__ int acc_swz5(int l) { return l ^ (((l >> 8) ^ (l >> 5)) & 31); }
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

#define FLUSH_CHUNK 8
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
// k_integrate_bricks: the same part arithmetic as k_integrate_bricks2, as a persistent, software-pipelined kernel.
// What bounded v2 was not the walk but the dependent device-memory round trips around it (part entry -> segment keys -> ray
// records before, rows / stores after: 1-2 us each at 8 waves per CU) and the two dispatch rounds of ~670 parts on 512 slots.
// Here 2 workgroups per CU stay resident and take parts in serpentine order over the cost-ordered part list (long, medium,
// short: the workgroup with the longest part of a tier gets the shortest of the next).  While a part is walked, the entry and the
// keys of the workgroup's next part are in flight; after the walk the next part's keys are length-sorted in their own 8 KiB of LDS
// and its ray records requested, and only then the current part is flushed -- every load has a phase of useful work to hide behind.
// =====================================================================================================
__device__ __forceinline__ int4 part_entry(const FrameDev& F, int rank, int nA, int nB)
{
    return rank < nA ? F.part_tab[rank] : (rank < nA + nB ? F.part_tab[F.part_cap + rank - nA] : F.part_tab[2 * (size_t)F.part_cap + rank - nA - nB]);
}
// rank of the t-th part of workgroup w of G in serpentine order
__device__ __forceinline__ int serp_rank(int t, int w, int G) { return t * G + ((t & 1) ? G - 1 - w : w); }

template <bool TEX, bool FASTDIV, int NT>
__global__ void __launch_bounds__(NT, (TEX && NT == 256) ? 1 : 2) k_integrate_bricks(MapDev M, FrameDev F, const FrameParams* __restrict__ Pp)
{
    // NT threads walk a part in chunks of CSEGS = 4 * NT segments (4 per thread); a part may hold several chunks (k_plan's psegs), so
    // a brick with up to psegs segments is integrated by one workgroup and never merged through HBM.  NT = 256: two workgroups per CU;
    // NT = 512: one (8 waves on one brick: half the walk latency per brick, half as many bricks in flight).
    constexpr int CSEGS = 4 * NT, SPT = 4, VPT = TSL_BRK3 / NT, CH = VPT < FLUSH_CHUNK ? VPT : FLUSH_CHUNK;
    const FrameParams& P = *Pp;
    __shared__ unsigned long long s_num[TSL_BRK3];              // 32 KiB
    __shared__ unsigned long long s_den[TSL_BRK3];              // 32 KiB
    __shared__ unsigned long long s_keys[CSEGS];                // 8 / 16 KiB: length sort of the NEXT chunk's keys while the planes hold the current sums
    __shared__ uint32_t s_win[TEX ? TSL_BRK3 : 1];              // texture: colour winner per voxel (first pixel of the ray + 1)
    __shared__ int s_bin[64];
    __shared__ int s_last;
    const int nA = min(F.counters[8], F.part_cap), nB = min(F.counters[9], F.part_cap), nC = min(F.counters[10], F.part_cap);
    const int nparts = (F.counters[11] != 0) ? 0 : nA + nB + nC;          // nothing is integrated when the frame overflowed its scratch
    const int G = gridDim.x, w = blockIdx.x;
    const StepK K = { P.vs, P.rvs, P.T[0], P.T[1], P.T[2], M.hN, M.hNz };
    long long uniq = 0;
    TSL_T0();
    {   // restore the "all zero between uses" invariant of this set's per-brick histogram / cursor
        const int nact = min(F.counters[1], F.max_frame_bricks);
        for (int i = blockIdx.x * NT + threadIdx.x; i < nact; i += gridDim.x * NT) { const int b = F.act_b[i]; F.bhist[b] = 0; F.bcursor[b] = 0; }
    }
    int t = 0, c = 0;                                          // t-th part of this workgroup, chunk c of it
    if (serp_rank(0, w, G) >= nparts) return;
    int4 ptc = part_entry(F, serp_rank(0, w, G), nA, nB);
    int4 ptn = make_int4(0, 0, -1, 0);
    if (serp_rank(1, w, G) < nparts) ptn = part_entry(F, serp_rank(1, w, G), nA, nB);

    unsigned long long kk[SPT]; uint4 recs[SPT]; uint32_t wids[SPT];
    // length-sort `nseg` keys (this thread holds k[q] = key q*NT+tid) through s_keys / s_bin, deal them out in alternating directions
    // and request the ray records of the dealt keys.  Counting sort by step count, descending: the lanes of a wave walk segments of
    // (almost) equal length and every thread gets about the same number of steps.  Contains 3 barriers; s_bin must be zero on entry.
#define TSL_SORT_DEAL(KIN, NSEG)                                                                                        \
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
                recs[q] = F.rayA[r];                                                                                    \
                wids[q] = TEX ? F.rayFirst[r] + 1u : 0u;                                                                \
            }                                                                                                           \
        }                                                                                                               \
        if (threadIdx.x < 64) s_bin[threadIdx.x] = 0;              /* ordered against the next use by the barriers in between */ \
    }

    {   // prologue: the first chunk's keys, planes cleared, sort, ray records
        unsigned long long k0[SPT];
        const int nseg0 = min(ptc.y & 0xffff, CSEGS);
#pragma unroll
        for (int q = 0; q < SPT; ++q) { const int i = q * NT + threadIdx.x; k0[q] = i < nseg0 ? F.seg_sorted[ptc.x + i] : 0ull; }
        ulonglong2* zn = reinterpret_cast<ulonglong2*>(s_num); ulonglong2* zd = reinterpret_cast<ulonglong2*>(s_den);
        for (int i = threadIdx.x; i < TSL_BRK3 / 2; i += NT) { zn[i] = make_ulonglong2(0ull, 0ull); zd[i] = make_ulonglong2(0ull, 0ull); }
        if (TEX) for (int i = threadIdx.x; i < TSL_BRK3; i += NT) s_win[i] = 0u;
        if (threadIdx.x < 64) s_bin[threadIdx.x] = 0;
        __syncthreads();
        TSL_SORT_DEAL(k0, nseg0)
    }
    uint32_t old[VPT];
    for (;;) {
        TSL_TICK(F, 0);
        const int nseg_part = ptc.y & 0xffff, np = ptc.y >> 16, p = ptc.z, rk = ptc.w;
#ifdef TSL_TIMING
        long long* const _rec = F.dbg + 131072 + (size_t)blockIdx.x * 64 + (size_t)(t < 8 ? t : 7) * 8;
        if (threadIdx.x == 0 && c == 0) { _rec[0] = wall_clock64(); _rec[1] = nseg_part; _rec[2] = np; }
#endif
        const int nseg = min(CSEGS, nseg_part - c * CSEGS);
        const bool last = (c + 1) * CSEGS >= nseg_part;               // last chunk of the part: flush after the walk
        const bool whole = np == 1;
        // the next unit of work: the next chunk of this part, or the first chunk of the workgroup's next part
        const bool has_next = !last || serp_rank(t + 1, w, G) < nparts;
        const int4 ptx = last ? ptn : ptc;
        const int cx = last ? 0 : c + 1;
        // requests that ride under the walk: the entry of the part after next, the next chunk's keys, this brick's rows
        int4 ptn2 = make_int4(0, 0, -1, 0);
        if (last && serp_rank(t + 2, w, G) < nparts) ptn2 = part_entry(F, serp_rank(t + 2, w, G), nA, nB);
        unsigned long long kn[SPT];
        const int nsegn = has_next ? min(CSEGS, (ptx.y & 0xffff) - cx * CSEGS) : 0;
#pragma unroll
        for (int q = 0; q < SPT; ++q) { const int i = q * NT + threadIdx.x; kn[q] = i < nsegn ? F.seg_sorted[ptx.x + cx * CSEGS + i] : 0ull; }
        if (!TEX && c == 0 && whole && p >= 0) {
            const uint32_t* twr = M.tw + (size_t)p * TSL_BRK3;
#pragma unroll
            for (int q = 0; q < VPT; ++q) old[q] = twr[q * NT + threadIdx.x];
        }
        TSL_TICK(F, 1);
        // ---- walk ----
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int i = q * NT + ((q & 1) ? NT - 1 - (int)threadIdx.x : (int)threadIdx.x);
            if (i >= nseg) continue;
#ifdef TSL_EXP_NOWALK
            if (nseg >= 0) continue;
#endif
            const unsigned long long key = kk[q];
            const int cnt = (int)(key & ((1u << SEG_CNT_BITS) - 1)), j0 = (int)((key >> SEG_CNT_BITS) & ((1u << SEG_J_BITS) - 1));
            const RayRegs R = make_ray(recs[q], 0, P);
            const uint32_t wid = wids[q];
            // lanes start at different offsets inside their (equally long) segments: rays that enter a brick together -- all
            // of them next to the sensor -- would otherwise hit the same few voxels in the same iteration
            int off = ((int)(threadIdx.x & 63u) * cnt) >> 6;
            for (int s = 0; s < cnt; s += 2) {
                const int ja = j0 + off; off = (off + 1 == cnt) ? 0 : off + 1;
                const int jb = j0 + off; off = (off + 1 == cnt) ? 0 : off + 1;
                int la, lb; float qa, qb;
                step_eval<FASTDIV>(R, K, ja, &la, &qa);
                step_eval<FASTDIV>(R, K, jb, &lb, &qb);
                long long na = (long long)(int)qa, nb = (long long)(int)qb;
                if (__builtin_expect(__any(!(fabsf(qa) < 2147483648.0f) || !(fabsf(qb) < 2147483648.0f)), 0)) { na = __float2ll_rn(qa); nb = __float2ll_rn(qb); }
                atomicAdd(&s_num[la], (unsigned long long)na);
                atomicAdd(&s_den[la], (unsigned long long)R.qden);
                if (TEX) atomicMax(&s_win[la], wid);                                           // dense_tsdf.py:268-269, order-free winner
                // the second step of an odd segment's last pair adds zeros to a voxel of the brick (no effect, no branch)
                const bool vb = s + 1 < cnt;
                atomicAdd(&s_num[lb], (unsigned long long)(vb ? nb : 0ll));
                atomicAdd(&s_den[lb], (unsigned long long)(vb ? R.qden : 0ll));
                if (TEX) atomicMax(&s_win[lb], vb ? wid : 0u);
            }
        }
        TSL_TICK(F, 2);
        __syncthreads();
        TSL_TICK(F, 3);
#ifdef TSL_TIMING
        if (threadIdx.x == 0 && last) _rec[3] = wall_clock64();
#endif
        // ---- the next chunk's keys are here: sort them and request its ray records, then flush under that latency ----
        if (has_next) TSL_SORT_DEAL(kn, nsegn)
        if (!last) { ++c; continue; }
        if (p >= 0 && threadIdx.x == 0) M.touch[p] = 1;               // the brick's TSDF changes in this frame (incremental ESDF)
        if (p >= 0 && whole) {
            uint32_t* tw = M.tw + (size_t)p * TSL_BRK3;
            int8_t* obs = M.obs + (size_t)p * TSL_BRK3;
            if (TEX) {
#pragma unroll
                for (int q = 0; q < VPT; ++q) old[q] = tw[q * NT + threadIdx.x];
            }
#pragma unroll
            for (int h = 0; h < VPT; h += CH) {                  // CH voxels at a time, no branch between them
                long long qn[CH], qd[CH]; uint32_t nv[CH]; bool small = true;
#pragma unroll
                for (int q = 0; q < CH; ++q) { const int ls = acc_swz5((h + q) * NT + threadIdx.x); qn[q] = (long long)s_num[ls]; qd[q] = (long long)s_den[ls]; small = small && fits_i32(qn[q]) && fits_i32(qd[q]); }
                apply_chunk<CH>(old + h, qn, qd, nv, __all(small));
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const int l = (h + q) * NT + threadIdx.x;
                    if (qd[q] != 0) {
                        tw[l] = nv[q];
                        if ((old[h + q] >> 16) == 0u) obs[l] = 1;       // W == 0 <=> never integrated; imported voxels already carry observed = 1
                        if (TEX) reinterpret_cast<uint2*>(M.col)[(size_t)p * TSL_BRK3 + l] = F.colpix[s_win[acc_swz5(l)] - 1u];
                        ++uniq;
                    }
                }
            }
        } else if (p >= 0) {
            // brick split over `np` workgroups: add the partial sums into the brick's HBM scratch slab; the last workgroup
            // to arrive (arrival ticket, agent-scope release/acquire) applies them and leaves the slab zeroed.
            unsigned long long* acc = F.acc + (size_t)rk * (TSL_BRK3 * 2);
#pragma unroll
            for (int q = 0; q < VPT; ++q) {
                const int l = q * NT + threadIdx.x, ls = acc_swz5(l);
#ifdef TSL_EXP_NOSPLITFLUSH
                const unsigned long long d = 0ull;
#else
                const unsigned long long d = s_den[ls];
#endif
                if (d != 0ull) {
                    __hip_atomic_fetch_add(acc + l * 2, s_num[ls], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(acc + l * 2 + 1, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (TEX) atomicMax(F.accw + (size_t)rk * TSL_BRK3 + l, s_win[ls]);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int tk = __hip_atomic_fetch_add(&F.ticket[rk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = (tk == np - 1) ? 1 : 0;
                if (s_last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); F.ticket[rk] = 0; }
            }
            __syncthreads();
            if (s_last) {
                ulonglong2* acc2 = reinterpret_cast<ulonglong2*>(acc);
                uint32_t* tw = M.tw + (size_t)p * TSL_BRK3;
                int8_t* obs = M.obs + (size_t)p * TSL_BRK3;
                // the sums were produced by L2 atomics of other CUs: read them at L2 as well, CH voxels in flight per thread
#pragma unroll
                for (int h = 0; h < VPT; h += CH) {
                    long long qn[CH], qd[CH]; uint32_t oldv[CH], nv[CH]; bool small = true;
#pragma unroll
                    for (int q = 0; q < CH; ++q) {
                        const int l = (h + q) * NT + threadIdx.x;
                        qn[q] = (long long)__hip_atomic_load(&acc[l * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        qd[q] = (long long)__hip_atomic_load(&acc[l * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        oldv[q] = tw[l];
                    }
#pragma unroll
                    for (int q = 0; q < CH; ++q) small = small && fits_i32(qn[q]) && fits_i32(qd[q]);
                    apply_chunk<CH>(oldv, qn, qd, nv, __all(small));
#pragma unroll
                    for (int q = 0; q < CH; ++q) {
                        const int l = (h + q) * NT + threadIdx.x;
                        if (qd[q] != 0) {
                            tw[l] = nv[q];
                            if ((oldv[q] >> 16) == 0u) obs[l] = 1;
                            acc2[l] = make_ulonglong2(0ull, 0ull);
                            if (TEX) {
                                uint32_t* wv = F.accw + (size_t)rk * TSL_BRK3 + l;
                                const uint32_t wsel = __hip_atomic_load(wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                reinterpret_cast<uint2*>(M.col)[(size_t)p * TSL_BRK3 + l] = F.colpix[wsel - 1u];
                                *wv = 0u;
                            }
                            ++uniq;
                        }
                    }
                }
            }
        }
        TSL_TICK(F, 4);
#ifdef TSL_TIMING
        if (threadIdx.x == 0) _rec[4] = wall_clock64();
        if (lane_id() == 0 && _wv < 16384) { F.dbg[_wv * 16 + 10] = nseg_part; F.dbg[_wv * 16 + 11] = whole; F.dbg[_wv * 16 + 12] = t; }
#endif
        if (!has_next) break;
        __syncthreads();                                                  // every thread has read the sums of the finished part
        {
            ulonglong2* zn = reinterpret_cast<ulonglong2*>(s_num); ulonglong2* zd = reinterpret_cast<ulonglong2*>(s_den);
            for (int i = threadIdx.x; i < TSL_BRK3 / 2; i += NT) { zn[i] = make_ulonglong2(0ull, 0ull); zd[i] = make_ulonglong2(0ull, 0ull); }
            if (TEX) for (int i = threadIdx.x; i < TSL_BRK3; i += NT) s_win[i] = 0u;
        }
        __syncthreads();
        ++t; c = 0; ptc = ptn; ptn = ptn2;
    }
#undef TSL_SORT_DEAL
    uniq = wave_sum_ll(uniq);
    if (lane_id() == 0 && uniq) atomic_add_i64(&F.stats->unique, uniq);
}

int check_variant2(tsl_tsdf* m)
{
    TSL_REQUIRE(m->F.max_frame_bricks <= 4096 && m->P.max_steps_f < (float)(1 << SEG_J_BITS) && m->F.max_points < (1 << STG_RAY_BITS) && m->nb3 < (1 << 24),
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
    prof_begin(m, TSL_K_BIN, st);
    hipLaunchKernelGGL(k_plan, dim3((B.f[0].max_frame_bricks + 255) / 256, B.n), dim3(256), 0, st, m->M, B, m->chunks * (m->wg == 512 ? 2048 : 1024));
    hipLaunchKernelGGL(k_scatter, dim3(256, B.n), dim3(256), 0, st, B);
    prof_end(m, st);
    return TSL_OK;
}

int launch_apply(tsl_tsdf* m, FSet& S, int total)
{
    FrameParams& P = m->P;
    FrameDev& F = S.F;
    if (P.variant == 2) {
        prof_begin(m, TSL_K_INTEGRATE);
        const FrameParams* Pd = (const FrameParams*)S.Pd;
        // resident workgroups: two 256-thread ones per CU (74 KiB of LDS each; textured 90 KiB: one), or one 512-thread one
#define TSL_LAUNCH_IB3(TEXV, FD) do { if (m->wg == 512) hipLaunchKernelGGL((k_integrate_bricks<TEXV, FD, 512>), dim3(m->ncu), dim3(512), 0, m->stream_, m->M, F, Pd); \
                                      else hipLaunchKernelGGL((k_integrate_bricks<TEXV, FD, 256>), dim3((TEXV ? 1 : 2) * m->ncu), dim3(256), 0, m->stream_, m->M, F, Pd); } while (0)
        if (P.tex) { if (P.fastdiv) TSL_LAUNCH_IB3(true, true); else TSL_LAUNCH_IB3(true, false); }
        else { if (P.fastdiv) TSL_LAUNCH_IB3(false, true); else TSL_LAUNCH_IB3(false, false); }
#undef TSL_LAUNCH_IB3
        prof_end(m);
    } else {
        const int iblocks = (int)(((int64_t)total * P.split + 255) / 256);
        prof_begin(m, TSL_K_INTEGRATE);
        if (P.variant == 1) hipLaunchKernelGGL(k_integrate<1>, dim3(iblocks), dim3(256), 0, m->stream_, m->M, F, (const FrameParams*)S.Pd);
        else hipLaunchKernelGGL(k_integrate<0>, dim3(iblocks), dim3(256), 0, m->stream_, m->M, F, (const FrameParams*)S.Pd);
        prof_end(m);
        prof_begin(m, TSL_K_FINALIZE);
        hipLaunchKernelGGL(k_finalize, dim3(1024), dim3(256), 0, m->stream_, m->M, F, (const int*)nullptr, (const int*)nullptr);
        prof_end(m);
    }
    TSL_HIP(hipGetLastError());
    return TSL_OK;
}

}  // namespace tsl
