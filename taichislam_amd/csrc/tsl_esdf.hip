// tsl_esdf.hip -- incremental ESDF of the active submap from its TSDF.
//
// The reference's ESDF (taichi_slam/mapping/dense_esdf.py:228-333, reference root) is a legacy module that cannot be constructed at
// HEAD and whose queue propagation is incomplete; it serves as the DEFINITION:
//   * voxels with |TSDF| < gamma are "fixed": ESDF := TSDF                                                    (:228-230, :313-317)
//   * every other observed voxel starts at sign(TSDF) * max_dist                                              (:325, :329)
//   * magnitudes are lowered through the 26-neighbourhood, edge cost |dir| * voxel (:282-297), only between voxels on the same side
//     of the surface (:289, :295); the per-frame hook (:400-402) only revisits what the frame changed ("updated_TSDF", raise / lower
//     queues :100-102, :255-333).
// The least fixed point of that relaxation is unique, which is what makes an incremental update checkable: it must equal the full
// recompute bit for bit (tests/test_esdf_gpu.py does so after every frame of a stream).
//
// Incremental update.  Every kernel that writes TSDF values marks the brick in MapDev.touch.  An update
//   1. collects the marked ("dirty") bricks of the active submap,
//   2. dilates them by ceil(max_dist / 16 voxels) bricks -> region R.  A voxel outside R is farther than max_dist (Chebyshev, hence
//      in path cost) from every changed voxel, so neither its value nor any path that determines it can involve one: it keeps its
//      value and serves as a boundary condition,
//   3. re-initialises R (flags from the current TSDF; band := |TSDF|, the rest := max_dist -- this is what lets values RISE),
//   4. relaxes R in rounds.  A round stages every brick on its work list in LDS with its one-voxel halo (18^3 values + flags) and runs a
//      push relaxation driven by per-voxel active bits -- band voxels (first visit) and halo voxels push, a voxel that improves becomes
//      active -- so the work follows the wavefront instead of sweeping 4096 voxels x 26 neighbours per pass.  A brick whose boundary
//      layer improved puts the neighbours that see it on the next round's list (device-side, deduplicated).  Rounds are plain launches on
//      the handle's stream; the host launches a batch sized by max_dist, synchronises ONCE and reads the counters.
//      (A single persistent kernel with an asynchronous brick queue was tried first: exact, but with no ordering between bricks each was
//      relaxed ~23 times -- 5.2 ms per update at 512^3.  Rounds keep the wavefront order: ~2-3 relaxations per brick.)
// The first update, an update after reset() / import / fusion, or one with different parameters is the same procedure with R = all bricks.
#include "tsl_tsdf.hpp"

namespace tsl {

#define ESDF_T 18
#define ESDF_T3 (ESDF_T * ESDF_T * ESDF_T)
#define EF_NODE 1
#define EF_NEG 2
#define EF_FIXED 4
#define ESDF_PAD (ESDF_T * ESDF_T + ESDF_T + 1)
#define ESDF_QCAP 256          // queue entries per wave and hand-over (four per lane)
#ifndef ESDF_SWEEPS
#define ESDF_SWEEPS 1
#endif

#ifdef TSL_TIMING
// developer timing: thread 0 of every relaxation adds the clock ticks (100 MHz) of its phases to E.ctr64[k]
#define ESDF_TICK(k) do { if (threadIdx.x == 0) { const long long _n = wall_clock64(); atomicAdd(&E.tm[k], (unsigned long long)(_n - _t)); _t = _n; } } while (0)
#else
#define ESDF_TICK(k) do {} while (0)
#endif
struct EsdfDev {
    float* mag;                // [max_bricks][4096] magnitude
    uint8_t* fl;               // [max_bricks][4096] EF_* of the voxel at the last (re)initialisation
    uint8_t* region;           // [max_bricks] 1: brick is part of this update's region, 2: relaxed once already
    int* stamp;                // [max_bricks] last round the brick was put on a work list for (dedupe)
    int* dirty;                // [max_bricks] dirty list
    uint32_t* note;            // [2][max_bricks] per round parity: bit q = neighbour q (of 27) changed its boundary layer in the previous round
    int* work;                 // [3][max_bricks] work lists of rounds k, k+1, k+2 (mod 3)
    int cap;                   // max_bricks
    unsigned long long* tm;    // developer timing (TSL_TIMING builds): ticks per phase, summed over relaxations
    int* ctr;                  // [0] dirty count [1] region count [2..4] work list lengths [5] brick relaxations [6] voxel pushes [7] rounds with work
};

// 1. dirty bricks of submap s (and, when `all`, every brick of it); touch marks are consumed
__global__ void __launch_bounds__(256) k_esdf_collect(MapDev M, EsdfDev E, int s, int all)
{
    // the brick count of this update: a snapshot of the pool counter copied into ctr[10] ahead of this kernel (on the device: the host
    // does not wait for the frames before it).  Phase A of frames queued AFTER the update may allocate bricks while it runs (their
    // phase A starts once this kernel has finished, esdf_gate): every kernel of the update ignores pool indices >= the snapshot, so a
    // brick that appears meanwhile -- owner / flags not written yet -- is simply not there for this update.
    const int nused = min(E.ctr[10], M.max_bricks);
    const int p = blockIdx.x * 256 + threadIdx.x;
    bool take = false;
    if (p < nused) {
        const bool mine = M.owner[p] / M.nb3 == s;
        if (mine) { take = all || M.touch[p] != 0; M.touch[p] = 0; }
        E.region[p] = 0; E.stamp[p] = -1; E.note[p] = 0u; E.note[E.cap + p] = 0u;
    }
    const int q = wave_reserve(&E.ctr[0], take);
    if (take) E.dirty[q] = p;
}

// class flags and band value of a voxel from its TSDF (what the relaxation depends on)
__device__ __forceinline__ void esdf_inputs(uint32_t obs_byte, uint32_t tw, float gamma, float max_dist, uint32_t* f, float* mg)
{
    *f = 0u; *mg = max_dist;
    if ((int8_t)obs_byte > 0) {
        const float t = h2f((h16)(tw & 0xffffu));
        *f = EF_NODE | (t < 0.0f ? EF_NEG : 0);
        if (fabsf(t) < gamma) { *f |= EF_FIXED; *mg = fabsf(t); }
    }
}

// 2. region = the bricks whose ESDF INPUTS changed, dilated by r bricks (existing bricks of the submap only).  A brick an integrate
//    kernel wrote to is only a candidate: most of what a frame touches is free space between the sensor and the surface, where the
//    TSDF value changes but neither the class (observed, sign, band membership) nor a band value does -- such a brick cannot change
//    any distance.  The stored flags / band values are those of the brick's last (re)initialisation.
__global__ void __launch_bounds__(256) k_esdf_dilate(MapDev M, EsdfDev E, int s, int r, int all, float gamma, float max_dist)
{
    const int nd = E.ctr[0], nsnap = min(E.ctr[10], M.max_bricks);
    const int side = 2 * r + 1, vol = side * side * side;
    for (int d = blockIdx.x; d < nd; d += gridDim.x) {
        const int pd = E.dirty[d];
        if (!all) {
            const size_t v = (size_t)pd * TSL_BRK3 + (size_t)threadIdx.x * 16;
            const uint4 ob = *reinterpret_cast<const uint4*>(M.obs + v);
            const uint4 fo = *reinterpret_cast<const uint4*>(E.fl + v);
            uint4 tw[4], mo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { tw[q] = reinterpret_cast<const uint4*>(M.tw + v)[q]; mo[q] = reinterpret_cast<const uint4*>(E.mag + v)[q]; }
            const uint32_t ow[4] = { ob.x, ob.y, ob.z, ob.w }, fw[4] = { fo.x, fo.y, fo.z, fo.w };
            const uint32_t tv[16] = { tw[0].x, tw[0].y, tw[0].z, tw[0].w, tw[1].x, tw[1].y, tw[1].z, tw[1].w, tw[2].x, tw[2].y, tw[2].z, tw[2].w, tw[3].x, tw[3].y, tw[3].z, tw[3].w };
            const uint32_t mv[16] = { mo[0].x, mo[0].y, mo[0].z, mo[0].w, mo[1].x, mo[1].y, mo[1].z, mo[1].w, mo[2].x, mo[2].y, mo[2].z, mo[2].w, mo[3].x, mo[3].y, mo[3].z, mo[3].w };
            bool changed = false;
#pragma unroll
            for (int z = 0; z < 16; ++z) {
                uint32_t f; float mg;
                esdf_inputs((ow[z >> 2] >> ((z & 3) * 8)) & 0xffu, tv[z], gamma, max_dist, &f, &mg);
                const uint32_t fs = (fw[z >> 2] >> ((z & 3) * 8)) & 0xffu;
                changed = changed || f != fs || ((f & EF_FIXED) && __float_as_uint(mg) != mv[z]);
            }
            if (!__syncthreads_or(changed)) continue;                 // (uniform: every thread of the workgroup leaves or stays)
        }
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&E.ctr[11], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int b = M.owner[pd] - s * M.nb3;
        const int bk = b % M.nbz, bj = (b / M.nbz) % M.nbx, bi = b / (M.nbz * M.nbx);
        for (int t = threadIdx.x; t < vol; t += 256) {
            const int i = bi + t / (side * side) - r, j = bj + (t / side) % side - r, k = bk + t % side - r;
            if (i < 0 || i >= M.nbx || j < 0 || j >= M.nbx || k < 0 || k >= M.nbz) continue;
            const int p = pool_lookup_ro(M, s, (i * M.nbx + j) * M.nbz + k);
            if (p >= 0 && p < nsnap) E.region[p] = 1;
        }
    }
}

// 3. (re)initialise the region's voxels; every brick of the region is on the work list of round 0
__global__ void __launch_bounds__(256) k_esdf_init(MapDev M, EsdfDev E, float gamma, float max_dist)
{
    const int nused = min(E.ctr[10], M.max_bricks);
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        if (E.region[p] != 1) continue;
        {   // a thread's 16 consecutive voxels: 1 + 4 wide loads, 1 + 4 wide stores
            const size_t v = (size_t)p * TSL_BRK3 + (size_t)threadIdx.x * 16;
            const uint4 ob = *reinterpret_cast<const uint4*>(M.obs + v);
            uint4 tw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) tw[q] = reinterpret_cast<const uint4*>(M.tw + v)[q];
            const uint32_t ow[4] = { ob.x, ob.y, ob.z, ob.w };
            const uint32_t tv[16] = { tw[0].x, tw[0].y, tw[0].z, tw[0].w, tw[1].x, tw[1].y, tw[1].z, tw[1].w, tw[2].x, tw[2].y, tw[2].z, tw[2].w, tw[3].x, tw[3].y, tw[3].z, tw[3].w };
            uint32_t fo[4] = { 0u, 0u, 0u, 0u }; float mo[16];
#pragma unroll
            for (int z = 0; z < 16; ++z) {
                uint32_t f; float mg;
                esdf_inputs((ow[z >> 2] >> ((z & 3) * 8)) & 0xffu, tv[z], gamma, max_dist, &f, &mg);
                fo[z >> 2] |= f << ((z & 3) * 8); mo[z] = mg;
            }
            *reinterpret_cast<uint4*>(E.fl + v) = make_uint4(fo[0], fo[1], fo[2], fo[3]);
#pragma unroll
            for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(E.mag + v)[q] = make_float4(mo[4 * q], mo[4 * q + 1], mo[4 * q + 2], mo[4 * q + 3]);
        }
        if (threadIdx.x == 0) {
            const int q = __hip_atomic_fetch_add(&E.ctr[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            E.work[q] = p; E.stamp[p] = 0;
            __hip_atomic_fetch_add(&E.ctr[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// 4. one relaxation round: every brick on this round's work list is staged in LDS with its one-voxel halo and relaxed to its local fixed
//    point; a brick whose boundary layer improved puts the neighbours that see it on the next round's list.  Rounds are separate launches
//    (the kernel boundary is the only synchronisation: no fences, no spinning); a launch whose list is empty returns at once.
//
//    LDS layout.  s_d holds the 18^3 tile with a margin of one plane + one row + one entry on both sides, so that a neighbour is entry
//    t + (dx*18 + dy)*18 + dz without any range check.  An entry of s_d is a TARGET word: side << 31 | magnitude bits for an interior,
//    observed, non-fixed voxel -- the only kind a push may lower -- and 0 for everything else (halo, fixed band, unobserved, margin).
//    With that encoding the 26 neighbour tests of a push need no flags: for a source on side sb the test "same side, is a target,
//    candidate is lower" is the single signed comparison (int)(word ^ sb) > (int)candidate (positive floats order like integers; a
//    target of the other side and the 0 of a non-target turn negative or stay 0).  Voxels that only ever push -- the halo and the fixed
//    band -- keep their values in s_hv / s_old (side << 31 | magnitude, 0 = unobserved); they push once, in the first pass.
__device__ __forceinline__ int esdf_halo_index(int tx, int ty, int tz)          // position of a halo entry in s_hv (inverse of the decode below)
{
    if (tx == 0 || tx == 17) return (tx / 17) * (ESDF_T * ESDF_T) + ty * ESDF_T + tz;
    if (ty == 0 || ty == 17) return 2 * ESDF_T * ESDF_T + (tx - 1) * (2 * ESDF_T) + (ty / 17) * ESDF_T + tz;
    return 2 * ESDF_T * ESDF_T + 16 * 2 * ESDF_T + (tx - 1) * 32 + (ty - 1) * 2 + tz / 17;
}
__global__ void __launch_bounds__(256, 3) k_esdf_round(MapDev M, EsdfDev E, int s, float vs, float max_dist, int round)
{
    constexpr int NH = ESDF_T3 - TSL_BRK3, HPER = (NH + 255) / 256;       // 1736 halo entries, 7 per thread
    __shared__ uint32_t s_dm[ESDF_T3 + 2 * ESDF_PAD];      // target words (see above)
    __shared__ __attribute__((aligned(16))) uint32_t s_old[TSL_BRK3];   // the brick's voxels as staged, brick order: side << 31 | magnitude, 0 = unobserved
    __shared__ uint32_t s_hv[HPER * 256];                  // the halo's values, same encoding
    __shared__ uint32_t s_a[(ESDF_T3 + 31) / 32 + 2];      // active bits: the entry pushes in the next pass
    __shared__ uint16_t s_q[4 * ESDF_QCAP];                // per wave: the entries it pushes next (compacted)
    __shared__ int s_nb[27];                       // pool index of the 27 bricks around (and including) this one, -1 = absent
    __shared__ int s_notify;
    uint32_t* const s_d = s_dm + ESDF_PAD;
    const int cur = round % 3, nxt = (round + 1) % 3, clr = (round + 2) % 3;
    const int n = E.ctr[2 + cur], nsnap = min(E.ctr[10], M.max_bricks);
    if (blockIdx.x == 0 && threadIdx.x == 0) { E.ctr[2 + clr] = 0; if (n) E.ctr[7] = round + 1; }      // list (round+2) was consumed in round-1
    if (n == 0) return;
    const float cost[4] = { 0.0f, 1.0f * vs, sqrtf(2.0f) * vs, sqrtf(3.0f) * vs };              // dense_esdf.py:286
    const int* list = E.work + (size_t)cur * E.cap;
    int* next = E.work + (size_t)nxt * E.cap;
    for (int i = threadIdx.x; i < ESDF_PAD; i += 256) { s_dm[i] = 0u; s_dm[ESDF_PAD + ESDF_T3 + i] = 0u; }
    for (int w = blockIdx.x; w < n; w += gridDim.x) {
#ifdef TSL_TIMING
        long long _t = wall_clock64();
#endif
        const int p = list[w];
        const int b = M.owner[p] - s * M.nb3;
        const int bk = b % M.nbz, bj = (b / M.nbz) % M.nbx, bi = b / (M.nbz * M.nbx);
        if (threadIdx.x < 27) {
            const int i = bi + (int)threadIdx.x / 9 - 1, j = bj + ((int)threadIdx.x / 3) % 3 - 1, k = bk + (int)threadIdx.x % 3 - 1;
            const int np = (i < 0 || i >= M.nbx || j < 0 || j >= M.nbx || k < 0 || k >= M.nbz) ? -1 : pool_lookup_ro(M, s, (i * M.nbx + j) * M.nbz + k);
            s_nb[threadIdx.x] = np < nsnap ? np : -1;                 // a brick allocated after the update's snapshot is not part of it
        }
        if (threadIdx.x == 0) s_notify = 0;
        for (int i = threadIdx.x; i < (ESDF_T3 + 31) / 32 + 2; i += 256) s_a[i] = 0u;
        // first relaxation in this update: the band voxels push, and so does every halo voxel; afterwards only the halo voxels of the
        // neighbours that changed their boundary layer since (the notification mask written for this round) push again
        const bool first = E.region[p] == 1;
        uint32_t* const my_note = E.note + (size_t)(round & 1) * E.cap + p;
        const uint32_t note = first ? ~0u : *my_note;
        __syncthreads();
        ESDF_TICK(0);
        if (threadIdx.x == 0) *my_note = 0u;                        // this parity is written again in round + 1, after this launch
        // ---- stage brick + halo as ONE batch of independent loads: thread tid owns the interior row (x, y) = (tid / 16, tid % 16) -- 16
        //      voxels = 4 + 1 wide loads -- and <= 7 of the 1736 halo entries (loaded from a valid address unconditionally so that
        //      nothing separates the requests) ----
        {
            const size_t v0 = (size_t)p * TSL_BRK3 + (size_t)threadIdx.x * 16;
            uint4 dq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) dq[q] = reinterpret_cast<const uint4*>(E.mag + v0)[q];
            const uint4 fq = *reinterpret_cast<const uint4*>(E.fl + v0);
            int ht[HPER], hq[HPER]; uint8_t hf[HPER]; uint32_t hd[HPER]; bool hok[HPER];
#pragma unroll
            for (int q = 0; q < HPER; ++q) {
                const int h = q * 256 + (int)threadIdx.x;
                int tx, ty, tz;
                if (h < 2 * ESDF_T * ESDF_T) { const int r = h % (ESDF_T * ESDF_T); tx = (h / (ESDF_T * ESDF_T)) * 17; ty = r / ESDF_T; tz = r % ESDF_T; }       // faces x = 0, 17
                else if (h < 2 * ESDF_T * ESDF_T + 16 * 2 * ESDF_T) { const int g = h - 2 * ESDF_T * ESDF_T, r = g % (2 * ESDF_T); tx = 1 + g / (2 * ESDF_T); ty = (r / ESDF_T) * 17; tz = r % ESDF_T; }   // rows y = 0, 17
                else { const int g = h - (2 * ESDF_T * ESDF_T + 16 * 2 * ESDF_T); tx = 1 + g / 32; ty = 1 + (g % 32) / 2; tz = (g & 1) * 17; }                // entries z = 0, 17
                ht[q] = h < NH ? (tx * ESDF_T + ty) * ESDF_T + tz : -1;
                hq[q] = h < NH ? (((tx + 15) >> 4) * 3 + ((ty + 15) >> 4)) * 3 + ((tz + 15) >> 4) : 13;     // which of the 27 bricks
                const int np = s_nb[hq[q]];
                hok[q] = h < NH && np >= 0;
                const size_t v = (size_t)(np >= 0 ? np : p) * TSL_BRK3 + ((((tx + 15) & 15) << 8) | (((ty + 15) & 15) << 4) | ((tz + 15) & 15));
                hf[q] = E.fl[v]; hd[q] = __float_as_uint(E.mag[v]);
            }
            const uint32_t dl[16] = { dq[0].x, dq[0].y, dq[0].z, dq[0].w, dq[1].x, dq[1].y, dq[1].z, dq[1].w, dq[2].x, dq[2].y, dq[2].z, dq[2].w, dq[3].x, dq[3].y, dq[3].z, dq[3].w };
            const uint32_t fw[4] = { fq.x, fq.y, fq.z, fq.w };
            const int t0 = (((int)(threadIdx.x >> 4) + 1) * ESDF_T + (int)(threadIdx.x & 15) + 1) * ESDF_T + 1;
            uint32_t am = 0u;                                        // active bits of the row
            uint32_t enc[16];
#pragma unroll
            for (int z = 0; z < 16; ++z) {
                const uint32_t f = (fw[z >> 2] >> ((z & 3) * 8)) & 0xffu;
                enc[z] = (f & EF_NODE) ? (dl[z] | ((f & EF_NEG) ? 0x80000000u : 0u)) : 0u;
                s_d[t0 + z] = (f & EF_FIXED) ? 0u : enc[z];
                am |= ((f & (EF_NODE | EF_FIXED)) == (EF_NODE | EF_FIXED) ? 1u : 0u) << z;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) reinterpret_cast<uint4*>(s_old)[threadIdx.x * 4 + q] = make_uint4(enc[4 * q], enc[4 * q + 1], enc[4 * q + 2], enc[4 * q + 3]);
            if (first && am) {                                       // the row's 16 bits lie in one or two words
                atomicOr(&s_a[t0 >> 5], am << (t0 & 31));
                if ((t0 & 31) > 16) atomicOr(&s_a[(t0 >> 5) + 1], am >> (32 - (t0 & 31)));
            }
#pragma unroll
            for (int q = 0; q < HPER; ++q) {
                const int t = ht[q];
                const uint32_t f = hok[q] ? (uint32_t)hf[q] : 0u;
                s_hv[q * 256 + threadIdx.x] = (f & EF_NODE) ? (hd[q] | ((f & EF_NEG) ? 0x80000000u : 0u)) : 0u;
                if (t < 0) continue;
                s_d[t] = 0u;
                // a source can only improve a target whose value exceeds its own by an edge cost, and no value exceeds max_dist
                if ((f & EF_NODE) && ((note >> hq[q]) & 1u) && __uint_as_float(hd[q]) + vs < max_dist) atomicOr(&s_a[t >> 5], 1u << (t & 31));
            }
        }
        __syncthreads();
        ESDF_TICK(1);
        // ---- push relaxation: active entries offer value + edge cost to the targets among their 26 neighbours.
        //      Entry t belongs to thread t mod 256 (a front -- a sheet of neighbouring voxels -- spreads over all threads).  Per pass a
        //      thread reads the active words of its 23 entries as one batch; an active entry is cleared with a non-returning atomic AND,
        //      then its word and the 26 neighbours' words are read as ONE batch of independent LDS loads (relaxed workgroup-scope
        //      atomic loads: plain ds_read, but never cached in registers across passes), and non-returning atomic mins go out for the
        //      candidates that beat what was read (a min that lost a race is a no-op and the extra activation is harmless); the active
        //      bits of the three z-neighbours of a row are set with one OR.  LDS operations of a wave execute in order: the value read
        //      follows the clear, the OR follows the min. ----
        long long pushes = 0; int passes = 0;
#define LDS_LD(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
        for (;;) {
            bool act = false;
            constexpr int PER = (ESDF_T3 + 255) / 256;
            // ESDF_SWEEPS scan + push sweeps per barrier (LDS atomics are visible to the other waves at once, the barrier is only needed to
            // agree that nothing is active any more).  One is the default: with two or four the front loses its order, voxels are
            // pushed 6 - 13 % more often and the update gets slower (0.84 / 1.06 ms against 0.80)
#pragma unroll 1
            for (int sweep = 0; sweep < ESDF_SWEEPS; ++sweep) {
            uint32_t mine = 0u;                                    // bit q: my q-th entry is active
            {
                uint32_t aw[PER];
#pragma unroll
                for (int q = 0; q < PER; ++q) { const int t = q * 256 + (int)threadIdx.x; aw[q] = t < ESDF_T3 ? LDS_LD(&s_a[t >> 5]) : 0u; }
#pragma unroll
                for (int q = 0; q < PER; ++q) { const int t = q * 256 + (int)threadIdx.x; mine |= ((aw[q] >> (t & 31)) & 1u) << q; }
            }
            // The wave's active entries are COMPACTED before they are pushed: every lane hands up to four of its entries to the wave's
            // queue (a prefix sum over the lanes gives the positions), then lane i pushes entries i, i + 64, ...  Without this a wave
            // iterates as often as its busiest lane has entries (~35 % of the lanes busy on average); the push body is ~450
            // instructions, the hand-over ~40.  LDS operations of one wave execute in order: no barrier between the queue's writes and reads.
#pragma unroll 1
            while (__any(mine != 0u)) {
            const int lane = (int)(threadIdx.x & 63u);
            uint16_t* const wq = s_q + (threadIdx.x >> 6) * ESDF_QCAP;
            const int c = min((int)__builtin_popcount(mine), ESDF_QCAP / 64);
            int inc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
            const int total = __shfl(inc, 63);
#pragma unroll
            for (int k = 0; k < ESDF_QCAP / 64; ++k) {
                if (k < c) {
                    const int t = (int)__builtin_ctz(mine) * 256 + (int)threadIdx.x; mine &= mine - 1u;
                    __hip_atomic_fetch_and(&s_a[t >> 5], ~(1u << (t & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(&wq[inc - c + k], (uint16_t)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
            for (int qi = lane; qi < total; qi += 64) {
                const int t = (int)__hip_atomic_load(&wq[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                uint32_t self = LDS_LD(&s_d[t]);
                ++pushes;
                uint32_t dn[26];
#pragma unroll
                for (int c = 0; c < 27; ++c) {
                    if (c == 13) continue;
                    dn[c < 13 ? c : c - 1] = LDS_LD(&s_d[t + ((c / 9 - 1) * ESDF_T + ((c / 3) % 3 - 1)) * ESDF_T + (c % 3 - 1)]);
                }
                if (self == 0u) {                                  // not a target: a halo or band voxel, its value is kept aside (first pass only)
                    const int tz = t % ESDF_T, ty = (t / ESDF_T) % ESDF_T, tx = t / (ESDF_T * ESDF_T);
                    const bool inner = tx >= 1 && tx <= 16 && ty >= 1 && ty <= 16 && tz >= 1 && tz <= 16;
                    self = inner ? s_old[((tx - 1) << 8) | ((ty - 1) << 4) | (tz - 1)] : s_hv[esdf_halo_index(tx, ty, tz)];
                }
                const uint32_t sb = self & 0x80000000u;
                const float dv = __uint_as_float(self & 0x7fffffffu);
                // the three candidates (face, edge, corner neighbour) and whether a voxel that takes one could improve anything itself
                const float cf[3] = { dv + cost[1], dv + cost[2], dv + cost[3] };
                const uint32_t cw[3] = { sb | __float_as_uint(cf[0]), sb | __float_as_uint(cf[1]), sb | __float_as_uint(cf[2]) };
                const uint32_t live[3] = { cf[0] + vs < max_dist ? 1u : 0u, cf[1] + vs < max_dist ? 1u : 0u, cf[2] + vs < max_dist ? 1u : 0u };
                uint32_t any = 0u;
#pragma unroll
                for (int r = 0; r < 9; ++r) {                      // the nine (dx, dy) rows of the neighbourhood, three z-neighbours each
                    const int dx = r / 3 - 1, dy = r % 3 - 1;
                    const int j0 = t + (dx * ESDF_T + dy) * ESDF_T - 1;
                    uint32_t bits = 0u;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int c = r * 3 + k;
                        if (c == 13) continue;
                        const int e = dx * dx + dy * dy + (k - 1) * (k - 1) - 1;
                        // same side, a target, and the candidate is lower <=> one signed comparison.  The minimum stays conditional: the
                        // LDS atomics, not the VALU, bound this loop (26 unconditional ones per push were 30 % slower)
                        if ((int)(dn[c < 13 ? c : c - 1] ^ sb) > (int)__float_as_uint(cf[e])) {
                            __hip_atomic_fetch_min(&s_d[j0 + k], cw[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            bits |= live[e] << k;
                        }
                    }
                    if (bits) {
                        const int sh = j0 & 31;
                        __hip_atomic_fetch_or(&s_a[j0 >> 5], bits << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (sh > 29) __hip_atomic_fetch_or(&s_a[(j0 >> 5) + 1], bits >> (32 - sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    any |= bits;
                }
                act = act || any != 0u;
            }
            }
            if (!__any(act)) break;                                // nothing pushed by this wave in this sweep: wait for the others
            }
            ++passes;
            if (!__syncthreads_or(act)) break;
        }
#undef LDS_LD
#ifdef TSL_TIMING
        if (threadIdx.x == 0) { const long long _n = wall_clock64(); const int pb = passes < 31 ? passes : 31;
            atomicAdd(&E.tm[8 + pb], (unsigned long long)(_n - _t)); atomicAdd(&E.tm[40 + pb], 1ull); atomicMax(&E.tm[72], (unsigned long long)(_n - _t)); }
        { const long long pw = wave_sum_ll(pushes); if (lane_id() == 0) atomicAdd(&E.tm[80 + (passes < 31 ? passes : 31)], (unsigned long long)pw); }
#endif
        ESDF_TICK(2);
        // ---- write back the targets that changed (against the staged copy); a changed voxel of the boundary layer marks the neighbours
        //      that hold it in their halo ----
        {
            float* gm = E.mag + (size_t)p * TSL_BRK3;
#pragma unroll
            for (int q = 0; q < TSL_BRK3 / 256; ++q) {
                const int l = q * 256 + threadIdx.x;
                const int x = (l >> 8) + 1, y = ((l >> 4) & 15) + 1, z = (l & 15) + 1;
                const uint32_t nv = s_d[(x * ESDF_T + y) * ESDF_T + z];
                if (nv != 0u && nv != s_old[l]) {
                    gm[l] = __uint_as_float(nv & 0x7fffffffu);
                    // the neighbours (ax, ay, az) in {0, 1, 2}^3 whose halo holds this voxel: per axis the centre, plus the lower / upper
                    // neighbour when the voxel lies in the first / last layer -- a product of three small bit sets
                    const int sx = x == 1 ? 3 : (x == 16 ? 6 : 2), sy = y == 1 ? 3 : (y == 16 ? 6 : 2), sz = z == 1 ? 3 : (z == 16 ? 6 : 2);
                    if ((sx | sy | sz) != 2) {
                        const int myz = ((sy & 1) ? sz : 0) | ((sy & 2) ? sz << 3 : 0) | ((sy & 4) ? sz << 6 : 0);
                        const int m = ((sx & 1) ? myz : 0) | ((sx & 2) ? myz << 9 : 0) | ((sx & 4) ? myz << 18 : 0);
                        atomicOr(&s_notify, m & ~(1 << 13));
                    }
                }
            }
        }
        __syncthreads();
        ESDF_TICK(3);
        pushes = wave_sum_ll(pushes);
        if (lane_id() == 0 && pushes) __hip_atomic_fetch_add(&E.ctr[6], (int)pushes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the neighbours that see a changed boundary voxel go on the next round's list: one lane per neighbour (the region check, the
        // stamp exchange and the list reservation are dependent device-memory round trips)
        if (threadIdx.x < 64) {
            const int q = (int)threadIdx.x;
            bool put = false; int np = -1;
            if (q < 27 && ((s_notify >> q) & 1)) {
                np = s_nb[q];
                if (np >= 0 && E.region[np] != 0) {                                            // present and inside this update's region
                    // seen from the neighbour this brick is neighbour 26 - q: its halo entries from here push in the next round
                    __hip_atomic_fetch_or(E.note + (size_t)((round + 1) & 1) * E.cap + np, 1u << (26 - q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    put = __hip_atomic_exchange(&E.stamp[np], round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != round + 1;   // not listed yet
                }
            }
            const int at = wave_reserve(&E.ctr[2 + nxt], put);
            if (put) next[at] = np;
        }
        if (threadIdx.x == 0) {
            E.region[p] = 2;
            __hip_atomic_fetch_add(&E.ctr[5], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&E.ctr[8], passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(&E.ctr[9], passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        ESDF_TICK(4);
    }
}

__global__ void __launch_bounds__(256) k_esdf_export(MapDev M, int s, int nused, const float* esdf, float gamma, int16_t* idx, float* out, long long cap, int* counter)
{
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        const int owner = M.owner[p];
        if (owner / M.nb3 != s) continue;
        const int b = owner - s * M.nb3;
        const int bk = b % M.nbz, bj = (b / M.nbz) % M.nbx, bi = b / (M.nbz * M.nbx);
        for (int l0 = 0; l0 < TSL_BRK3; l0 += 256) {
            const int l = l0 + threadIdx.x;
            const size_t v = (size_t)p * TSL_BRK3 + l;
            const bool pred = M.obs[v] > 0;
            const int o = wave_reserve(counter, pred);
            if (pred && o < cap) {
                idx[(size_t)o * 3] = (int16_t)(bi * 16 + (l >> 8) - M.hN); idx[(size_t)o * 3 + 1] = (int16_t)(bj * 16 + ((l >> 4) & 15) - M.hN);
                idx[(size_t)o * 3 + 2] = (int16_t)(bk * 16 + (l & 15) - M.hNz);
                const float t = h2f((h16)(M.tw[v] & 0xffffu));
                out[o] = fabsf(t) < gamma ? t : (float)sgn_f(t) * esdf[v];
            }
        }
    }
}

// cvt_ESDF_to_voxels_slice  dense_esdf.py:498-509: every observed voxel of the active submap with _index - 0.5 < k < _index + 0.5
// (k counted from the bottom of the volume, as the legacy module does) -> export_ESDF / export_ESDF_xyz, num_export_ESDF_particles
struct PoseE { float R[9], T[3]; };
__global__ void __launch_bounds__(256) k_esdf_slice(MapDev M, int s, int nused, const float* esdf, float gamma, PoseE B, int is_global, float vs, float index_f,
                                                    float* xyz, float* val, long long cap, int* counter)
{
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        const int owner = M.owner[p];
        if (owner / M.nb3 != s) continue;
        const int b = owner - s * M.nb3;
        const int bk = b % M.nbz, bj = (b / M.nbz) % M.nbx, bi = b / (M.nbz * M.nbx);
        for (int l0 = 0; l0 < TSL_BRK3; l0 += 256) {
            const int l = l0 + threadIdx.x;
            const size_t v = (size_t)p * TSL_BRK3 + l;
            const int ku = bk * 16 + (l & 15);                                                   // k from the bottom of the volume
            const bool pred = M.obs[v] > 0 && index_f - 0.5f < (float)ku && (float)ku < index_f + 0.5f;      // :504
            const int o = wave_reserve(counter, pred);                                            // :505
            if (pred && o < cap) {
                const float t = h2f((h16)(M.tw[v] & 0xffffu));
                val[o] = fabsf(t) < gamma ? t : (float)sgn_f(t) * esdf[v];                        // :507
                const int i = bi * 16 + (l >> 8) - M.hN, j = bj * 16 + ((l >> 4) & 15) - M.hN, k = ku - M.hNz;
                const float p0 = (float)i * vs, p1 = (float)j * vs, p2 = (float)k * vs;           // :508  submap_i_j_k_to_xyz (mapping_common.py:221-232)
                if (is_global) { xyz[(size_t)o * 3] = p0; xyz[(size_t)o * 3 + 1] = p1; xyz[(size_t)o * 3 + 2] = p2; }
                else for (int a = 0; a < 3; ++a) xyz[(size_t)o * 3 + a] = ((B.R[a * 3] * p0 + B.R[a * 3 + 1] * p1) + B.R[a * 3 + 2] * p2) + B.T[a];
            }
        }
    }
}

// ---- host side.  An update is a fixed sequence of launches on the handle's stream (collect, dilate, init, a batch of rounds sized by
// max_dist -- a round without work returns at once) followed by a copy of the counters into a pinned slot and an event.  Whether the
// last launched round still had work (never seen with this batch size) is all the host needs to know, and it does not need to know
// it now: tsl_esdf_update with n_relaxed == NULL returns after the enqueue and the slot is looked at by a later call.  Everything that
// hands ESDF values or statistics out (update with n_relaxed, last_stats, totals, export) goes through esdf_finish() first, which
// waits for the outstanding updates and, should one have stopped early, recomputes -- so the values a caller sees are always the
// fixed point of the current TSDF. ----
static void esdf_retire(tsl_tsdf* m, bool wait_all)
{
    while (m->esdf_npend > 0) {
        EsdfSlot& S = m->esdf_slot[m->esdf_tail];
        if (!wait_all && hipEventQuery(S.ev) != hipSuccess) break;
        if (wait_all) (void)hipEventSynchronize(S.ev);
        const int* h = S.host;
        tsl_esdf_stats st = S.st;
        st.dirty_bricks = h[0]; st.changed_bricks = h[11]; st.region_bricks = h[1]; st.brick_relaxations = h[5]; st.voxel_pushes = h[6];
        st.rounds = h[7]; st.passes = h[8]; st.max_passes = h[9]; st.total_bricks = h[10] < m->M.max_bricks ? h[10] : m->M.max_bricks;
        if (h[2 + S.rounds % 3] != 0) m->esdf_short = true;                  // the last launched round still had work
        if (st.incremental && st.rounds > m->esdf_rounds_seen) m->esdf_rounds_seen = st.rounds;
        m->esdf_stats = st;
        m->esdf_tot.updates += 1; m->esdf_tot.incremental += st.incremental; m->esdf_tot.dirty_bricks += st.dirty_bricks; m->esdf_tot.region_bricks += st.region_bricks;
        m->esdf_tot.brick_relaxations += st.brick_relaxations; m->esdf_tot.voxel_pushes += st.voxel_pushes; m->esdf_tot.passes += st.passes;
#ifdef TSL_TIMING
        { const unsigned long long* tm = (const unsigned long long*)&h[16]; const double n = st.brick_relaxations ? st.brick_relaxations : 1;
          std::fprintf(stderr, "esdf timing: us per relaxation: setup %.2f stage %.2f relax %.2f writeback %.2f notify %.2f (%lld relaxations)\n",
                       tm[0] / n / 100.0, tm[1] / n / 100.0, tm[2] / n / 100.0, tm[3] / n / 100.0, tm[4] / n / 100.0, (long long)st.brick_relaxations);
          std::fprintf(stderr, "esdf relax by passes (count: mean us, mean pushes); max relax %.1f us\n", tm[72] / 100.0);
          for (int k = 0; k < 32; ++k) if (tm[40 + k]) std::fprintf(stderr, "  %2d passes: %5llu relaxations, %7.1f us, %7.0f pushes\n", k, tm[40 + k], tm[8 + k] / (double)tm[40 + k] / 100.0, tm[80 + k] / (double)tm[40 + k]); }
#endif
        m->esdf_tail = (m->esdf_tail + 1) % TSL_ESDF_SLOTS; --m->esdf_npend;
    }
}

static int esdf_enqueue(tsl_tsdf* m, float gamma, float max_dist, bool force_full, int extra_rounds)
{
    int rc;
    const int nb = m->M.max_bricks;
    if (!m->esdf) {
        if ((rc = dev_alloc(m, (void**)&m->esdf, sizeof(float) * (size_t)nb * TSL_BRK3, 0))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->esdf_fl, (size_t)nb * TSL_BRK3, 0))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->esdf_region, (size_t)nb, 0))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->esdf_inq, sizeof(int) * (size_t)nb, 0xff))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->esdf_list, sizeof(int) * (size_t)nb, 0))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->esdf_note, sizeof(uint32_t) * 2 * (size_t)nb, 0))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->esdf_queue, sizeof(int) * 3 * (size_t)nb, 0))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->esdf_ctr, sizeof(int) * 256, 0))) return rc;
        for (int i = 0; i < TSL_ESDF_SLOTS; ++i) {
            TSL_HIP(hipEventCreateWithFlags(&m->esdf_slot[i].ev, hipEventDisableTiming));
            TSL_HIP(hipHostMalloc((void**)&m->esdf_slot[i].host, sizeof(int) * 256, hipHostMallocDefault));
        }
        TSL_HIP(hipEventCreateWithFlags(&m->esdf_gate, hipEventDisableTiming));
        m->esdf_valid = false;
    }
    esdf_retire(m, false);
    if (m->esdf_npend == TSL_ESDF_SLOTS) { EsdfSlot& S = m->esdf_slot[m->esdf_tail]; (void)hipEventSynchronize(S.ev); esdf_retire(m, false); }
    hipStream_t q = ms(m);                                   // issues the queued frames first
    const int s = m->cfg.is_global_map ? 0 : m->active;
    // a changed voxel influences voxels up to max_dist away: that many voxels = `reach` bricks in every direction
    int reach = (int)std::ceil((double)max_dist / ((double)m->P.vs * 16.0)); if (reach < 1) reach = 1;
    const bool full = force_full || m->esdf_force_full || !m->esdf_valid || m->esdf_submap != s || m->esdf_gamma != gamma || m->esdf_maxd != max_dist ||
                      2 * reach + 1 >= m->nbx;              // the dilation would cover the grid anyway
    EsdfDev E = { m->esdf, m->esdf_fl, m->esdf_region, m->esdf_inq, m->esdf_list, m->esdf_note, m->esdf_queue, nb, (unsigned long long*)(m->esdf_ctr + 16), m->esdf_ctr };
    EsdfSlot& S = m->esdf_slot[(m->esdf_tail + m->esdf_npend) % TSL_ESDF_SLOTS];
    std::memset(&S.st, 0, sizeof(S.st));
    S.st.incremental = full ? 0 : 1;
    TSL_HIP(hipMemsetAsync(m->esdf_ctr, 0, sizeof(int) * 256, q));
    TSL_HIP(hipMemcpyAsync(m->esdf_ctr + 10, m->M.pool_top, sizeof(int), hipMemcpyDeviceToDevice, q));      // the update's brick-count snapshot
    prof_begin(m, TSL_K_ESDF);                                   // one event pair around the update's launches (collect .. last round)
    m->prof_group = true;
    const int nbk = (nb + 255) / 256;
    hipLaunchKernelGGL(k_esdf_collect, dim3(nbk), dim3(256), 0, q, m->M, E, s, full ? 1 : 0);
    // phase A of frames queued from now on allocates bricks (pool counter, table entry, owner -- in that order): it starts after the
    // snapshot + collect, so that every pool index below the snapshot has its owner written (launch_batch_t waits for the gate)
    TSL_HIP(hipEventRecord(m->esdf_gate, q)); m->esdf_gate_set = true;
    hipLaunchKernelGGL(k_esdf_dilate, dim3(1024), dim3(256), 0, q, m->M, E, s, full ? 0 : reach, full ? 1 : 0, gamma, max_dist);
    hipLaunchKernelGGL(k_esdf_init, dim3(2048), dim3(256), 0, q, m->M, E, gamma, max_dist);
    // information crosses one brick per round: `reach` rounds carry a value as far as it can matter, bends and late improvements add a
    // few more (8 rounds had work at reach = 4 on the benchmark stream).
    // The batch is launched blind: 2 * reach + 8 rounds to begin with and for full recomputes, afterwards three more than the most any
    // completed update of this handle needed (a round without work costs ~5 us; stopping early costs a full recompute, see esdf_finish).
    int rounds = ((full || m->esdf_rounds_seen == 0) ? 2 * reach + 8 : std::min(2 * reach + 8, std::max(reach + 3, m->esdf_rounds_seen + 3))) + extra_rounds;
    const int grid = 4 * m->ncu;
    if (m->esdf_round_cap > 0 && extra_rounds == 0 && rounds > m->esdf_round_cap) rounds = m->esdf_round_cap;      // test knob: provoke the repair path
    for (int k = 0; k < rounds; ++k) hipLaunchKernelGGL(k_esdf_round, dim3(grid), dim3(256), 0, q, m->M, E, s, m->P.vs, max_dist, k);
    m->prof_group = false; prof_end(m);
    TSL_HIP(hipMemcpyAsync(S.host, m->esdf_ctr, sizeof(int) * 256, hipMemcpyDeviceToHost, q));
    TSL_HIP(hipEventRecord(S.ev, q));
    TSL_HIP(hipGetLastError());
    S.rounds = rounds;
    ++m->esdf_npend;
    m->esdf_gamma = gamma; m->esdf_maxd = max_dist; m->esdf_submap = s; m->esdf_valid = true;
    return TSL_OK;
}

// wait for the outstanding updates; if one of them stopped with work left, recompute everything with ever longer batches
int esdf_finish(tsl_tsdf* m)
{
    if (!m->esdf) return TSL_OK;
    TSL_HIP(hipSetDevice(m->device));
    esdf_retire(m, true);
    for (int extra = 16; m->esdf_short; extra *= 2) {
        m->esdf_short = false;
        const int rc = esdf_enqueue(m, m->esdf_gamma, m->esdf_maxd, true, extra); if (rc) return rc;
        esdf_retire(m, true);
        TSL_REQUIRE(extra < (1 << 16), "esdf: the relaxation does not terminate");
    }
    return TSL_OK;
}

void esdf_release(tsl_tsdf* m)
{
    esdf_retire(m, true);
    for (int i = 0; i < TSL_ESDF_SLOTS; ++i) {
        if (m->esdf_slot[i].ev) (void)hipEventDestroy(m->esdf_slot[i].ev);
        if (m->esdf_slot[i].host) (void)hipHostFree(m->esdf_slot[i].host);
        m->esdf_slot[i].ev = nullptr; m->esdf_slot[i].host = nullptr;
    }
    if (m->esdf_gate) { (void)hipEventDestroy(m->esdf_gate); m->esdf_gate = nullptr; }
}

}  // namespace tsl

using namespace tsl;

extern "C" {

int tsl_esdf_update(tsl_tsdf* m, float gamma, float max_dist, int32_t* n_relaxed)
{
    TSL_REQUIRE(m, "esdf_update: null handle"); TSL_REQUIRE(gamma > 0 && max_dist > 0, "esdf_update: gamma and max_dist must be positive");
    TSL_HIP(hipSetDevice(m->device));
    int rc;
    if (m->esdf_short && (rc = esdf_finish(m))) return rc;       // an earlier update stopped early: repair before building on it
    if ((rc = esdf_enqueue(m, gamma, max_dist, false, 0))) return rc;
    if (n_relaxed) { if ((rc = esdf_finish(m))) return rc; *n_relaxed = (int32_t)m->esdf_stats.brick_relaxations; }
    return TSL_OK;
}

int tsl_esdf_last_stats(tsl_tsdf* m, tsl_esdf_stats* out)
{
    TSL_REQUIRE(m && out, "null");
    const int rc = esdf_finish(m); if (rc) return rc;
    *out = m->esdf_stats; return TSL_OK;
}

int tsl_esdf_totals(tsl_tsdf* m, tsl_esdf_totals_t* out)
{
    TSL_REQUIRE(m && out, "null");
    const int rc = esdf_finish(m); if (rc) return rc;
    *out = m->esdf_tot; return TSL_OK;
}

// compaction of the observed voxels into the handle's staging buffer: int16 idx[cap][3] | f32 esdf[cap]; *c = true count
static int esdf_export_stage(tsl_tsdf* m, int64_t cap, int16_t** didx, float** dval, int* c)
{
    TSL_REQUIRE(m->esdf, "esdf_export: call tsl_esdf_update first");
    TSL_HIP(hipSetDevice(m->device));
    int rc = esdf_finish(m); if (rc) return rc;
    int nused = 0; rc = tsl_tsdf_bricks_in_use(m, &nused); if (rc) return rc;
    const size_t need = (((size_t)cap * 6 + 15) / 16) * 16 + (size_t)cap * 4 + 64;
    rc = grow(&m->xbuf, &m->xbuf_bytes, need); if (rc) return rc;
    *didx = (int16_t*)m->xbuf; *dval = (float*)((char*)m->xbuf + (((size_t)cap * 6 + 15) / 16) * 16);
    int* counter = m->num_particles + 2;
    TSL_HIP(hipMemsetAsync(counter, 0, sizeof(int), ms(m)));
    const int s = m->cfg.is_global_map ? 0 : m->active;
    if (nused > 0) hipLaunchKernelGGL(k_esdf_export, dim3(nused < 8192 ? nused : 8192), dim3(256), 0, ms(m), m->M, s, nused, m->esdf, m->esdf_gamma, *didx, *dval, (long long)cap, counter);
    TSL_HIP(hipMemcpyAsync(m->h_ints, counter, sizeof(int), hipMemcpyDeviceToHost, ms(m)));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    *c = m->h_ints[0];
    return TSL_OK;
}

int tsl_esdf_export(tsl_tsdf* m, int16_t* idx, float* esdf, int64_t cap, int64_t* n)
{
    TSL_REQUIRE(m && n && cap >= 0, "esdf_export: bad argument");
    int16_t* didx; float* dval; int c = 0;
    int rc = esdf_export_stage(m, cap, &didx, &dval, &c); if (rc) return rc;
    *n = c;
    const size_t k = (size_t)(c < cap ? c : cap);
    if (k && idx) TSL_HIP(hipMemcpy(idx, didx, k * 6, hipMemcpyDeviceToHost));
    if (k && esdf) TSL_HIP(hipMemcpy(esdf, dval, k * 4, hipMemcpyDeviceToHost));
    return TSL_OK;
}

/* the same compaction, left on the device: *idx_dev = int16 [min(n, cap)][3], *val_dev = f32 [min(n, cap)] inside the handle's staging
 * buffer (valid until the next call on this handle that exports, imports or queries through host buffers) */
int tsl_esdf_export_dev(tsl_tsdf* m, int64_t cap, void** idx_dev, void** val_dev, int64_t* n)
{
    TSL_REQUIRE(m && n && idx_dev && val_dev && cap >= 0, "esdf_export_dev: bad argument");
    int16_t* didx; float* dval; int c = 0;
    int rc = esdf_export_stage(m, cap, &didx, &dval, &c); if (rc) return rc;
    *n = c; *idx_dev = didx; *val_dev = dval;
    return TSL_OK;
}

/* cvt_ESDF_to_voxels_slice(z)  dense_esdf.py:498-509: the ESDF of the voxel layer at height z of the active submap -> the handle's
 * export_ESDF_xyz / export_ESDF buffers (max_disp_particles rows, device-resident), *n = num_export_ESDF_particles (true count) */
int tsl_esdf_slice(tsl_tsdf* m, float z, int32_t* n)
{
    TSL_REQUIRE(m && n, "esdf_slice: bad argument"); TSL_REQUIRE(m->esdf, "esdf_slice: call tsl_esdf_update first");
    TSL_HIP(hipSetDevice(m->device));
    int rc = esdf_finish(m); if (rc) return rc;
    if (!m->esdf_exp_xyz) {
        if ((rc = dev_alloc(m, (void**)&m->esdf_exp_xyz, sizeof(float) * 3 * (size_t)m->max_disp, 0))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->esdf_exp_val, sizeof(float) * (size_t)m->max_disp, 0))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->esdf_exp_count, sizeof(int) * 4, 0))) return rc;
    }
    int nused = 0; rc = tsl_tsdf_bricks_in_use(m, &nused); if (rc) return rc;
    TSL_HIP(hipMemsetAsync(m->esdf_exp_count, 0, sizeof(int), ms(m)));                    // :500
    const int s = m->cfg.is_global_map ? 0 : m->active;
    PoseE B;
    for (int a = 0; a < 9; ++a) B.R[a] = m->baseRf[(size_t)m->active * 9 + a];
    for (int a = 0; a < 3; ++a) B.T[a] = m->baseTf[(size_t)m->active * 3 + a];
    // _index = (z + map_size_[2] / 2) / voxel_scale: Python floats at trace time (z is a ti.template()), an f32 constant in the kernel (:503)
    const float index_f = (float)(((double)z + (double)m->Nz * m->cfg.voxel_scale / 2.0) / m->cfg.voxel_scale);
    if (nused > 0) hipLaunchKernelGGL(k_esdf_slice, dim3(nused < 8192 ? nused : 8192), dim3(256), 0, ms(m), m->M, s, nused, m->esdf, m->esdf_gamma, B, m->cfg.is_global_map, m->P.vs,
                                      index_f, m->esdf_exp_xyz, m->esdf_exp_val, (long long)m->max_disp, m->esdf_exp_count);
    TSL_HIP(hipMemcpyAsync(m->h_ints, m->esdf_exp_count, sizeof(int), hipMemcpyDeviceToHost, ms(m)));
    TSL_HIP(hipStreamSynchronize(ms(m)));
    m->esdf_exp_n = m->h_ints[0];
    *n = m->esdf_exp_n;
    return TSL_OK;
}
/* rows [0, n) of export_ESDF_xyz / export_ESDF (either may be NULL) */
int tsl_esdf_read_slice(tsl_tsdf* m, float* xyz, float* val, int64_t n)
{
    TSL_REQUIRE(m, "null handle"); TSL_REQUIRE(n >= 0 && n <= m->max_disp, "esdf_read_slice: n out of range"); TSL_HIP(hipSetDevice(m->device));
    if (n == 0) return TSL_OK;
    TSL_REQUIRE(m->esdf_exp_xyz, "esdf_read_slice: call tsl_esdf_slice first");
    TSL_HIP(hipStreamSynchronize(ms(m)));
    if (xyz) TSL_HIP(hipMemcpy(xyz, m->esdf_exp_xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
    if (val) TSL_HIP(hipMemcpy(val, m->esdf_exp_val, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    return TSL_OK;
}
/* the same buffers as device pointers (valid for the lifetime of the handle) + the count of the last tsl_esdf_slice */
int tsl_esdf_slice_dev(tsl_tsdf* m, void** xyz_dev, void** val_dev, int32_t* n)
{
    TSL_REQUIRE(m && m->esdf_exp_xyz, "esdf_slice_dev: call tsl_esdf_slice first");
    if (xyz_dev) *xyz_dev = m->esdf_exp_xyz; if (val_dev) *val_dev = m->esdf_exp_val; if (n) *n = m->esdf_exp_n;
    return TSL_OK;
}

}  // extern "C"
