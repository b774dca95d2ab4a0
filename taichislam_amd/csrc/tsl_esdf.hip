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
//   4. relaxes R in rounds.  A round stages every brick on its work list in LDS with its one-voxel halo (18^3 words) and relaxes it to its
//      local fixed point with six concurrent directional sweeps, one wave per direction (k_esdf_round below).  Round 0 lists the bricks that
//      hold a band voxel (or touch a brick outside R); afterwards a brick lists the neighbours in whose voxels -- its own halo -- it could
//      lower something (device-side, deduplicated), so information crosses one brick per round and a brick that only received values
//      does not call the giver back.  Rounds are plain launches; the host launches a batch sized by max_dist and by what earlier updates
//      needed, never waits, and reads the counters when somebody asks for ESDF values.
//      (Tried before: a single persistent kernel with an asynchronous brick queue -- exact, but with no ordering between bricks each was
//      relaxed ~23 times, 5.2 ms per update at 512^3; a push relaxation driven by per-voxel active bits -- one LDS pass per voxel of
//      distance, 17 passes and 70 us to cross a brick, 0.80 ms per update.  Sweeps: 0.33 ms.)
// The first update, an update after reset() / import / fusion, or one with different parameters is the same procedure with R = all bricks.
#include <hip/hip_ext.h>
#include "tsl_tsdf.hpp"

namespace tsl {

#define ESDF_T 18
#define ESDF_T3 (ESDF_T * ESDF_T * ESDF_T)
#define EF_NODE 1
#define EF_NEG 2
#define EF_FIXED 4
#define ES_CTR 256               // ints of counters, followed by
#define ES_STAT_SLOTS 64         // slots of 16 ints (one cache line each) of statistics, summed by the host: hundreds of workgroups adding to ONE
                                 // line serialise in the L2 (~12 ns each) and hold back the list reservations that share it
#define ES_STAT(E, k) (&(E).ctr[ES_CTR + (blockIdx.x & (ES_STAT_SLOTS - 1)) * 16 + (k)])      // [0] relaxations [1] lowered [2] sets [3] max sets [4] region [5] changed [6] raise sets [7] voxels re-derived by the raise sweeps [8] most raise sets of one visit
#define ES_UNOBS 0x7fffffffu
#define ES_INF 0x7f800000u

#ifdef TSL_TIMING
// developer timing: thread 0 of every relaxation adds the clock ticks (100 MHz) of its phases to E.ctr64[k]
#define ESDF_TICK(k) do { if (threadIdx.x == 0) { const long long _n = wall_clock64(); atomicAdd(&E.tm[k], (unsigned long long)(_n - _t)); _t = _n; } } while (0)
#define ESDF_TICKF(k) do { if (threadIdx.x == 0) { const long long _n = wall_clock64(); atomicAdd(&E.tm[k], (unsigned long long)(_n - _t)); if (first && (k) >= 2) { atomicAdd(&E.tm[72 + (k)], (unsigned long long)(_n - _t)); if ((k) == 4) atomicAdd(&E.tm[73], 1ull); } _t = _n; } } while (0)
#else
#define ESDF_TICK(k) do {} while (0)
#define ESDF_TICKF(k) do {} while (0)
#endif
struct EsdfDev {
    float* mag;                // [max_bricks][4096] the signed distance: side << 31 | magnitude bits (a negative float on the negative side), ES_UNOBS (a NaN) where
                               // the voxel was not observed at the last (re)initialisation -- the word the relaxation's tile holds
    uint8_t* fl;               // [max_bricks][4096] EF_* of the voxel at the last (re)initialisation
    uint8_t* region;           // [max_bricks] 1: brick is part of this update's region, 2: relaxed once already
    int* stamp;                // [max_bricks] last round the brick was put on a work list for (dedupe)
    int* dirty;                // [max_bricks] dirty list
    uint32_t* note;            // [2][max_bricks] per round parity: bit q = neighbour q (of 27) changed its boundary layer in the previous round
    int* work;                 // [3][max_bricks] work lists of rounds k, k+1, k+2 (mod 3)
    int* nbr;                  // [max_bricks][27] pool indices of the bricks around a region brick (-1: absent), written by k_esdf_init (esdf_mode 0)
    uint8_t* par;              // [max_bricks][4096] esdf_mode 1: direction code of the voxel's PARENT -- the neighbour its value was taken from --
                               // (dx + 1) << 4 | (dy + 1) << 2 | (dz + 1), 0x15 = none (band voxel, max_dist, unobserved)          dense_esdf.py:96, :290, :296
    uint8_t* ok;               // [max_bricks] esdf_mode 1: the brick's mag / fl / par describe the current (submap, gamma, max_dist)
    int cap;                   // max_bricks
    unsigned long long* tm;    // developer timing (TSL_TIMING builds): ticks per phase, summed over relaxations
    int* ctr_next;             // the counter block of the NEXT update (the two alternate): zeroed by this update's collect kernel, so that an update
                               // needs neither a memset nor a copy of the brick count in front of it (two stream operations, ~13 us per update)
    int* ctr;                  // [0] dirty count [2..4] work list lengths [7] rounds with work [10] brick-count snapshot [240..] bricks per round; statistics: ES_STAT
};

// 1. dirty bricks of submap s (and, when `all`, every brick of it); touch marks are consumed
__global__ void __launch_bounds__(256) k_esdf_collect(MapDev M, EsdfDev E, int s, int all, int wavefront)
{
    // the brick count of this update: a snapshot of the pool counter, taken here (every thread reads the same value: the frames before
    // the update have finished, phase A of frames queued AFTER it starts once this kernel has finished, esdf_gate) and left in ctr[10] for
    // the kernels that follow.  Those frames may allocate bricks while the update runs: every kernel of the update ignores pool indices
    // >= the snapshot, so a brick that appears meanwhile -- owner / flags not written yet -- is simply not there for this update.
    const int snap = M.pool_top[0];
    const int nused = min(snap, M.max_bricks);
    if (blockIdx.x == 0 && threadIdx.x == 0) E.ctr[10] = snap;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ES_CTR + ES_STAT_SLOTS * 16; i += gridDim.x * 256) E.ctr_next[i] = 0;
    const int p = blockIdx.x * 256 + threadIdx.x;
    bool take = false;
    if (all && E.ok && p < E.cap) E.ok[p] = 0;               // (esdf_mode 1) a full recompute forgets every brick's state, also of bricks the pool has not handed out yet
    if (p < nused) {
        const bool mine = M.owner[p] / M.nb3 == s;
        if (mine && wavefront) {                                // (esdf_mode 1) the 27 bricks around this one, for every brick the wave may reach: the pool may have grown since the last update
            const int b = M.owner[p] - s * M.nb3, bi = b / (M.nbz * M.nbx), bj = (b / M.nbz) % M.nbx, bk = b % M.nbz;
            int nb27[27];
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const int i = bi + t / 9 - 1, j = bj + (t / 3) % 3 - 1, k = bk + t % 3 - 1;
                nb27[t] = (i < 0 || i >= M.nbx || j < 0 || j >= M.nbx || k < 0 || k >= M.nbz) ? -1 : pool_lookup_ro(M, s, (i * M.nbx + j) * M.nbz + k);
            }
#pragma unroll
            for (int t = 0; t < 27; ++t) E.nbr[(size_t)p * 27 + t] = nb27[t] >= nused ? -1 : nb27[t];
        }
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
                changed = changed || f != fs || ((f & EF_FIXED) && __float_as_uint(mg) != (mv[z] & 0x7fffffffu));
            }
            if (!__syncthreads_or(changed)) continue;                 // (uniform: every thread of the workgroup leaves or stays)
        }
        if (threadIdx.x == 0) __hip_atomic_fetch_add(ES_STAT(E, 5), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// 3. (re)initialise the region's voxels.  Round 0 relaxes the bricks that hold a source of their own -- a band voxel -- or that touch a
//    brick outside the region (whose values stand as boundary conditions); every other brick of the region waits until a neighbour
//    reports values that can lower something in it (it would otherwise be relaxed from partial information and again a round later).
__global__ void __launch_bounds__(256) k_esdf_init(MapDev M, EsdfDev E, int s, float gamma, float max_dist)
{
    const int nused = min(E.ctr[10], M.max_bricks);
    for (int p = blockIdx.x; p < nused; p += gridDim.x) {
        if (E.region[p] != 1) continue;
        bool seed = false;
        if (threadIdx.x >= 192 && threadIdx.x < 192 + 27) {      // the 27 bricks around this one (a brick allocated after the update's snapshot is not part of it)
            const int t = (int)threadIdx.x - 192;
            const int b = M.owner[p] - s * M.nb3;
            const int i = b / (M.nbz * M.nbx) + t / 9 - 1, j = (b / M.nbz) % M.nbx + (t / 3) % 3 - 1, k = b % M.nbz + t % 3 - 1;
            int np = (i < 0 || i >= M.nbx || j < 0 || j >= M.nbx || k < 0 || k >= M.nbz) ? -1 : pool_lookup_ro(M, s, (i * M.nbx + j) * M.nbz + k);
            if (np >= nused) np = -1;
            E.nbr[(size_t)p * 27 + t] = np;
            seed = np >= 0 && E.region[np] == 0;                    // (k_esdf_dilate, the launch before this one, wrote the region marks)
        }
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
                seed = seed || (f & EF_FIXED);
                fo[z >> 2] |= f << ((z & 3) * 8); mo[z] = __uint_as_float((f & EF_NODE) ? (__float_as_uint(mg) | ((f & EF_NEG) ? 0x80000000u : 0u)) : ES_UNOBS);
            }
            *reinterpret_cast<uint4*>(E.fl + v) = make_uint4(fo[0], fo[1], fo[2], fo[3]);
#pragma unroll
            for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(E.mag + v)[q] = make_float4(mo[4 * q], mo[4 * q + 1], mo[4 * q + 2], mo[4 * q + 3]);
        }
        const bool listed = __syncthreads_or(seed) != 0;
        if (threadIdx.x == 0) {
            if (listed) { const int q = __hip_atomic_fetch_add(&E.ctr[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); E.work[q] = p; E.stamp[p] = 0; }
            __hip_atomic_fetch_add(ES_STAT(E, 4), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// 4. one relaxation round: every brick on this round's work list is staged in LDS with its one-voxel halo and relaxed to its local fixed
//    point; a brick whose boundary layer improved puts the neighbours that see it on the next round's list.  Rounds are separate launches
//    (the kernel boundary is the only synchronisation: no fences, no spinning); a launch whose list is empty returns at once.
//
//    The local relaxation is a set of six DIRECTIONAL SWEEPS, one per wave (+x, -x, +y, -y, +z, -z), all running at the same time on the
//    one tile.  A sweep walks the 16 planes of the brick along its axis; in a plane every voxel PULLS from its nine neighbours in the plane
//    before it (one face, four edge, four corner neighbours: min over a class, then ONE add of the class' edge cost -- fl(a + c) is
//    monotone in a) and lowers itself with an LDS atomic min.  A wave holds a whole plane (64 lanes x 4 voxels in a row), so a sweep needs
//    no barrier: the LDS operations of one wave execute in order.  A shortest path through free space only uses steps that advance along
//    its dominant axis, so ONE sweep carries a value across the whole brick exactly (the push relaxation this replaces needed one pass --
//    scan, compaction, 26 atomics per voxel, a barrier -- per voxel of distance: 17 passes, 70 us, to cross a brick); paths that bend
//    around the other side of the surface or around unobserved space take another set.  The six waves race on the tile; every write is a
//    monotone atomic min of a realisable path cost, so a stale read only delays an improvement, and the relaxation ends with a set in
//    which NO wave lowered anything: in such a set every read saw the final state and the six planes-before cover all 26 neighbours,
//    i.e. the tile is at the fixed point.
//    A brick that is visited again (a neighbour changed its boundary layer) starts with an entry check: only the sweeps that enter through
//    a face with a notified neighbour run, and one that lowers nothing in its first plane stops there (the rest of the tile was at its
//    fixed point already).
//
//    Tile word: side << 31 | magnitude bits of an observed voxel, ES_UNOBS otherwise.  For a voxel on the positive side the unsigned
//    minimum over raw words is the least magnitude among its positive neighbours (negative-side words and ES_UNOBS are larger than any
//    magnitude; clamped to +inf they never win); for the negative side the signed minimum does the same.  Only interior, observed,
//    non-band voxels ("targets", a bit mask per lane) are ever written.
#ifndef ESDF_WPE
#define ESDF_WPE 4                        // waves per SIMD the register budget is set for: two workgroups per CU (three, with 96 registers and
                                          // spills, or four were 7 - 25 % slower: the sweeps saturate the LDS and the VALU of a CU with two)
#endif
#define ES_SY 19                          // row pitch (18 entries + 1: spreads the rows of a plane over the LDS banks)
#define ES_SX (18 * ES_SY + 1)            // plane pitch
#define ES_TILE (18 * ES_SX)
#define ES_LD(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

// the four mask bits of a lane's voxels in plane s of a sweep along AXIS (lane = r * 4 + q: row r, voxels 4q .. 4q + 3 of the row)
template <int AXIS> __device__ __forceinline__ uint32_t esdf_nibble(const uint32_t* bits, int s, int r, int q)
{
    if (AXIS == 0) return (ES_LD(&bits[s * 8 + (r >> 1)]) >> ((r & 1) * 16 + 4 * q)) & 15u;          // voxel (s, r, 4q + j)
    if (AXIS == 1) return (ES_LD(&bits[r * 8 + (s >> 1)]) >> ((s & 1) * 16 + 4 * q)) & 15u;          // voxel (r, s, 4q + j)
    const uint32_t a = ES_LD(&bits[r * 8 + 2 * q]) >> s, b = ES_LD(&bits[r * 8 + 2 * q + 1]) >> s;   // voxel (r, 4q + j, s)
    return (a & 1u) | ((a >> 15) & 2u) | ((b & 1u) << 2) | ((b >> 13) & 8u);
}

// the target / negative-target bits of a lane's 64 voxels in sweep order (bit 4 i + j: voxel j of the i-th plane)
template <int AXIS, int SIGN>
__device__ __forceinline__ void esdf_masks(const uint32_t* s_tgt, const uint32_t* s_neg, unsigned long long& mT, unsigned long long& mN)
{
    const int lane = (int)(threadIdx.x & 63u), q = lane & 3, r = lane >> 2;
    mT = 0ull; mN = 0ull;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int sp = SIGN > 0 ? i : 15 - i;
        const uint32_t t = esdf_nibble<AXIS>(s_tgt, sp, r, q);
        mT |= (unsigned long long)t << (4 * i); mN |= (unsigned long long)(t & esdf_nibble<AXIS>(s_neg, sp, r, q)) << (4 * i);
    }
}

// one sweep of the calling wave.  Returns whether this lane lowered a voxel.
template <int AXIS, int SIGN>
__device__ __forceinline__ bool esdf_sweep(uint32_t* s_t, uint32_t* s_chg, unsigned long long mT, unsigned long long mN, const float c1, const float c2, const float c3,
                                           const bool entry_check, int& lowered)
{
    constexpr int SD = AXIS == 0 ? ES_SX : (AXIS == 1 ? ES_SY : 1);        // along the sweep
    constexpr int SR = AXIS == 0 ? ES_SY : ES_SX;                          // between the three rows a lane reads
    constexpr int SJ = AXIS == 2 ? ES_SY : 1;                              // along a lane's four voxels
    constexpr int PO = SIGN > 0 ? 0 : SD, OO = SIGN > 0 ? SD : 0;          // the plane before / the own plane, from the lower of the two
    const int lane = (int)(threadIdx.x & 63u), q = lane & 3, r = lane >> 2;
    bool changed = false;
    // the lower of (own plane, plane before) in tile coordinates: i for a forward sweep, 16 - i for a backward one
    uint32_t* b = s_t + r * SR + 4 * q * SJ + (SIGN > 0 ? 0 : 16 * SD);
#pragma unroll 1
    for (int i = 0; i < 16; ++i, b += SIGN * SD, mT >>= 4, mN >>= 4) {
        const uint32_t tn = (uint32_t)mT & 15u, nn = (uint32_t)mN & 15u;
        const bool hasP = __any((tn & ~nn) != 0u), hasN = __any(nn != 0u);
        if (!hasP && !hasN) { if (entry_check && i == 0) return false; continue; }
        uint32_t w[3][6], own[4];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int k = 0; k < 6; ++k) w[rr][k] = ES_LD(b + PO + rr * SR + k * SJ);
#pragma unroll
        for (int j = 0; j < 4; ++j) own[j] = ES_LD(b + OO + SR + (j + 1) * SJ);
        uint32_t cb[4];                                            // bits of the candidate of voxel j (+inf: none)
        if (hasP && hasN) {
            // both sides in the plane: the words seen from the side of each voxel (the other side and unobserved turn >= +inf)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t sj = ((nn >> j) & 1u) << 31;
                const uint32_t f = min(w[1][j + 1] ^ sj, ES_INF);
                const uint32_t e = min(min(min(w[1][j] ^ sj, w[1][j + 2] ^ sj), min(w[0][j + 1] ^ sj, w[2][j + 1] ^ sj)), ES_INF);
                const uint32_t c = min(min(min(w[0][j] ^ sj, w[0][j + 2] ^ sj), min(w[2][j] ^ sj, w[2][j + 2] ^ sj)), ES_INF);
                cb[j] = __float_as_uint(fminf(fminf(__uint_as_float(f) + c1, __uint_as_float(e) + c2), __uint_as_float(c) + c3));
            }
        } else if (hasP) {
            uint32_t pr[3][4];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int j = 0; j < 4; ++j) pr[rr][j] = min(w[rr][j], w[rr][j + 2]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t f = min(w[1][j + 1], ES_INF), e = min(min(pr[1][j], min(w[0][j + 1], w[2][j + 1])), ES_INF), c = min(min(pr[0][j], pr[2][j]), ES_INF);
                cb[j] = __float_as_uint(fminf(fminf(__uint_as_float(f) + c1, __uint_as_float(e) + c2), __uint_as_float(c) + c3));
            }
        } else {
            int pr[3][4];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int j = 0; j < 4; ++j) pr[rr][j] = min((int)w[rr][j], (int)w[rr][j + 2]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t f = min(w[1][j + 1] ^ 0x80000000u, ES_INF);
                const uint32_t e = min((uint32_t)min(pr[1][j], min((int)w[0][j + 1], (int)w[2][j + 1])) ^ 0x80000000u, ES_INF);
                const uint32_t c = min((uint32_t)min(pr[0][j], pr[2][j]) ^ 0x80000000u, ES_INF);
                cb[j] = __float_as_uint(fminf(fminf(__uint_as_float(f) + c1, __uint_as_float(e) + c2), __uint_as_float(c) + c3));
            }
        }
        uint32_t cm = 0u;                                          // bit j: voxel j is a target and its candidate is lower
#pragma unroll
        for (int j = 0; j < 4; ++j) cm |= (cb[j] < (own[j] & 0x7fffffffu) ? 1u : 0u) << j;
        cm &= tn;
        const bool ch = cm != 0u;
        if (__any(ch)) {                                           // (rare once the tile has settled: a whole set without it ends the relaxation)
#pragma unroll
            for (int j = 0; j < 4; ++j)                            // a minimum with ~0 changes nothing
                __hip_atomic_fetch_min(b + OO + SR + (j + 1) * SJ, ((cm >> j) & 1u) ? (cb[j] | (own[j] & 0x80000000u)) : ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int sp = SIGN > 0 ? i : 15 - i;
            if (AXIS == 0) __hip_atomic_fetch_or(&s_chg[sp * 8 + (r >> 1)], cm << ((r & 1) * 16 + 4 * q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (AXIS == 1) __hip_atomic_fetch_or(&s_chg[r * 8 + (sp >> 1)], cm << ((sp & 1) * 16 + 4 * q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else {
                __hip_atomic_fetch_or(&s_chg[r * 8 + 2 * q], ((cm & 1u) | ((cm & 2u) << 15)) << sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(&s_chg[r * 8 + 2 * q + 1], (((cm >> 2) & 1u) | ((cm & 8u) << 13)) << sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            lowered += __builtin_popcount(cm);
        }
        changed = changed || ch;
        if (entry_check && i == 0 && !__any(ch)) return false;
    }
    return changed;
}

__global__ void __launch_bounds__(384, ESDF_WPE) k_esdf_round(MapDev M, EsdfDev E, int s, float vs, float max_dist, int round)
{
    constexpr int NH = ESDF_T3 - TSL_BRK3, HPER = (NH + 383) / 384;      // 1736 halo entries, 5 per thread
    __shared__ uint32_t s_t[ES_TILE];                      // the tile (see above)
    __shared__ __attribute__((aligned(4))) uint16_t s_tgt16[256], s_neg16[256];     // per interior row (x, y): bit z = target / negative side
    __shared__ uint32_t s_chg[TSL_BRK3 / 32];              // interior voxels lowered in this visit
    __shared__ int s_nb[27];                               // pool index of the 27 bricks around (and including) this one, -1 = absent
    __shared__ int s_notify, s_flags;
    const uint32_t* const s_tgt = reinterpret_cast<const uint32_t*>(s_tgt16);
    const uint32_t* const s_neg = reinterpret_cast<const uint32_t*>(s_neg16);
    const int cur = round % 3, nxt = (round + 1) % 3, clr = (round + 2) % 3;
    const int n = E.ctr[2 + cur];
    if (blockIdx.x == 0 && threadIdx.x == 0) { E.ctr[2 + clr] = 0; if (n) { E.ctr[7] = round + 1; if (round < 16) E.ctr[240 + round] = n; } }      // list (round+2) was consumed in round-1
    if (n == 0) return;
    const float c1 = 1.0f * vs, c2 = sqrtf(2.0f) * vs, c3 = sqrtf(3.0f) * vs;                   // dense_esdf.py:286
    const int* list = E.work + (size_t)cur * E.cap;
    int* next = E.work + (size_t)nxt * E.cap;
    const int wave = (int)(threadIdx.x >> 6);
    // this thread's halo entries, the same for every brick: tile index | which of the 27 bricks << 13 | voxel in that brick << 18, packed so
    // that they cost HPER registers while the sweeps run (~0u: none)
    uint32_t hpk[HPER], hnb[HPER];
#pragma unroll
    for (int q = 0; q < HPER; ++q) {
        const int h = q * 384 + (int)threadIdx.x;
        int tx, ty, tz;
        if (h < 2 * ESDF_T * ESDF_T) { const int r = h % (ESDF_T * ESDF_T); tx = (h / (ESDF_T * ESDF_T)) * 17; ty = r / ESDF_T; tz = r % ESDF_T; }       // faces x = 0, 17
        else if (h < 2 * ESDF_T * ESDF_T + 16 * 2 * ESDF_T) { const int g = h - 2 * ESDF_T * ESDF_T, r = g % (2 * ESDF_T); tx = 1 + g / (2 * ESDF_T); ty = (r / ESDF_T) * 17; tz = r % ESDF_T; }   // rows y = 0, 17
        else { const int g = h - (2 * ESDF_T * ESDF_T + 16 * 2 * ESDF_T); tx = 1 + g / 32; ty = 1 + (g % 32) / 2; tz = (g & 1) * 17; }                // entries z = 0, 17
        const uint32_t hq = (uint32_t)((((tx + 15) >> 4) * 3 + ((ty + 15) >> 4)) * 3 + ((tz + 15) >> 4));
        const uint32_t vo = (uint32_t)((((tx + 15) & 15) << 8) | (((ty + 15) & 15) << 4) | ((tz + 15) & 15));
        hpk[q] = h < NH ? (uint32_t)(tx * ES_SX + ty * ES_SY + tz) | hq << 13 | vo << 18 : ~0u;
        // for the notification test: the interior voxels next to the entry are base + a * su + b * sv, a, b in {-1, 0, 1}, where the axes on
        // which the entry lies outside the brick are moved one step inwards (nf of them) and u, v are the others (stride code 0: none)
        const int fx = tx == 0 || tx == 17, fy = ty == 0 || ty == 17, fz = tz == 0 || tz == 17;
        const int ix = tx == 0 ? 1 : (tx == 17 ? 16 : tx), iy = ty == 0 ? 1 : (ty == 17 ? 16 : ty), iz = tz == 0 ? 1 : (tz == 17 ? 16 : tz);
        const int ucode = !fx ? 1 : (!fy ? 2 : (!fz ? 3 : 0)), vcode = !fx ? (!fy ? 2 : (!fz ? 3 : 0)) : ((!fy && !fz) ? 3 : 0);
        const int cu = ucode == 1 ? tx : (ucode == 2 ? ty : tz), cv = vcode == 2 ? ty : tz;
        hnb[q] = (uint32_t)(ix * ES_SX + iy * ES_SY + iz) | (uint32_t)ucode << 13 | (uint32_t)vcode << 15 | (uint32_t)cu << 17 | (uint32_t)cv << 22 | (uint32_t)(fx + fy + fz) << 27;
    }
    for (int w = blockIdx.x; w < n; w += gridDim.x) {
#ifdef TSL_TIMING
        long long _t = wall_clock64();
#endif
        const int p = list[w];
        if (threadIdx.x < 27) s_nb[threadIdx.x] = E.nbr[(size_t)p * 27 + threadIdx.x];          // written by k_esdf_init
        if (threadIdx.x == 0) { s_notify = 0; s_flags = 0; }
        if (threadIdx.x < TSL_BRK3 / 32) s_chg[threadIdx.x] = 0u;
        // first relaxation in this update: everything is new; afterwards only the halo entries of the neighbours that changed their
        // boundary layer since (the notification mask written for this round) are
        const bool first = E.region[p] == 1;
        uint32_t* const my_note = E.note + (size_t)(round & 1) * E.cap + p;
        const uint32_t note = first ? ~0u : *my_note;
        __syncthreads();
        ESDF_TICK(0);
        if (threadIdx.x == 0) *my_note = 0u;                        // this parity is written again in round + 1, after this launch
        // ---- stage brick + halo as ONE batch of independent loads: thread tid < 256 owns the interior row (x, y) = (tid / 16, tid % 16) --
        //      16 voxels = 4 + 1 wide loads -- and every thread <= 5 of the 1736 halo entries (loaded from a valid address unconditionally
        //      so that nothing separates the requests) ----
        int flags = 0;                                              // 1: a target, 2: a value that can lower a neighbour (value + edge < max_dist)
        {
            const int row = (int)threadIdx.x & 255;
            const size_t v0 = (size_t)p * TSL_BRK3 + (size_t)row * 16;
            uint4 dq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) dq[q] = reinterpret_cast<const uint4*>(E.mag + v0)[q];
            const uint4 fq = *reinterpret_cast<const uint4*>(E.fl + v0);
            uint32_t hd[HPER], hk[HPER];
#pragma unroll
            for (int q = 0; q < HPER; ++q) {
                hk[q] = hpk[q];
                asm volatile("" : "+v"(hk[q]));                    // (decode here, do not keep the decoded fields across the loop)
                const int np = hk[q] != ~0u ? s_nb[(hk[q] >> 13) & 31u] : -1;
                if (np < 0) hk[q] = ~0u;
                hd[q] = __float_as_uint(E.mag[(size_t)(np >= 0 ? np : p) * TSL_BRK3 + (np >= 0 ? hk[q] >> 18 : 0u)]);
            }
            if (threadIdx.x < 256) {
                const uint32_t dl[16] = { dq[0].x, dq[0].y, dq[0].z, dq[0].w, dq[1].x, dq[1].y, dq[1].z, dq[1].w, dq[2].x, dq[2].y, dq[2].z, dq[2].w, dq[3].x, dq[3].y, dq[3].z, dq[3].w };
                const uint32_t fw[4] = { fq.x, fq.y, fq.z, fq.w };
                const int t0 = ((int)(threadIdx.x >> 4) + 1) * ES_SX + ((int)(threadIdx.x & 15) + 1) * ES_SY + 1;
                uint32_t tg = 0u, ng = 0u;
#pragma unroll
                for (int z = 0; z < 16; ++z) {
                    const uint32_t f = (fw[z >> 2] >> ((z & 3) * 8)) & 0xffu;
                    s_t[t0 + z] = dl[z];
                    if (__uint_as_float(dl[z] & 0x7fffffffu) + c1 < max_dist) flags |= 2;          // (false for the NaN of an unobserved voxel)
                    tg |= ((f & (EF_NODE | EF_FIXED)) == EF_NODE ? 1u : 0u) << z;
                    ng |= ((f & EF_NEG) ? 1u : 0u) << z;
                }
                s_tgt16[threadIdx.x] = (uint16_t)tg; s_neg16[threadIdx.x] = (uint16_t)ng;
                if (tg != 0u) flags |= 1;
            }
#pragma unroll
            for (int q = 0; q < HPER; ++q) {
                if (hpk[q] == ~0u) continue;                      // (an entry of an absent brick is written as unobserved)
                const uint32_t hw = hk[q] != ~0u ? hd[q] : ES_UNOBS;
                s_t[hpk[q] & 0x1fffu] = hw;
                if (__uint_as_float(hw & 0x7fffffffu) + c1 < max_dist) flags |= 2;
            }
        }
        {
            const int f1 = __any(flags & 1) ? 1 : 0, f2 = __any(flags & 2) ? 2 : 0;
            if ((threadIdx.x & 63u) == 0 && (f1 | f2)) atomicOr(&s_flags, f1 | f2);
        }
        __syncthreads();
        // nothing to relax without a target, and nothing can be lowered if no value in the tile is an edge below max_dist (the outer
        // bricks of the region in round 0)
        const bool work = s_flags == 3;
        ESDF_TICK(1);
        // ---- the sweeps ----
        int lowered = 0, sets = 0;
        if (work) {
            // wave -> (axis, sign); on a later visit a sweep takes part in the entry check if a neighbour on its entry side was notified
            const uint32_t entry_mask = wave == 0 ? 0x1ffu : wave == 1 ? 0x1ffu << 18 : wave == 2 ? 0x01c0e07u : wave == 3 ? 0x01c0e07u << 6 : wave == 4 ? 0x1249249u : 0x1249249u << 2;
            unsigned long long mT, mN;
            switch (wave) {
            case 0: esdf_masks<0, +1>(s_tgt, s_neg, mT, mN); break;
            case 1: esdf_masks<0, -1>(s_tgt, s_neg, mT, mN); break;
            case 2: esdf_masks<1, +1>(s_tgt, s_neg, mT, mN); break;
            case 3: esdf_masks<1, -1>(s_tgt, s_neg, mT, mN); break;
            case 4: esdf_masks<2, +1>(s_tgt, s_neg, mT, mN); break;
            default: esdf_masks<2, -1>(s_tgt, s_neg, mT, mN); break;
            }
            bool full = first;
            for (;;) {
                bool chg = false;
                if (full || (note & entry_mask)) {
                    switch (wave) {
                    case 0: chg = esdf_sweep<0, +1>(s_t, s_chg, mT, mN, c1, c2, c3, !full, lowered); break;
                    case 1: chg = esdf_sweep<0, -1>(s_t, s_chg, mT, mN, c1, c2, c3, !full, lowered); break;
                    case 2: chg = esdf_sweep<1, +1>(s_t, s_chg, mT, mN, c1, c2, c3, !full, lowered); break;
                    case 3: chg = esdf_sweep<1, -1>(s_t, s_chg, mT, mN, c1, c2, c3, !full, lowered); break;
                    case 4: chg = esdf_sweep<2, +1>(s_t, s_chg, mT, mN, c1, c2, c3, !full, lowered); break;
                    default: chg = esdf_sweep<2, -1>(s_t, s_chg, mT, mN, c1, c2, c3, !full, lowered); break;
                    }
                }
                ++sets;
                if (!__syncthreads_or(chg)) break;
                full = true;
            }
        }
#ifdef TSL_TIMING
        if (threadIdx.x == 0) { const long long _n = wall_clock64(); const int pb = sets < 31 ? sets : 31;
            atomicAdd(&E.tm[8 + pb], (unsigned long long)(_n - _t)); atomicAdd(&E.tm[40 + pb], 1ull); atomicMax(&E.tm[72], (unsigned long long)(_n - _t)); }
        { const long long pw = wave_sum_ll((long long)lowered); if (lane_id() == 0) atomicAdd(&E.tm[80 + (sets < 31 ? sets : 31)], (unsigned long long)pw); }
#endif
        ESDF_TICK(2);
        // ---- write back the voxels that were lowered ----
        if (threadIdx.x < 256) {      // a thread per interior row: the row goes back whole (four 16-byte stores) when any of its voxels was lowered (round 6: a
                                      // store per lowered voxel, eleven bit tests per thread, took 4.3 us of a visit; this takes 1.2)
            const int row = (int)threadIdx.x;
            if ((s_chg[row >> 1] >> ((row & 1) * 16)) & 0xffffu) {
                const int t0 = ((row >> 4) + 1) * ES_SX + ((row & 15) + 1) * ES_SY + 1;
                uint32_t wv[16];
#pragma unroll
                for (int z = 0; z < 16; ++z) wv[z] = s_t[t0 + z];
                uint4* gm = reinterpret_cast<uint4*>(E.mag + (size_t)p * TSL_BRK3 + (size_t)row * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) gm[q] = make_uint4(wv[4 * q], wv[4 * q + 1], wv[4 * q + 2], wv[4 * q + 3]);
            }
        }
        // ---- which neighbours to tell: a neighbour is listed for the next round iff one of its voxels in this tile's halo could be lowered
        //      from this brick -- the pull of the sweeps, turned outwards: for a halo entry, the least value + edge cost over the <= 9
        //      interior voxels next to it, against the entry as it was staged.  (The entry may have been lowered by its owner since, and it
        //      may be a band voxel, which nothing lowers: then the neighbour is told for nothing and finds nothing to do.  It is never
        //      higher than staged.)  A brick that only took values from its inner neighbours does not call them back this way, and a brick
        //      of the region that was never told anything is never relaxed. ----
        if (first || __syncthreads_or(lowered != 0)) {
            int told = 0;
#pragma unroll
            for (int q = 0; q < HPER; ++q) {
                uint32_t hk = hpk[q], hn = hnb[q];
                asm volatile("" : "+v"(hk), "+v"(hn));
                if (hk == ~0u) continue;
                const uint32_t hw = ES_LD(&s_t[hk & 0x1fffu]);
                if (hw == ES_UNOBS) continue;
                const uint32_t sb = hw & 0x80000000u;
                const int base = (int)(hn & 0x1fffu), uc = (int)((hn >> 13) & 3u), vc = (int)((hn >> 15) & 3u), cu = (int)((hn >> 17) & 31u), cv = (int)((hn >> 22) & 31u), nf = (int)(hn >> 27);
                const int su = uc == 1 ? ES_SX : (uc == 2 ? ES_SY : (uc == 3 ? 1 : 0)), sv = vc == 2 ? ES_SY : (vc == 3 ? 1 : 0);
                uint32_t m[3] = { ~0u, ~0u, ~0u };                 // least same-side magnitude by the number of free axes stepped along
#pragma unroll
                for (int a = -1; a <= 1; ++a)
#pragma unroll
                    for (int b = -1; b <= 1; ++b) {
                        const bool ok = (a == 0 || (uc != 0 && cu + a >= 1 && cu + a <= 16)) && (b == 0 || (vc != 0 && cv + b >= 1 && cv + b <= 16));
                        const uint32_t x = ES_LD(&s_t[base + (ok ? a * su + b * sv : 0)]) ^ sb;
                        if (ok) m[(a != 0) + (b != 0)] = min(m[(a != 0) + (b != 0)], x);
                    }
                const float cc[5] = { 0.0f, c1, c2, c3, __uint_as_float(ES_INF) };
                float best = __uint_as_float(ES_INF);
#pragma unroll
                for (int e = 0; e < 3; ++e) best = fminf(best, __uint_as_float(min(m[e], ES_INF)) + cc[min(nf + e, 4)]);
                if (__float_as_uint(best) < (hw & 0x7fffffffu)) told |= 1 << ((hk >> 13) & 31u);
            }
            told &= ~(1 << 13);
            for (int d = 32; d >= 1; d >>= 1) told |= __shfl_xor(told, d);
            if ((threadIdx.x & 63u) == 0 && told) atomicOr(&s_notify, told);
        }
        __syncthreads();
        ESDF_TICK(3);
        {
            const long long lw = wave_sum_ll((long long)lowered);
            if (lane_id() == 0 && lw) __hip_atomic_fetch_add(ES_STAT(E, 1), (int)lw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the neighbours that see a changed boundary voxel go on the next round's list: one lane per neighbour (the region check, the
        // stamp exchange and the list reservation are dependent device-memory round trips)
        if (threadIdx.x < 64) {
            const int q = (int)threadIdx.x;
            bool put = false; int np = -1;
            if (q < 27 && ((s_notify >> q) & 1) && (np = s_nb[q]) >= 0) {
                // (the region mark and the stamp travel together: a stamp outside the region means nothing and is reset by the next collect)
                const uint8_t inside = E.region[np];
                const int seen = __hip_atomic_exchange(&E.stamp[np], round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (inside != 0) {                                                             // present and inside this update's region
                    // seen from the neighbour this brick is neighbour 26 - q: its halo entries from here are new in the next round
                    __hip_atomic_fetch_or(E.note + (size_t)((round + 1) & 1) * E.cap + np, 1u << (26 - q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    put = seen != round + 1;                                                   // not listed yet
                }
            }
            const int at = wave_reserve(&E.ctr[2 + nxt], put);
            if (put) next[at] = np;
        }
        if (threadIdx.x == 0) {
            E.region[p] = 2;
            __hip_atomic_fetch_add(ES_STAT(E, 0), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(ES_STAT(E, 2), sets, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(ES_STAT(E, 3), sets, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        ESDF_TICK(4);
    }
}

// =====================================================================================================================================
// esdf_mode 1 (round 6; opt-in, see the measurements at the end of this comment): the update as a RAISE / LOWER WAVEFRONT with parent directions, dense_esdf.py:255-333.
//
// The reference keeps `parent_dir` per voxel (:96): a lowered voxel remembers the neighbour it took its value from (:290, :296); when a
// voxel's value goes up, the raise queue visits exactly the voxels whose parent chain passes through it (:255-273), and the lower queue then
// repairs them from whatever is still valid (:275-299).  Here the same bookkeeping drives the brick pipeline:
//   * INVARIANT (between updates).  Every observed non-band voxel x either sits at max_dist without a parent, or has a parent n = x + dir on its
//     own side with  value(x) == fl(value(n) + |dir| * voxel)  ("supported"), and no neighbour offers less ("relaxed").  Supported chains
//     descend strictly and end in band voxels, so every value is a realisable path cost; relaxed + realisable == the least fixed point, the
//     same field the regional recompute (esdf_mode 0), the full recompute and the oracle's Dijkstra give, bit for bit.
//   * k_esdf_diff visits the bricks an integrate kernel wrote and compares each voxel's ESDF INPUTS (observed, side, band membership, band
//     value) with the stored ones: a band voxel takes its new value, a voxel that left the band / changed side / is new starts from
//     max_dist without a parent, everything else keeps value AND parent.  Only bricks where something changed start the wave; nothing is
//     dilated, nothing is re-initialised.
//   * A visit of a brick (k_esdf_wave) stages the 18^3 tile and the parent codes, then
//       RAISE: every voxel whose parent link no longer holds is re-derived THROUGH THAT LINK -- value := fl(value(parent) + cost), the value
//              its own chain gives it now; max_dist and no parent if the parent left its side -- parents before children, by six concurrent
//              directional sweeps that only follow parent links (a chain in free space advances along one dominant axis, so one sweep carries a
//              whole chain).  This is the reference's raise wave with one difference: a raised voxel is not thrown back to max_dist when its
//              chain still exists, it keeps the chain's new cost -- still a realisable upper bound, and for the common case (the band's f16
//              values drift by an ulp from frame to frame) already the final value.  Descendants inside the tile are all re-derived before
//              anything is lowered, so no voxel can be lowered from a stale descendant of itself inside a brick.
//       LOWER: the six directional pull sweeps of esdf_mode 0 (atomic min on the tile), to the local fixed point.
//       PARENTS of the voxels the lower wave wrote: the neighbour that supports the final value (one exists at the fixed point).
//     and writes back the voxels (and parent codes) that changed.
//   * The wave crosses bricks through the work lists: a brick lists neighbour B for the next round iff, seen from here, a voxel of B in this
//     tile's halo is unsupported (its parent is one of this brick's voxels and the link no longer holds: RAISE) or could be lowered from this
//     brick (LOWER) -- a test on B's STAGED state, exact when B was not visited in this round -- or B was visited in this same round (it may
//     have staged values this visit has since changed) and the layer of this brick next to B changed.  At quiescence every brick's last
//     visit saw its neighbours' final values or was followed by a neighbour's test on its final state: the invariant holds everywhere.
//     (A value that has to rise by more than its descendants' lead -- the surface under it vanished -- can borrow from a stale descendant in
//     ANOTHER brick and is then corrected round by round; should the rounds launched not suffice, the host repairs with a full recompute,
//     as for esdf_mode 0.)
// MEASURED (512^3 / 2 cm benchmark stream, max_dist 1 m, MI355X; profiles/r06_esdf_wavefront.txt): exact on every frame of every test stream; 0.96 M voxel writes
// per update (0.87 M re-derived by the raise wave, 0.10 M lowered) against 3.1 M for esdf_mode 0, 461 bricks reached against 705 -- and 0.51 ms per update
// against 0.35: the stream's f16 band values drift by an ulp every frame, so the wave passes through almost every voxel within max_dist of the visible
// surface anyway, and a visit costs a raise phase, a lower phase and a parent lookup where esdf_mode 0 only lowers (a wave's ~10 k instructions per visit at
// three waves per SIMD ARE the visit's latency, and eight rounds of them the update's).  A scene where a surface vanishes (a ball moving in front of the
// wall: tools/esdf_sparse_probe.py) makes values rise past their descendants in neighbouring bricks; those updates end in the full-recompute repair.
// esdf_mode 0 therefore stays the default; every ESDF test runs for both.
#define EP_NONE 0x15u
__device__ __forceinline__ int esdf_code_off(uint32_t code) { const int a = (int)(code >> 4), b = (int)((code >> 2) & 3u), c = (int)(code & 3u); return (a - 1) * ES_SX + (b - 1) * ES_SY + (c - 1); }
// what the parent link `code` of the voxel at tile index idx (word own) gives it now: side | fl(parent + cost) and the code, or side | max_dist and no parent
__device__ __forceinline__ void esdf_rederive(const uint32_t* s_t, int idx, uint32_t code, uint32_t own, float c1, float c2, float c3, float max_dist, uint32_t& nw, uint32_t& npar)
{
    const int a = (int)(code >> 4), b = (int)((code >> 2) & 3u), c = (int)(code & 3u);
    const int nz = (a != 1) + (b != 1) + (c != 1);
    const uint32_t pw = ES_LD(&s_t[idx + (a - 1) * ES_SX + (b - 1) * ES_SY + (c - 1)]);
    const bool ok = ((pw ^ own) & 0x80000000u) == 0u && (pw & 0x7fffffffu) <= ES_INF;          // the parent is observed and on this voxel's side
    const float tv = __uint_as_float(pw & 0x7fffffffu) + (nz == 1 ? c1 : (nz == 2 ? c2 : c3));
    const bool keep = ok && tv < max_dist;
    nw = (own & 0x80000000u) | (keep ? __float_as_uint(tv) : __float_as_uint(max_dist));
    npar = keep ? code : EP_NONE;
}

// 2'. the bricks an integrate kernel wrote: new ESDF inputs against the stored ones
__global__ void __launch_bounds__(256) k_esdf_diff(MapDev M, EsdfDev E, int s, int all, float gamma, float max_dist)
{
    const int nd = E.ctr[0];
    const uint32_t maxd = __float_as_uint(max_dist);
    for (int d = blockIdx.x; d < nd; d += gridDim.x) {
        const int pd = E.dirty[d];
        const bool fresh = all || E.ok[pd] == 0;              // nothing valid stored for this brick: everything counts as unobserved before
        const size_t v = (size_t)pd * TSL_BRK3 + (size_t)threadIdx.x * 16;
        const uint4 ob = *reinterpret_cast<const uint4*>(M.obs + v);
        uint4 fo = make_uint4(0u, 0u, 0u, 0u), po = make_uint4(0x15151515u, 0x15151515u, 0x15151515u, 0x15151515u);
        uint4 tw[4], mo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { tw[q] = reinterpret_cast<const uint4*>(M.tw + v)[q]; mo[q] = make_uint4(ES_UNOBS, ES_UNOBS, ES_UNOBS, ES_UNOBS); }
        if (!fresh) {
            fo = *reinterpret_cast<const uint4*>(E.fl + v); po = *reinterpret_cast<const uint4*>(E.par + v);
#pragma unroll
            for (int q = 0; q < 4; ++q) mo[q] = reinterpret_cast<const uint4*>(E.mag + v)[q];
        }
        const uint32_t ow[4] = { ob.x, ob.y, ob.z, ob.w }, fw[4] = { fo.x, fo.y, fo.z, fo.w }, pw[4] = { po.x, po.y, po.z, po.w };
        const uint32_t tv[16] = { tw[0].x, tw[0].y, tw[0].z, tw[0].w, tw[1].x, tw[1].y, tw[1].z, tw[1].w, tw[2].x, tw[2].y, tw[2].z, tw[2].w, tw[3].x, tw[3].y, tw[3].z, tw[3].w };
        const uint32_t mv[16] = { mo[0].x, mo[0].y, mo[0].z, mo[0].w, mo[1].x, mo[1].y, mo[1].z, mo[1].w, mo[2].x, mo[2].y, mo[2].z, mo[2].w, mo[3].x, mo[3].y, mo[3].z, mo[3].w };
        uint32_t nf[4] = { 0u, 0u, 0u, 0u }, np[4] = { 0u, 0u, 0u, 0u }, nm[16];
        bool changed = false, seed = false;
#pragma unroll
        for (int z = 0; z < 16; ++z) {
            uint32_t f; float mg;
            esdf_inputs((ow[z >> 2] >> ((z & 3) * 8)) & 0xffu, tv[z], gamma, max_dist, &f, &mg);
            const uint32_t fs = (fw[z >> 2] >> ((z & 3) * 8)) & 0xffu, ps = (pw[z >> 2] >> ((z & 3) * 8)) & 0xffu;
            const uint32_t side = (f & EF_NEG) ? 0x80000000u : 0u;
            uint32_t w, pc = EP_NONE;
            if (!(f & EF_NODE)) w = ES_UNOBS;
            else if (f & EF_FIXED) w = __float_as_uint(mg) | side;                              // band: the TSDF value itself (:313-317)
            else if (fs == f) { w = mv[z]; pc = ps; }                                           // same class as before, outside the band: value and parent stand
            else w = maxd | side;                                                               // new / left the band / changed side (:325, :329): to be lowered
            changed = changed || w != mv[z] || fs != f;
            seed = seed || (f & EF_FIXED);
            nf[z >> 2] |= f << ((z & 3) * 8); np[z >> 2] |= pc << ((z & 3) * 8); nm[z] = w;
        }
        if (changed || fresh) {                                 // (a brick without a valid state is written whole: its arrays hold whatever an earlier map left)
            *reinterpret_cast<uint4*>(E.fl + v) = make_uint4(nf[0], nf[1], nf[2], nf[3]);
            *reinterpret_cast<uint4*>(E.par + v) = make_uint4(np[0], np[1], np[2], np[3]);
#pragma unroll
            for (int q = 0; q < 4; ++q) reinterpret_cast<uint4*>(E.mag + v)[q] = make_uint4(nm[4 * q], nm[4 * q + 1], nm[4 * q + 2], nm[4 * q + 3]);
        }
        const bool any = __syncthreads_or(changed) != 0;
        const bool band = __syncthreads_or(seed) != 0;
        if (threadIdx.x == 0) {
            if (fresh) E.ok[pd] = 1;
            if (any) {
                E.region[pd] = 1;                               // its first visit of this update looks at everything
                __hip_atomic_fetch_add(ES_STAT(E, 5), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // a full recompute starts from the bricks that hold a source; the others wait until a neighbour reports something for them
                if (!all || band) { const int q = __hip_atomic_fetch_add(&E.ctr[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); E.work[q] = pd; E.stamp[pd] = 0; }
            }
        }
    }
}

// the RAISE sweep of the calling wave: voxels whose parent lies in the plane before (along AXIS, direction SIGN) take the value their link gives them now
template <int AXIS, int SIGN>
__device__ __forceinline__ void esdf_rsweep(uint32_t* s_t, uint8_t* s_par, uint32_t* s_chg, unsigned long long mT, const float c1, const float c2, const float c3, const float max_dist, int& raised)
{
    const int lane = (int)(threadIdx.x & 63u), q = lane & 3, r = lane >> 2;
    constexpr uint32_t WANT = SIGN > 0 ? 0u : 2u;                       // the link's component along AXIS: parent one plane back
    constexpr int SH = AXIS == 0 ? 4 : (AXIS == 1 ? 2 : 0);
#pragma unroll 1
    for (int i = 0; i < 16; ++i, mT >>= 4) {
        const uint32_t tn = (uint32_t)mT & 15u;
        const int sp = SIGN > 0 ? i : 15 - i;
        // the four voxels of this lane in the plane: interior index l0 + j * LJ, tile index t0 + j * TJ
        const int x = AXIS == 0 ? sp : r, y0 = AXIS == 0 ? r : (AXIS == 1 ? sp : 4 * q), z0 = AXIS == 2 ? sp : 4 * q;
        constexpr int LJ = AXIS == 2 ? 16 : 1, TJ = AXIS == 2 ? ES_SY : 1;
        const int l0 = (x << 8) | (y0 << 4) | z0, t0 = (x + 1) * ES_SX + (y0 + 1) * ES_SY + z0 + 1;
        uint32_t cw;                                                       // their parent codes
        if (AXIS != 2) cw = *reinterpret_cast<const volatile uint32_t*>(s_par + l0);
        else cw = (uint32_t)s_par[l0] | (uint32_t)s_par[l0 + 16] << 8 | (uint32_t)s_par[l0 + 32] << 16 | (uint32_t)s_par[l0 + 48] << 24;
        uint32_t el = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t code = (cw >> (8 * j)) & 0xffu; el |= ((code != EP_NONE && ((code >> SH) & 3u) == WANT) ? 1u : 0u) << j; }
        el &= tn;
        if (!__any(el != 0u)) continue;
        uint32_t cm = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t code = (el >> j) & 1u ? (cw >> (8 * j)) & 0xffu : EP_NONE;          // (a voxel that does not take part reads itself)
            const int idx = t0 + j * TJ;
            const uint32_t own = ES_LD(&s_t[idx]);
            uint32_t nw, npar;
            esdf_rederive(s_t, idx, code, own, c1, c2, c3, max_dist, nw, npar);
            if (((el >> j) & 1u) && nw != own) {
                __hip_atomic_store(&s_t[idx], nw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (npar != code) s_par[l0 + j * LJ] = (uint8_t)npar;
                cm |= 1u << j;
            }
        }
        if (cm) {                                                          // one OR per lane (the four bits of a lane share a word, or two for the z sweeps), as esdf_sweep does
            if (AXIS == 0) __hip_atomic_fetch_or(&s_chg[sp * 8 + (r >> 1)], cm << ((r & 1) * 16 + 4 * q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (AXIS == 1) __hip_atomic_fetch_or(&s_chg[r * 8 + (sp >> 1)], cm << ((sp & 1) * 16 + 4 * q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else {
                if (cm & 3u) __hip_atomic_fetch_or(&s_chg[r * 8 + 2 * q], ((cm & 1u) | ((cm & 2u) << 15)) << sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cm & 12u) __hip_atomic_fetch_or(&s_chg[r * 8 + 2 * q + 1], (((cm >> 2) & 1u) | ((cm & 8u) << 13)) << sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        raised += __builtin_popcount(cm);
    }
}

__global__ void __launch_bounds__(384, ESDF_WPE) k_esdf_wave(MapDev M, EsdfDev E, int s, float vs, float max_dist, int round)
{
    constexpr int NH = ESDF_T3 - TSL_BRK3, HPER = (NH + 383) / 384;      // 1736 halo entries, 5 per thread
    __shared__ uint32_t s_t[ES_TILE];                      // the tile: side << 31 | magnitude, ES_UNOBS where unobserved
    __shared__ __attribute__((aligned(16))) uint8_t s_par[TSL_BRK3];   // parent codes of the interior voxels
    __shared__ uint8_t s_hp[HPER * 384];                   // parent codes of the halo entries (for the notification test)
    __shared__ __attribute__((aligned(4))) uint16_t s_tgt16[256], s_neg16[256];
    __shared__ uint32_t s_chg[TSL_BRK3 / 32];              // interior voxels lowered in this visit (their parents are looked up afterwards)
    __shared__ uint32_t s_chr[TSL_BRK3 / 32];              // interior voxels re-derived by the raise sweeps
    __shared__ uint32_t s_fr[TSL_BRK3 / 32], s_cur[TSL_BRK3 / 32];      // the lower wave's frontier: voxels lowered whose neighbourhood has not been looked at since
    __shared__ uint16_t s_list[TSL_BRK3];                  // a compacted voxel list (frontier / the voxels whose parents are looked up)
    __shared__ int s_cnt;
    __shared__ int s_nb[27];
    __shared__ int s_notify, s_flags, s_layer;
    const uint32_t* const s_tgt = reinterpret_cast<const uint32_t*>(s_tgt16);
    const uint32_t* const s_neg = reinterpret_cast<const uint32_t*>(s_neg16);
    const int cur = round % 3, nxt = (round + 1) % 3, clr = (round + 2) % 3;
    const int n = E.ctr[2 + cur];
    if (blockIdx.x == 0 && threadIdx.x == 0) { E.ctr[2 + clr] = 0; if (n) { E.ctr[7] = round + 1; if (round < 16) E.ctr[240 + round] = n; } }
    if (n == 0) return;
    const float c1 = 1.0f * vs, c2 = sqrtf(2.0f) * vs, c3 = sqrtf(3.0f) * vs;                   // dense_esdf.py:286
    const int* list = E.work + (size_t)cur * E.cap;
    int* next = E.work + (size_t)nxt * E.cap;
    const int wave = (int)(threadIdx.x >> 6);
    uint32_t hpk[HPER], hnb[HPER];                        // as in k_esdf_round: tile index | brick of 27 << 13 | voxel in it << 18; the interior voxels next to the entry
#pragma unroll
    for (int q = 0; q < HPER; ++q) {
        const int h = q * 384 + (int)threadIdx.x;
        int tx, ty, tz;
        if (h < 2 * ESDF_T * ESDF_T) { const int r = h % (ESDF_T * ESDF_T); tx = (h / (ESDF_T * ESDF_T)) * 17; ty = r / ESDF_T; tz = r % ESDF_T; }
        else if (h < 2 * ESDF_T * ESDF_T + 16 * 2 * ESDF_T) { const int g = h - 2 * ESDF_T * ESDF_T, r = g % (2 * ESDF_T); tx = 1 + g / (2 * ESDF_T); ty = (r / ESDF_T) * 17; tz = r % ESDF_T; }
        else { const int g = h - (2 * ESDF_T * ESDF_T + 16 * 2 * ESDF_T); tx = 1 + g / 32; ty = 1 + (g % 32) / 2; tz = (g & 1) * 17; }
        const uint32_t hq = (uint32_t)((((tx + 15) >> 4) * 3 + ((ty + 15) >> 4)) * 3 + ((tz + 15) >> 4));
        const uint32_t vo = (uint32_t)((((tx + 15) & 15) << 8) | (((ty + 15) & 15) << 4) | ((tz + 15) & 15));
        hpk[q] = h < NH ? (uint32_t)(tx * ES_SX + ty * ES_SY + tz) | hq << 13 | vo << 18 : ~0u;
        const int fx = tx == 0 || tx == 17, fy = ty == 0 || ty == 17, fz = tz == 0 || tz == 17;
        const int ix = tx == 0 ? 1 : (tx == 17 ? 16 : tx), iy = ty == 0 ? 1 : (ty == 17 ? 16 : ty), iz = tz == 0 ? 1 : (tz == 17 ? 16 : tz);
        const int ucode = !fx ? 1 : (!fy ? 2 : (!fz ? 3 : 0)), vcode = !fx ? (!fy ? 2 : (!fz ? 3 : 0)) : ((!fy && !fz) ? 3 : 0);
        const int cu = ucode == 1 ? tx : (ucode == 2 ? ty : tz), cv = vcode == 2 ? ty : tz;
        hnb[q] = (uint32_t)(ix * ES_SX + iy * ES_SY + iz) | (uint32_t)ucode << 13 | (uint32_t)vcode << 15 | (uint32_t)cu << 17 | (uint32_t)cv << 22 | (uint32_t)(fx + fy + fz) << 27;
    }
    for (int w = blockIdx.x; w < n; w += gridDim.x) {
#ifdef TSL_TIMING
        long long _t = wall_clock64(); const long long _t0 = _t;
#endif
        const int p = list[w];
        if (threadIdx.x < 27) s_nb[threadIdx.x] = E.nbr[(size_t)p * 27 + threadIdx.x];          // written by this update's k_esdf_collect
        if (threadIdx.x == 0) { s_notify = 0; s_flags = 0; s_layer = 0; }
        if (threadIdx.x < TSL_BRK3 / 32) { s_chg[threadIdx.x] = 0u; s_chr[threadIdx.x] = 0u; }
        const bool first = E.region[p] == 1;                        // changed by k_esdf_diff and not visited since: every link and every plane is looked at
        uint32_t* const my_note = E.note + (size_t)(round & 1) * E.cap + p;
        const uint32_t note = first ? ~0u : *my_note;
        __syncthreads();
        ESDF_TICKF(0);
        if (threadIdx.x == 0) *my_note = 0u;
        // ---- stage brick + halo, values and parent codes, as ONE batch of independent loads ----
        int flags = 0;
        {
            const int row = (int)threadIdx.x & 255;
            const size_t v0 = (size_t)p * TSL_BRK3 + (size_t)row * 16;
            uint4 dq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) dq[q] = reinterpret_cast<const uint4*>(E.mag + v0)[q];
            const uint4 fq = *reinterpret_cast<const uint4*>(E.fl + v0);
            const uint4 pq = *reinterpret_cast<const uint4*>(E.par + v0);
            uint32_t hd[HPER], hk[HPER]; uint8_t hc[HPER];
#pragma unroll
            for (int q = 0; q < HPER; ++q) {
                hk[q] = hpk[q];
                asm volatile("" : "+v"(hk[q]));
                const int np = hk[q] != ~0u ? s_nb[(hk[q] >> 13) & 31u] : -1;
                if (np < 0) hk[q] = ~0u;
                const size_t at = (size_t)(np >= 0 ? np : p) * TSL_BRK3 + (np >= 0 ? hk[q] >> 18 : 0u);
                hd[q] = __float_as_uint(E.mag[at]); hc[q] = E.par[at];
            }
            if (threadIdx.x < 256) {
                const uint32_t dl[16] = { dq[0].x, dq[0].y, dq[0].z, dq[0].w, dq[1].x, dq[1].y, dq[1].z, dq[1].w, dq[2].x, dq[2].y, dq[2].z, dq[2].w, dq[3].x, dq[3].y, dq[3].z, dq[3].w };
                const uint32_t fw[4] = { fq.x, fq.y, fq.z, fq.w };
                const int t0 = ((int)(threadIdx.x >> 4) + 1) * ES_SX + ((int)(threadIdx.x & 15) + 1) * ES_SY + 1;
                uint32_t tg = 0u, ng = 0u;
#pragma unroll
                for (int z = 0; z < 16; ++z) {
                    const uint32_t f = (fw[z >> 2] >> ((z & 3) * 8)) & 0xffu;
                    s_t[t0 + z] = dl[z];
                    if (__uint_as_float(dl[z] & 0x7fffffffu) + c1 < max_dist) flags |= 2;
                    tg |= ((f & (EF_NODE | EF_FIXED)) == EF_NODE ? 1u : 0u) << z;
                    ng |= ((f & EF_NEG) ? 1u : 0u) << z;
                }
                s_tgt16[threadIdx.x] = (uint16_t)tg; s_neg16[threadIdx.x] = (uint16_t)ng;
                *reinterpret_cast<uint4*>(s_par + threadIdx.x * 16) = pq;
                if (tg != 0u) flags |= 1;
            }
#pragma unroll
            for (int q = 0; q < HPER; ++q) {
                if (hpk[q] == ~0u) continue;
                const uint32_t hw = hk[q] != ~0u ? hd[q] : ES_UNOBS;
                s_t[hpk[q] & 0x1fffu] = hw;
                s_hp[q * 384 + threadIdx.x] = hk[q] != ~0u ? hc[q] : (uint8_t)EP_NONE;
                if (__uint_as_float(hw & 0x7fffffffu) + c1 < max_dist) flags |= 2;
            }
        }
        {
            const int f1 = __any(flags & 1) ? 1 : 0, f2 = __any(flags & 2) ? 2 : 0;
            if ((threadIdx.x & 63u) == 0 && (f1 | f2)) atomicOr(&s_flags, f1 | f2);
        }
        __syncthreads();
        const bool targets = (s_flags & 1) != 0, work = s_flags == 3;
        ESDF_TICKF(1);
        int lowered = 0, raised = 0, sets = 0, rsets = 0;
        unsigned long long mT = 0ull, mN = 0ull;
        if (targets) {
            switch (wave) {
            case 0: esdf_masks<0, +1>(s_tgt, s_neg, mT, mN); break;
            case 1: esdf_masks<0, -1>(s_tgt, s_neg, mT, mN); break;
            case 2: esdf_masks<1, +1>(s_tgt, s_neg, mT, mN); break;
            case 3: esdf_masks<1, -1>(s_tgt, s_neg, mT, mN); break;
            case 4: esdf_masks<2, +1>(s_tgt, s_neg, mT, mN); break;
            default: esdf_masks<2, -1>(s_tgt, s_neg, mT, mN); break;
            }
            // ---- RAISE: links that no longer hold, re-derived parents first.  A flat check of every link decides whether a set of sweeps is needed
            //      (and, behind one, whether it sufficed: the six waves race, and a chain that bends against its sweep takes another set) ----
            for (int rs = 0; ; ++rs) {
                // a flat pass over every link: broken ones are re-derived on the spot (each voxel by the one thread that owns it).  The first pass
                // tells whether anything is broken at all; a set of sweeps then carries whole chains; the passes behind it mend what the racing sweeps
                // left (a child re-derived before its parent) -- shallow, so a few passes do what a second set of sweeps did at ten times the cost
                bool broken = false;
                for (int g = threadIdx.x; g < TSL_BRK3 / 4; g += 384) {
                    const uint32_t cw = *reinterpret_cast<const volatile uint32_t*>(s_par + 4 * g);
                    if (cw == 0x15151515u) continue;
                    const int l = 4 * g, t0 = ((l >> 8) + 1) * ES_SX + (((l >> 4) & 15) + 1) * ES_SY + (l & 15) + 1;
                    uint32_t cm = 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t code = (cw >> (8 * j)) & 0xffu;
                        const uint32_t own = ES_LD(&s_t[t0 + j]);
                        uint32_t nw, npar;
                        esdf_rederive(s_t, t0 + j, code, own, c1, c2, c3, max_dist, nw, npar);          // (no parent: the voxel reads itself and fails the test below)
                        if (code != EP_NONE && nw != own) {
                            __hip_atomic_store(&s_t[t0 + j], nw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (npar != code) s_par[l + j] = (uint8_t)npar;
                            cm |= 1u << j;
                        }
                    }
                    if (cm) { __hip_atomic_fetch_or(&s_chr[l >> 5], cm << (l & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); raised += __builtin_popcount(cm); broken = true; }
                }
                if (!__syncthreads_or(broken)) break;
                if (rs >= 200) { if (threadIdx.x == 0) E.ctr[8] = 1; break; }          // (never seen: the host repairs with a full recompute)
                if (rs % 8 != 0) continue;                                  // sweeps behind the first pass, and again should the passes not settle.  (Mending a handful of
                                                                            // broken links by flat passes alone was tried: a moved halo value changes a whole chain, one
                                                                            // level per pass -- 0.65 ms per update against 0.50)
                ++rsets;
                switch (wave) {
                case 0: esdf_rsweep<0, +1>(s_t, s_par, s_chr, mT, c1, c2, c3, max_dist, raised); break;
                case 1: esdf_rsweep<0, -1>(s_t, s_par, s_chr, mT, c1, c2, c3, max_dist, raised); break;
                case 2: esdf_rsweep<1, +1>(s_t, s_par, s_chr, mT, c1, c2, c3, max_dist, raised); break;
                case 3: esdf_rsweep<1, -1>(s_t, s_par, s_chr, mT, c1, c2, c3, max_dist, raised); break;
                case 4: esdf_rsweep<2, +1>(s_t, s_par, s_chr, mT, c1, c2, c3, max_dist, raised); break;
                default: esdf_rsweep<2, -1>(s_t, s_par, s_chr, mT, c1, c2, c3, max_dist, raised); break;
                }
                __syncthreads();
            }
        }
        ESDF_TICKF(5);
        // ---- LOWER.  Everything in the tile is a realisable value now.  What can still be lowered lies around what changed: the halo entries of the
        //      notified neighbours (the entering sweeps of esdf_mode 0), the voxels the raise wave re-derived, and whatever gets lowered on the way.
        //      A visit that changed much (the first one of a brick k_esdf_diff touched, a recompute) sweeps all six directions over all planes once;
        //      the rest is a push / pull FRONTIER around the voxels that changed: (x -> y) pairs with neither end changed were relaxed before ----
        // (the staging's "some value can lower a neighbour" flag was taken before the raise wave, which may have brought a value below that mark)
        if (targets && (work || __syncthreads_or(raised != 0))) {
            const uint32_t entry_mask = wave == 0 ? 0x1ffu : wave == 1 ? 0x1ffu << 18 : wave == 2 ? 0x01c0e07u : wave == 3 ? 0x01c0e07u << 6 : wave == 4 ? 0x1249249u : 0x1249249u << 2;
            if (threadIdx.x == 0) s_cnt = 0;
            __syncthreads();
            if (threadIdx.x < TSL_BRK3 / 32) { const uint32_t f = s_chr[threadIdx.x]; if (f) atomicAdd(&s_cnt, __builtin_popcount(f)); }
            __syncthreads();
            const int nraised = s_cnt;
            bool full = first || nraised > 0;          // (a push / pull frontier around a few re-derived voxels instead of the set was tried: level by level it is slower than one sweep)
            const bool local = !full && nraised > 0;             // few voxels re-derived: look around them instead of sweeping everything
            bool classic = false;
            for (;;) {
                bool chg = false;
                if (full || (note & entry_mask)) {
                    switch (wave) {
                    case 0: chg = esdf_sweep<0, +1>(s_t, s_chg, mT, mN, c1, c2, c3, !full, lowered); break;
                    case 1: chg = esdf_sweep<0, -1>(s_t, s_chg, mT, mN, c1, c2, c3, !full, lowered); break;
                    case 2: chg = esdf_sweep<1, +1>(s_t, s_chg, mT, mN, c1, c2, c3, !full, lowered); break;
                    case 3: chg = esdf_sweep<1, -1>(s_t, s_chg, mT, mN, c1, c2, c3, !full, lowered); break;
                    case 4: chg = esdf_sweep<2, +1>(s_t, s_chg, mT, mN, c1, c2, c3, !full, lowered); break;
                    default: chg = esdf_sweep<2, -1>(s_t, s_chg, mT, mN, c1, c2, c3, !full, lowered); break;
                    }
                }
                ++sets;
                if (!__syncthreads_or(chg) && !(local && !classic && sets == 1)) break;
                if (classic) { full = true; continue; }
                // the frontier: every voxel lowered so far (push to its neighbours) and, first time round, every voxel the raise wave re-derived (pull from
                // its neighbours -- it may have risen -- and push -- it may have fallen)
                if (threadIdx.x == 0) s_cnt = 0;
                __syncthreads();
                if (threadIdx.x < TSL_BRK3 / 32) {
                    const uint32_t f = s_chg[threadIdx.x] | (local ? s_chr[threadIdx.x] : 0u);
                    s_cur[threadIdx.x] = f; s_fr[threadIdx.x] = 0u; if (f) atomicAdd(&s_cnt, __builtin_popcount(f));
                }
                __syncthreads();
                if (s_cnt > 1024) { classic = true; full = true; continue; }          // a recompute from scratch: sets of sweeps to the fixed point, as esdf_mode 0
                for (int it = 0; ; ++it) {
                    const int nfr = s_cnt;
                    __syncthreads();
                    if (threadIdx.x == 0) s_cnt = 0;
                    __syncthreads();
                    if (threadIdx.x < TSL_BRK3 / 32) {
                        uint32_t f = s_cur[threadIdx.x];
                        if (f) { int at = atomicAdd(&s_cnt, __builtin_popcount(f)); while (f) { const int b = __builtin_ctz(f); s_list[at++] = (uint16_t)(threadIdx.x * 32 + b); f &= f - 1u; } }
                    }
                    __syncthreads();
                    bool more = false;
                    for (int k = threadIdx.x; k < nfr; k += 384) {
                        const int l = s_list[k], x = l >> 8, y = (l >> 4) & 15, z = l & 15;
                        const int idx = (x + 1) * ES_SX + (y + 1) * ES_SY + z + 1;
                        uint32_t own = ES_LD(&s_t[idx]);
                        const bool pull = local && it == 0 && ((s_chr[l >> 5] >> (l & 31)) & 1u) && ((s_tgt16[l >> 4] >> (l & 15)) & 1u);
                        uint32_t best = own & 0x7fffffffu;                     // pull: the least neighbour value + edge cost on this voxel's side
#pragma unroll
                        for (int a = -1; a <= 1; ++a)
#pragma unroll
                            for (int b = -1; b <= 1; ++b)
#pragma unroll
                                for (int c = -1; c <= 1; ++c) {
                                    if (a == 0 && b == 0 && c == 0) continue;
                                    const uint32_t w = ES_LD(&s_t[idx + a * ES_SX + b * ES_SY + c]);
                                    const int nzc = (a != 0) + (b != 0) + (c != 0);
                                    const float cc = nzc == 1 ? c1 : (nzc == 2 ? c2 : c3);
                                    const bool same = ((w ^ own) & 0x80000000u) == 0u && (w & 0x7fffffffu) <= ES_INF;
                                    const float up = __uint_as_float(w & 0x7fffffffu) + cc;
                                    if (pull && same && __float_as_uint(up) < best) best = __float_as_uint(up);
                                }
                        if (best < (own & 0x7fffffffu)) {
                            __hip_atomic_fetch_min(&s_t[idx], best | (own & 0x80000000u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_or(&s_chg[l >> 5], 1u << (l & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            own = best | (own & 0x80000000u); ++lowered;
                        }
                        const float om = __uint_as_float(own & 0x7fffffffu);
#pragma unroll
                        for (int a = -1; a <= 1; ++a)
#pragma unroll
                            for (int b = -1; b <= 1; ++b)
#pragma unroll
                                for (int c = -1; c <= 1; ++c) {
                                    if (a == 0 && b == 0 && c == 0) continue;
                                    const int nx = x + a, ny = y + b, nz_ = z + c;
                                    const bool in = (unsigned)nx < 16u && (unsigned)ny < 16u && (unsigned)nz_ < 16u;
                                    const int nl = in ? (nx << 8) | (ny << 4) | nz_ : l;
                                    const bool tgt = in && ((s_tgt16[nl >> 4] >> (nl & 15)) & 1u);
                                    const int nidx = idx + (in ? a * ES_SX + b * ES_SY + c : 0);
                                    const uint32_t w = ES_LD(&s_t[nidx]);
                                    const int nzc = (a != 0) + (b != 0) + (c != 0);
                                    const float cand = om + (nzc == 1 ? c1 : (nzc == 2 ? c2 : c3));
                                    if (tgt && ((w ^ own) & 0x80000000u) == 0u && __float_as_uint(cand) < (w & 0x7fffffffu)) {
                                        __hip_atomic_fetch_min(&s_t[nidx], __float_as_uint(cand) | (w & 0x80000000u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        __hip_atomic_fetch_or(&s_chg[nl >> 5], 1u << (nl & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        __hip_atomic_fetch_or(&s_fr[nl >> 5], 1u << (nl & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        ++lowered; more = true;
                                    }
                                }
                    }
                    if (!__syncthreads_or(more)) break;
                    if (threadIdx.x == 0) s_cnt = 0;
                    __syncthreads();
                    if (threadIdx.x < TSL_BRK3 / 32) { const uint32_t f = s_fr[threadIdx.x]; s_cur[threadIdx.x] = f; s_fr[threadIdx.x] = 0u; if (f) atomicAdd(&s_cnt, __builtin_popcount(f)); }
                    __syncthreads();
                }
                break;
            }
        }
        ESDF_TICKF(2);
        // ---- PARENTS of the voxels the lower wave wrote: the neighbour whose value + edge cost IS the voxel's value.  At the local fixed point the
        //      neighbour a voxel was last lowered from still offers exactly that (it can only have been lowered since, and then the voxel with it) ----
        int orphans = 0;
        if (__syncthreads_or(lowered != 0)) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        if (threadIdx.x < TSL_BRK3 / 32) {
            uint32_t f = s_chg[threadIdx.x];
            if (f) { int at = atomicAdd(&s_cnt, __builtin_popcount(f)); while (f) { const int b = __builtin_ctz(f); s_list[at++] = (uint16_t)(threadIdx.x * 32 + b); f &= f - 1u; } }
        }
        __syncthreads();
        const int nlow = s_cnt;
        for (int k = threadIdx.x; k < nlow; k += 384) {
            const int l = s_list[k];
            const int idx = ((l >> 8) + 1) * ES_SX + (((l >> 4) & 15) + 1) * ES_SY + (l & 15) + 1;
            const uint32_t own = ES_LD(&s_t[idx]);
            uint32_t found = EP_NONE;
            const uint32_t om = own & 0x7fffffffu;
            // all 26 candidates as independent LDS reads; the lowest code that supports the value wins (any supporter will do: the choice is only made reproducible)
#pragma unroll
            for (int a = 2; a >= 0; --a)
#pragma unroll
                for (int b = 2; b >= 0; --b)
#pragma unroll
                    for (int c = 2; c >= 0; --c) {
                        if (a == 1 && b == 1 && c == 1) continue;
                        const uint32_t pw = ES_LD(&s_t[idx + (a - 1) * ES_SX + (b - 1) * ES_SY + (c - 1)]);
                        const int nz = (a != 1) + (b != 1) + (c != 1);
                        const float tv = __uint_as_float(pw & 0x7fffffffu) + (nz == 1 ? c1 : (nz == 2 ? c2 : c3));
                        const bool hit = ((pw ^ own) & 0x80000000u) == 0u && (pw & 0x7fffffffu) <= ES_INF && __float_as_uint(tv) == om && tv < max_dist;
                        found = hit ? (uint32_t)(a << 4 | b << 2 | c) : found;
                    }
            if (found == EP_NONE) ++orphans;
            s_par[l] = (uint8_t)found;
        }
        }
        __syncthreads();
        ESDF_TICKF(6);
        // ---- write back the rows that changed (values + parent codes), and note which of the 26 outer layers did ----
        if (threadIdx.x < 256) {
            const int row = (int)threadIdx.x, x = row >> 4, y = row & 15;
            const uint32_t m16 = ((s_chg[row >> 1] | s_chr[row >> 1]) >> ((row & 1) * 16)) & 0xffffu;
            if (m16) {
                const int t0 = (x + 1) * ES_SX + (y + 1) * ES_SY + 1;
                uint32_t wv[16];
#pragma unroll
                for (int z = 0; z < 16; ++z) wv[z] = s_t[t0 + z];
                const size_t v0 = (size_t)p * TSL_BRK3 + (size_t)row * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) reinterpret_cast<uint4*>(E.mag + v0)[q] = make_uint4(wv[4 * q], wv[4 * q + 1], wv[4 * q + 2], wv[4 * q + 3]);
                *reinterpret_cast<uint4*>(E.par + v0) = *reinterpret_cast<const uint4*>(s_par + row * 16);
                // the neighbours (qx, qy, qz in {0, 1, 2}) whose halo holds a voxel of this row that changed
                int lay = 0;
#pragma unroll
                for (int qx = 0; qx < 3; ++qx)
#pragma unroll
                    for (int qy = 0; qy < 3; ++qy)
#pragma unroll
                        for (int qz = 0; qz < 3; ++qz) {
                            const bool okx = qx == 1 || (qx == 0 ? x == 0 : x == 15), oky = qy == 1 || (qy == 0 ? y == 0 : y == 15);
                            const uint32_t zm = qz == 1 ? m16 : (qz == 0 ? (m16 & 1u) : (m16 >> 15));
                            if (okx && oky && zm) lay |= 1 << ((qx * 3 + qy) * 3 + qz);
                        }
                atomicOr(&s_layer, lay & ~(1 << 13));
            }
        }
        ESDF_TICKF(7);
        // ---- which neighbours to tell.  Seen from here a voxel z of a neighbour (an entry of this tile's halo, as STAGED) needs a visit of its brick iff
        //      RAISE: its parent is a voxel of this brick and the link no longer holds, or LOWER: a voxel of this brick now offers it less ----
        if (first || __syncthreads_or((lowered | raised) != 0)) {
            int told = 0;
#pragma unroll
            for (int q = 0; q < HPER; ++q) {
                uint32_t hk = hpk[q], hn = hnb[q];
                asm volatile("" : "+v"(hk), "+v"(hn));
                if (hk == ~0u) continue;
                const int hidx = (int)(hk & 0x1fffu);
                const uint32_t hw = ES_LD(&s_t[hidx]);
                if (hw == ES_UNOBS) continue;
                const uint32_t hcode = s_hp[q * 384 + threadIdx.x];
                if (hcode != EP_NONE) {
                    const int pidx = hidx + esdf_code_off(hcode);
                    const int px = pidx / ES_SX, py = (pidx - px * ES_SX) / ES_SY, pz = pidx - px * ES_SX - py * ES_SY;
                    if (px >= 1 && px <= 16 && py >= 1 && py <= 16 && pz >= 1 && pz <= 16) {          // the parent is one of this brick's voxels
                        uint32_t nw, npar;
                        esdf_rederive(s_t, hidx, hcode, hw, c1, c2, c3, max_dist, nw, npar);
                        if (nw != hw) told |= 1 << ((hk >> 13) & 31u);
                    }
                }
                const uint32_t sb = hw & 0x80000000u;
                const int base = (int)(hn & 0x1fffu), uc = (int)((hn >> 13) & 3u), vc = (int)((hn >> 15) & 3u), cu = (int)((hn >> 17) & 31u), cv = (int)((hn >> 22) & 31u), nf = (int)(hn >> 27);
                const int su = uc == 1 ? ES_SX : (uc == 2 ? ES_SY : (uc == 3 ? 1 : 0)), sv = vc == 2 ? ES_SY : (vc == 3 ? 1 : 0);
                uint32_t m[3] = { ~0u, ~0u, ~0u };
#pragma unroll
                for (int a = -1; a <= 1; ++a)
#pragma unroll
                    for (int b = -1; b <= 1; ++b) {
                        const bool ok = (a == 0 || (uc != 0 && cu + a >= 1 && cu + a <= 16)) && (b == 0 || (vc != 0 && cv + b >= 1 && cv + b <= 16));
                        const uint32_t x = ES_LD(&s_t[base + (ok ? a * su + b * sv : 0)]) ^ sb;
                        if (ok) m[(a != 0) + (b != 0)] = min(m[(a != 0) + (b != 0)], x);
                    }
                const float cc[5] = { 0.0f, c1, c2, c3, __uint_as_float(ES_INF) };
                float best = __uint_as_float(ES_INF);
#pragma unroll
                for (int e = 0; e < 3; ++e) best = fminf(best, __uint_as_float(min(m[e], ES_INF)) + cc[min(nf + e, 4)]);
                if (__float_as_uint(best) < (hw & 0x7fffffffu)) told |= 1 << ((hk >> 13) & 31u);
            }
            told &= ~(1 << 13);
            for (int d = 32; d >= 1; d >>= 1) told |= __shfl_xor(told, d);
            if ((threadIdx.x & 63u) == 0 && told) atomicOr(&s_notify, told);
        }
        __syncthreads();
        ESDF_TICKF(3);
        {
            const long long lw = wave_sum_ll((long long)(lowered + raised));
            if (lane_id() == 0 && lw) __hip_atomic_fetch_add(ES_STAT(E, 1), (int)lw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long rw = wave_sum_ll((long long)raised);
            if (lane_id() == 0 && rw) __hip_atomic_fetch_add(ES_STAT(E, 7), (int)rw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long ow = wave_sum_ll((long long)orphans);
            if (lane_id() == 0 && ow) __hip_atomic_fetch_add(&E.ctr[9], (int)ow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x < 64) {
            const int q = (int)threadIdx.x;
            bool put = false; int np = -1;
            if (q < 27 && q != 13 && (np = s_nb[q]) >= 0) {
                const bool told = (s_notify >> q) & 1, layer = (s_layer >> q) & 1;
                // a neighbour visited in THIS round may have staged values this visit has changed since: it looks again (and so does this brick, by the
                // neighbour's same rule, if the neighbour's layer next to it changed)
                // (... or is on the next round's list already -- through another neighbour, whose note bit says nothing about THIS side: without the bit
                //  for this side the entering sweeps from here would not run)
                const int st = (layer && !told) ? __hip_atomic_load(&E.stamp[np], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
                if (told || st == round || st == round + 1) {
                    const int seen = __hip_atomic_exchange(&E.stamp[np], round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_or(E.note + (size_t)((round + 1) & 1) * E.cap + np, 1u << (26 - q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    put = seen != round + 1;
                }
            }
            const int at = wave_reserve(&E.ctr[2 + nxt], put);
            if (put) next[at] = np;
        }
        if (threadIdx.x == 0) {
            if (E.region[p] != 2) __hip_atomic_fetch_add(ES_STAT(E, 4), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // bricks the wave reached
            E.region[p] = 2;
            __hip_atomic_fetch_add(ES_STAT(E, 0), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(ES_STAT(E, 2), sets, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(ES_STAT(E, 3), sets, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (rsets) { __hip_atomic_fetch_add(ES_STAT(E, 6), rsets, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_max(ES_STAT(E, 8), rsets, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        __syncthreads();
        ESDF_TICKF(4);
#ifdef TSL_TIMING      // histogram of whole visits by duration (bins of 4 us)
        if (threadIdx.x == 0) { const long long d = wall_clock64() - _t0; int pb = (int)(d / 400); pb = pb < 31 ? pb : 31;
            atomicAdd(&E.tm[8 + pb], (unsigned long long)d); atomicAdd(&E.tm[40 + pb], 1ull); atomicMax(&E.tm[72], (unsigned long long)d); }
        { const long long pw = wave_sum_ll((long long)(lowered + raised)); if (lane_id() == 0) { const long long d = wall_clock64() - _t0; int pb = (int)(d / 400); pb = pb < 31 ? pb : 31; atomicAdd(&E.tm[80 + pb], (unsigned long long)pw); } }
#endif
    }
}

__global__ void __launch_bounds__(256) k_esdf_export(MapDev M, int s, int nused, const float* esdf, float gamma, float max_dist, int16_t* idx, float* out, long long cap, int* counter)
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
                const float e = esdf[v];                                   // (a NaN: not observed yet when the ESDF was last updated)
                out[o] = fabsf(t) < gamma ? t : (float)sgn_f(t) * (e != e ? max_dist : fabsf(e));
            }
        }
    }
}

// cvt_ESDF_to_voxels_slice  dense_esdf.py:498-509: every observed voxel of the active submap with _index - 0.5 < k < _index + 0.5
// (k counted from the bottom of the volume, as the legacy module does) -> export_ESDF / export_ESDF_xyz, num_export_ESDF_particles
struct PoseE { float R[9], T[3]; };
__global__ void __launch_bounds__(256) k_esdf_slice(MapDev M, int s, int nused, const float* esdf, float gamma, float max_dist, PoseE B, int is_global, float vs, float index_f,
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
                const float e = esdf[v];
                val[o] = fabsf(t) < gamma ? t : (float)sgn_f(t) * (e != e ? max_dist : fabsf(e));  // :507
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
        long long sum[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int k = 0; k < ES_STAT_SLOTS; ++k) for (int c = 0; c < 9; ++c) { const int v = h[ES_CTR + k * 16 + c]; if (c == 3 || c == 8) sum[c] = v > sum[c] ? v : sum[c]; else sum[c] += v; }
        st.dirty_bricks = h[0]; st.changed_bricks = (int)sum[5]; st.region_bricks = (int)sum[4]; st.brick_relaxations = sum[0]; st.voxel_pushes = sum[1];
        st.rounds = h[7]; st.raise_sets = (int)sum[6]; st.voxels_raised = sum[7]; st.max_raise_sets = (int)sum[8]; st.passes = sum[2]; st.max_passes = (int)sum[3]; st.total_bricks = h[10] < m->M.max_bricks ? h[10] : m->M.max_bricks;
        if (h[2 + S.rounds % 3] != 0 || h[8] != 0) m->esdf_short = true;      // the last launched round still had work (or a raise did not settle)
        m->esdf_orphans += h[9];
        if (st.incremental && st.rounds > m->esdf_rounds_seen) m->esdf_rounds_seen = st.rounds;
        m->esdf_stats = st;
        m->esdf_tot.updates += 1; m->esdf_tot.incremental += st.incremental; m->esdf_tot.dirty_bricks += st.dirty_bricks; m->esdf_tot.region_bricks += st.region_bricks;
        m->esdf_tot.brick_relaxations += st.brick_relaxations; m->esdf_tot.voxel_pushes += st.voxel_pushes; m->esdf_tot.passes += st.passes;
#ifdef TSL_TIMING
        { const unsigned long long* tm = (const unsigned long long*)&h[16]; const double n = st.brick_relaxations ? st.brick_relaxations : 1;
          std::fprintf(stderr, "esdf timing: us per relaxation: setup %.2f stage %.2f [raise %.2f] relax %.2f [parents %.2f writeback %.2f] notify-test %.2f list %.2f (%lld relaxations)\n",
                       tm[0] / n / 100.0, tm[1] / n / 100.0, tm[5] / n / 100.0, tm[2] / n / 100.0, tm[6] / n / 100.0, tm[7] / n / 100.0, tm[3] / n / 100.0, tm[4] / n / 100.0, (long long)st.brick_relaxations);
          if (tm[73]) { const double f = (double)tm[73]; std::fprintf(stderr, "esdf timing, FIRST visits (%llu): [raise %.2f] relax %.2f [parents %.2f writeback %.2f] notify-test %.2f list %.2f\n", tm[73],
                       tm[77] / f / 100.0, tm[74] / f / 100.0, tm[78] / f / 100.0, tm[79] / f / 100.0, tm[75] / f / 100.0, tm[76] / f / 100.0); }
          std::fprintf(stderr, "esdf bricks per round:"); for (int k = 0; k < 16 && h[240 + k]; ++k) std::fprintf(stderr, " %d", h[240 + k]); std::fprintf(stderr, "\n");
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
        if ((rc = dev_alloc(m, (void**)&m->esdf_nbr, sizeof(int) * 27 * (size_t)nb, 0xff))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->esdf_par, (size_t)nb * TSL_BRK3, 0x15))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->esdf_ok, (size_t)nb, 0))) return rc;
        if ((rc = dev_alloc(m, (void**)&m->esdf_ctr, sizeof(int) * 2 * (ES_CTR + ES_STAT_SLOTS * 16), 0))) return rc;
        for (int i = 0; i < TSL_ESDF_SLOTS; ++i) {
            TSL_HIP(hipEventCreateWithFlags(&m->esdf_slot[i].ev, hipEventDisableTiming));
            TSL_HIP(hipHostMalloc((void**)&m->esdf_slot[i].host, sizeof(int) * (ES_CTR + ES_STAT_SLOTS * 16), hipHostMallocDefault));
        }
        TSL_HIP(hipEventCreate(&m->esdf_gate)); TSL_HIP(hipEventCreate(&m->esdf_read));       // (attached to a dispatch below: plain events)
        TSL_HIP(hipEventCreateWithFlags(&m->esdf_in, hipEventDisableTiming));
        m->esdf_valid = false;
    }
    esdf_retire(m, false);
    if (m->esdf_npend == TSL_ESDF_SLOTS) { EsdfSlot& S = m->esdf_slot[m->esdf_tail]; (void)hipEventSynchronize(S.ev); esdf_retire(m, false); }
    const hipStream_t q0 = ms(m);                            // issues the queued frames first
    // The update runs beside the handle's stream: it starts when everything queued there so far has finished (esdf_in), and the handle's
    // stream goes on once the update has READ the TSDF (collect, dilate, init: esdf_read) -- the relaxation rounds only touch the ESDF's
    // own arrays, so they run beside the integration of the next frames.  The stream is the phase-A stream of the batch slot that comes
    // into use last (the runtime maps streams onto four hardware queues: a stream of its own shared a queue with one of the phase-A
    // streams and held every third frame back; a high-priority stream completed its kernels in ~57 us quanta); consecutive updates are
    // on different streams and follow each other through the previous update's event.
    const hipStream_t q = (m->esdf_overlap && m->overlap != 0) ? m->batch[(m->cur + 1) % TSL_NBATCH].st : q0;
    if (q != q0) {
        TSL_HIP(hipEventRecord(m->esdf_in, q0)); TSL_HIP(hipStreamWaitEvent(q, m->esdf_in, 0));
        if (m->esdf_last) TSL_HIP(hipStreamWaitEvent(q, m->esdf_last, 0));
    }
    const int s = m->cfg.is_global_map ? 0 : m->active;
    // a changed voxel influences voxels up to max_dist away: that many voxels = `reach` bricks in every direction
    int reach = (int)std::ceil((double)max_dist / ((double)m->P.vs * 16.0)); if (reach < 1) reach = 1;
    const bool full = force_full || m->esdf_force_full || !m->esdf_valid || m->esdf_submap != s || m->esdf_gamma != gamma || m->esdf_maxd != max_dist ||
                      2 * reach + 1 >= m->nbx;              // the dilation would cover the grid anyway
    int* const ctr = m->esdf_ctr + (size_t)m->esdf_ctr_idx * (ES_CTR + ES_STAT_SLOTS * 16);          // this update's counters (zero: allocation / the update before)
    int* const ctr_next = m->esdf_ctr + (size_t)(1 - m->esdf_ctr_idx) * (ES_CTR + ES_STAT_SLOTS * 16);
    m->esdf_ctr_idx = 1 - m->esdf_ctr_idx;
    EsdfDev E = { m->esdf, m->esdf_fl, m->esdf_region, m->esdf_inq, m->esdf_list, m->esdf_note, m->esdf_queue, m->esdf_nbr, m->esdf_par, m->esdf_ok, nb, (unsigned long long*)(ctr + 16), ctr_next, ctr };
    const bool wavefront = m->esdf_mode != 0;
    EsdfSlot& S = m->esdf_slot[(m->esdf_tail + m->esdf_npend) % TSL_ESDF_SLOTS];
    std::memset(&S.st, 0, sizeof(S.st));
    S.st.incremental = full ? 0 : 1;
    prof_begin(m, TSL_K_ESDF, q);                                // one event pair around the update's launches (collect .. last round)
    m->prof_group = true;
    const int nbk = (nb + 255) / 256;
    // phase A of frames queued from now on allocates bricks (pool counter, table entry, owner -- in that order): it starts after the
    // snapshot + collect, so that every pool index below the snapshot has its owner written (launch_batch_t waits for the gate).  The gate
    // is the collect kernel's own completion (an event recorded behind it costs a marker packet and ~6 us before the next kernel starts)
    // Beside the next frame (q != q0) ONE event serves both purposes, the completion of the init kernel: phase A has slack there -- the update is
    // the longer chain -- and every event attached to a dispatch costs the update ~5 us before its next kernel starts.
    if (q == q0) { hipExtLaunchKernelGGL(k_esdf_collect, dim3(nbk), dim3(256), 0, q, nullptr, m->esdf_gate, 0, m->M, E, s, full ? 1 : 0, wavefront ? 1 : 0); m->esdf_gate_ev = m->esdf_gate; }
    else { hipLaunchKernelGGL(k_esdf_collect, dim3(nbk), dim3(256), 0, q, m->M, E, s, full ? 1 : 0, wavefront ? 1 : 0); m->esdf_gate_ev = m->esdf_read; }
    m->esdf_gate_set = true; m->esdf_gate_mask = 0;
    if (wavefront) {
        // raise / lower wavefront: the bricks whose ESDF inputs changed start the wave, nothing is dilated or re-initialised
        if (q != q0) {
            hipExtLaunchKernelGGL(k_esdf_diff, dim3(1024), dim3(256), 0, q, nullptr, m->esdf_read, 0, m->M, E, s, full ? 1 : 0, gamma, max_dist);
            TSL_HIP(hipStreamWaitEvent(q0, m->esdf_read, 0));
        } else hipLaunchKernelGGL(k_esdf_diff, dim3(1024), dim3(256), 0, q, m->M, E, s, full ? 1 : 0, gamma, max_dist);
    } else {
    hipLaunchKernelGGL(k_esdf_dilate, dim3(1024), dim3(256), 0, q, m->M, E, s, full ? 0 : reach, full ? 1 : 0, gamma, max_dist);
    if (q != q0) {
        hipExtLaunchKernelGGL(k_esdf_init, dim3(2048), dim3(256), 0, q, nullptr, m->esdf_read, 0, m->M, E, s, gamma, max_dist);
        TSL_HIP(hipStreamWaitEvent(q0, m->esdf_read, 0));
    } else hipLaunchKernelGGL(k_esdf_init, dim3(2048), dim3(256), 0, q, m->M, E, s, gamma, max_dist);
    }
    // information crosses one brick per round: `reach` rounds carry a value as far as it can matter, bends and late improvements add a
    // few more (8 rounds had work at reach = 4 on the benchmark stream).
    // The batch is launched blind: 2 * reach + 8 rounds to begin with and for full recomputes, afterwards two more than the most any
    // completed update of this handle needed (a round without work costs ~5 us; stopping early costs a full recompute, see esdf_finish).
    int rounds = ((full || m->esdf_rounds_seen == 0) ? 2 * reach + 8 : std::min(2 * reach + 8, std::max(reach + 2, m->esdf_rounds_seen + 2))) + extra_rounds;
    const int grid = m->esdf_grid > 0 ? m->esdf_grid : 4 * m->ncu;
    if (m->esdf_round_cap > 0 && extra_rounds == 0 && rounds > m->esdf_round_cap) rounds = m->esdf_round_cap;      // test knob: provoke the repair path
    for (int k = 0; k < rounds; ++k) {
        if (wavefront) hipLaunchKernelGGL(k_esdf_wave, dim3(grid), dim3(384), 0, q, m->M, E, s, m->P.vs, max_dist, k);
        else hipLaunchKernelGGL(k_esdf_round, dim3(grid), dim3(384), 0, q, m->M, E, s, m->P.vs, max_dist, k);
    }
    m->prof_group = false; prof_end(m, q);
    TSL_HIP(hipMemcpyAsync(S.host, ctr, sizeof(int) * (ES_CTR + ES_STAT_SLOTS * 16), hipMemcpyDeviceToHost, q));
    TSL_HIP(hipEventRecord(S.ev, q)); m->esdf_last = S.ev;
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
    if (m->esdf_in) { (void)hipEventDestroy(m->esdf_in); m->esdf_in = nullptr; }
    if (m->esdf_read) { (void)hipEventDestroy(m->esdf_read); m->esdf_read = nullptr; }
    m->esdf_last = nullptr;
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
    if (nused > 0) hipLaunchKernelGGL(k_esdf_export, dim3(nused < 8192 ? nused : 8192), dim3(256), 0, ms(m), m->M, s, nused, m->esdf, m->esdf_gamma, m->esdf_maxd, *didx, *dval, (long long)cap, counter);
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
    if (nused > 0) hipLaunchKernelGGL(k_esdf_slice, dim3(nused < 8192 ? nused : 8192), dim3(256), 0, ms(m), m->M, s, nused, m->esdf, m->esdf_gamma, m->esdf_maxd, B, m->cfg.is_global_map, m->P.vs,
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
