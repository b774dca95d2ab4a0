// tsl_common.hpp -- shared host/device helpers of the MI355X dense-mapping backend (gfx950 only).
//
// Numeric contract (DESIGN.md "Defined semantics"): all f32 arithmetic is IEEE, no FMA contraction
// (-ffp-contract=off), correctly rounded divide/sqrt; f16 values are stored as raw bits and every
// f16 (op) f16 is computed in f32 and rounded RNE (v_cvt_f16_f32).  Per-frame voxel contributions are
// summed in exact 2^-24 fixed point (int64 atomics), so results do not depend on thread order.
#pragma once
#include <cstring>
#include <string.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/taichislam_hip.h"

#define TSL_BRK   16
#define TSL_BRK3  4096
#define TSL_EMPTY  (-1)
#define TSL_LOCKED (-2)
#define TSL_FULL   (-3)

namespace tsl {

void set_error(const std::string& s);
#define TSL_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    tsl::set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); return TSL_ERR_HIP; } } while (0)
#define TSL_REQUIRE(cond, msg) do { if (!(cond)) { tsl::set_error(msg); return TSL_ERR_ARG; } } while (0)

// ---- f16 as raw bits -----------------------------------------------------------------------------
typedef uint16_t h16;
__device__ __forceinline__ h16 f2h(float x) { _Float16 v = (_Float16)x; return __builtin_bit_cast(h16, v); }
__device__ __forceinline__ float h2f(h16 b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ h16 hadd(h16 a, h16 b) { return f2h(h2f(a) + h2f(b)); }
__device__ __forceinline__ h16 hsub(h16 a, h16 b) { return f2h(h2f(a) - h2f(b)); }
__device__ __forceinline__ h16 hmul(h16 a, h16 b) { return f2h(h2f(a) * h2f(b)); }
__device__ __forceinline__ h16 hdiv(h16 a, h16 b) { return f2h(h2f(a) / h2f(b)); }
// sqrtf() is the correctly rounded form under -fhip-fp32-correctly-rounded-divide-sqrt; __fsqrt_rn lowers to a bare
// v_sqrt_f32 (1 ulp) on gfx950 and breaks bit parity with the CPU oracle.
__device__ __forceinline__ float sqrt_rn(float x) { return sqrtf(x); }
// the same correctly rounded root for x >= 2^-96 (or 0) without the library's rescaling of tiny inputs: hardware estimate,
// then one step down / one step up decided by the exact residuals (tsl_selftest(1) compares it with sqrtf for every float)
__device__ __forceinline__ float sqrt_rn_norm(float x)
{
    const float y = __builtin_amdgcn_sqrtf(x);
    const float yd = __uint_as_float(__float_as_uint(y) - 1u), yu = __uint_as_float(__float_as_uint(y) + 1u);
    const float ed = __builtin_fmaf(-yd, y, x), eu = __builtin_fmaf(-yu, y, x);
    float r = (ed <= 0.0f) ? yd : y;
    r = (eu > 0.0f) ? yu : r;
    return r;
}
__device__ __forceinline__ h16 hsqrt(h16 a) { return f2h(sqrt_rn(h2f(a))); }

// ti.round(x, ti.i32): round half away from zero (mapping_common.py:263-266, assumption A1)
__device__ __forceinline__ float rnd_f(float x)
{
    float r = truncf(x);
    float d = fabsf(x - r);
    if (d >= 0.5f) r += copysignf(1.0f, x);
    return r;
}
// the same integer in three instructions: adding the float just below one half (with the sign of x) never carries a value
// from below a tie over it, and carries every value from the tie on; the conversion truncates.  tsl_selftest(0) checks the
// identity against rnd_f for every float.
__device__ __forceinline__ int rnd_i(float x) { return (int)(x + copysignf(0.49999997f, x)); }
__device__ __forceinline__ int rnd_i_ref(float x) { return (int)rnd_f(x); }
__device__ __forceinline__ int sgn_f(float v) { return (0.0f < v) - (v < 0.0f); }     // mapping_common.py:5-7

// x / vs with a loop-invariant divisor.  IEEE division costs ~10 VALU ops on gfx950; with y = RN(1/vs) the sequence
// q0 = x*y, r = fma(-q0, vs, x), q = fma(r, y, q0) is the correctly rounded quotient (Markstein) -- and because the parity
// contract is bit-exactness this is not assumed: tsl_tsdf_create checks it on the device against IEEE division for every
// float x in the range that can reach an integer conversion and only then sets `fast`.
__device__ __forceinline__ float div_vs(float x, float vs, float rvs, int fast)
{
    if (fast) { const float q0 = x * rvs; const float r = __builtin_fmaf(-q0, vs, x); return __builtin_fmaf(r, rvs, q0); }
    return x / vs;
}

// ---- 2^-24 fixed point ------------------------------------------------------------------------------
#define TSL_FIX_SCALE 16777216.0f
#define TSL_FIX_INV   (1.0 / 16777216.0)
#define TSL_W_CLAMP   65536.0f
#define TSL_WMAX      1000.0f                                                             // dense_tsdf.py:8
__device__ __forceinline__ long long to_fix(float v)
{
    const float q = rintf(v * TSL_FIX_SCALE);                  // integer-valued
    if (fabsf(q) < 2147483648.0f) return (long long)(int)q;    // common case: one v_cvt_i32_f32 + sign extension
    return __float2ll_rn(q);
}
// inside hot loops: the 64-bit conversion is only executed when some lane of the wave needs it
__device__ __forceinline__ long long to_fix_wave(float v)
{
    const float q = rintf(v * TSL_FIX_SCALE);
    if (__builtin_expect(__any(!(fabsf(q) < 2147483648.0f)), 0)) return __float2ll_rn(q);
    return (long long)(int)q;
}
__device__ __forceinline__ float from_fix(long long q) { return (float)((double)q * TSL_FIX_INV); }

// ---- wave64 helpers -----------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }
// number of set bits of m strictly below this lane
__device__ __forceinline__ int rank_below(unsigned long long m)
{ return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }

// one atomic per wave: adds `v` of every lane with pred, returns nothing
__device__ __forceinline__ void atomic_add_i64(int64_t* ctr, long long v)
{ __hip_atomic_fetch_add(reinterpret_cast<long long*>(ctr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wave_count_add(int64_t* ctr, bool pred)
{
    unsigned long long m = __ballot(pred);
    if (m && lane_id() == (int)__builtin_ctzll(m)) atomic_add_i64(ctr, (long long)popc64(m));
}
// block-wide count of `pred` with ONE global atomic per block (same-address global atomics cost ~12 ns each on
// MI355X and serialise; tools/ubench/atomics.hip).  Must be reached by every thread of the block.
__device__ __forceinline__ void block_count_add(int64_t* ctr, bool pred)
{
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned long long m = __ballot(pred);
    if (m && lane_id() == (int)__builtin_ctzll(m)) atomicAdd(&s_cnt, popc64(m));
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomic_add_i64(ctr, (long long)s_cnt);
    __syncthreads();
}
__device__ __forceinline__ long long wave_sum_ll(long long v)
{
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
// wave-aggregated slot reservation: lanes with pred get consecutive indices from *ctr
__device__ __forceinline__ int wave_reserve(int* ctr, bool pred)
{
    unsigned long long m = __ballot(pred);
    if (!m) return -1;
    int leader = (int)__builtin_ctzll(m);
    int base = 0;
    if (lane_id() == leader) base = __hip_atomic_fetch_add(ctr, popc64(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = __shfl(base, leader);
    return pred ? base + rank_below(m) : -1;
}

// block-aggregated reservation of n items per thread (256-thread blocks): returns the first index of this thread's range
__device__ __forceinline__ int block_reserve_n(int* ctr, int n)
{
    __shared__ int s_w[4];
    __shared__ int s_b;
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    int inc = n;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) s_w[wid] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        s_b = tot ? __hip_atomic_fetch_add(ctr, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    }
    __syncthreads();
    int off = s_b + inc - n;
    for (int w = 0; w < wid; ++w) off += s_w[w];
    __syncthreads();
    return off;
}

// block-aggregated reservation (256-thread blocks): ONE global atomic per block; must be reached by every thread
__device__ __forceinline__ int block_reserve(int* ctr, bool pred)
{
    __shared__ int s_w[4];
    __shared__ int s_b;
    const unsigned long long m = __ballot(pred);
    const int wid = threadIdx.x >> 6;
    if (lane_id() == 0) s_w[wid] = popc64(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        s_b = tot ? __hip_atomic_fetch_add(ctr, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    }
    __syncthreads();
    int off = s_b + rank_below(m);
    for (int w = 0; w < wid; ++w) off += s_w[w];
    __syncthreads();
    return pred ? off : -1;
}

// Claim-or-read an index stored in *entry (EMPTY -> allocate from *counter).  Safe inside divergent
// SIMT code: a lane never waits on a lane of its own wave (winners publish in the same iteration).
__device__ __forceinline__ int claim_index_1(int* entry, int* counter, int cap)
{
    int v = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (v == TSL_EMPTY || v == TSL_LOCKED) {
        if (v == TSL_EMPTY) {
            int old = atomicCAS(entry, TSL_EMPTY, TSL_LOCKED);
            if (old == TSL_EMPTY) {
                // out of capacity: the entry goes back to EMPTY (a later reset / larger pool can allocate it) and the caller
                // reports the failure; TSL_FULL is only ever a return value, never a table entry.  The counter is looked at first:
                // once the pool is full it is not bumped any more (every later access of an absent brick repeats this claim -- the
                // counter stays within `cap` + the claims that were in flight when it filled up, and is not an atomic hot spot)
                int idx = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (idx < cap) idx = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(entry, idx >= cap ? TSL_EMPTY : idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return idx >= cap ? TSL_FULL : idx;
            }
            v = old;
        } else {
            __builtin_amdgcn_s_sleep(2);
            v = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    return v;
}
// Wave-cooperative form: the lanes that reach this point together usually want the SAME entry (coherent rays
// enter a new brick in the same step).  64 same-address CAS cost 64 x ~12 ns on MI355X, so one lane per
// distinct entry does the claim and the result is broadcast.  Callable from divergent code: only the
// currently active lanes take part.
__device__ __forceinline__ int claim_index(int* entry, int* counter, int cap)
{
    int v = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long todo = __ballot(v == TSL_EMPTY || v == TSL_LOCKED);
    while (todo) {
        const int leader = (int)__builtin_ctzll(todo);
        const unsigned long long lead = __shfl((unsigned long long)entry, leader);
        int r = 0;
        if (lane_id() == leader) r = claim_index_1(entry, counter, cap);
        r = __shfl(r, leader);
        const bool mine = ((unsigned long long)entry == lead);
        if (mine) v = r;
        todo &= ~__ballot(mine);
    }
    return v;
}

// ---- device-side views (PODs passed by value to kernels) -------------------------------------------------
struct MapDev {
    int N, Nz, nbx, nbz, nb3, nsub;
    int hN, hNz;               // N/2, Nz/2
    int max_bricks;
    int* table;                // [nsub][nb3] -> pool brick index, TSL_EMPTY when absent
    uint32_t* tw;              // [max_bricks][4096]  lo16 = TSDF f16 bits, hi16 = W_TSDF f16 bits
    int8_t* obs;               // [max_bricks][4096]
    int8_t* occ;               // [max_bricks][4096]
    uint16_t* col;             // [max_bricks][4096][4] f16 rgb (+pad) or nullptr
    int* owner;                // [max_bricks] -> s*nb3 + b
    uint8_t* touch;            // [max_bricks] set by the integrate kernels when they write a brick's TSDF (consumed by the incremental ESDF)
    int* pool_top;             // bricks handed out so far
    int* err;                  // sticky device error flags (bit 0 brick pool full, 1 frame scratch, 2 ray segments, 3 crowded sensor voxel)
};

__device__ __forceinline__ bool in_volume(const MapDev& M, int i, int j, int k)
{
    return i >= -M.hN && i < M.N - M.hN && j >= -M.hN && j < M.N - M.hN && k >= -M.hNz && k < M.Nz - M.hNz;
}
// brick id inside a submap + voxel id inside the brick (k fastest)
__device__ __forceinline__ int brick_of(const MapDev& M, int i, int j, int k, int* local)
{
    int ui = i + M.hN, uj = j + M.hN, uk = k + M.hNz;
    *local = ((ui & 15) << 8) | ((uj & 15) << 4) | (uk & 15);
    return ((ui >> 4) * M.nbx + (uj >> 4)) * M.nbz + (uk >> 4);
}
__device__ __forceinline__ int pool_lookup(const MapDev& M, int s, int b)
{ return __hip_atomic_load(M.table + (size_t)s * M.nb3 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// read-only lookup through the normal (cached) path: for kernels that run after all allocation is done
__device__ __forceinline__ int pool_lookup_ro(const MapDev& M, int s, int b) { return M.table[(size_t)s * M.nb3 + b]; }
// COOP: wave-cooperative claim (callers are coherent rays that usually want the same brick); !COOP: every lane claims
// on its own (callers hold distinct bricks).
template <bool COOP = true>
__device__ __forceinline__ int pool_claim(const MapDev& M, int s, int b)
{
    int* e = M.table + (size_t)s * M.nb3 + b;
    int v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= 0) return v;
    v = COOP ? claim_index(e, M.pool_top, M.max_bricks) : claim_index_1(e, M.pool_top, M.max_bricks);
    if (v >= 0) M.owner[v] = s * M.nb3 + b;      // idempotent: every claimer of a fresh brick writes the same value
    else atomicOr(M.err, 1);
    return v;
}

// one voxel of a (reset) global map from the fusion's sums: {sum w*t, sum w} in 2^-24 fixed point, c = contributions * 65536 + occupancy sum
// (fuse_with_interploation dense_tsdf.py:272-280 applied once per voxel; used by tsl_fuse.hip and tsl_merge.hip)
__device__ __forceinline__ void fuse_write_voxel(const MapDev& G, size_t v, long long qn, long long qd, int c)
{
    const int occ_sum = (int)(int16_t)(c & 0xffff);
    const float num = from_fix(qn), den = from_fix(qd);
    G.tw[v] = (uint32_t)f2h(num / den) | ((uint32_t)f2h(den) << 16);          // empty global map: T0 = W0 = 0  (:275,:278)
    G.obs[v] = 1;
    G.occ[v] = (int8_t)occ_sum;                                                // i8 wrap as in the reference (:280, Q7)
}


}  // namespace tsl
