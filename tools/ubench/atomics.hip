// micro-benchmarks that price the design choices of the ray-integrate kernel on MI355X:
//   global int64 atomic throughput (scattered / contended), LDS int64 atomic throughput, plain scattered RMW.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// each thread does `per` no-return u64 atomic pairs at pseudo-random slots of a table with `slots` entries
__global__ void g_atomic_scatter(unsigned long long* t, uint32_t slots, int per, uint32_t seed)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = hash32(tid * 2654435761u + seed);
    for (int i = 0; i < per; ++i) {
        x = hash32(x + i);
        unsigned long long* p = t + (size_t)(x % slots) * 2;
        __hip_atomic_fetch_add(p, 12345ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(p + 1, 7ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// coherent: a wave's 64 lanes hit 64 consecutive slots (one 1 KiB region), waves scattered
__global__ void g_atomic_coherent(unsigned long long* t, uint32_t slots, int per, uint32_t seed)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t w = tid >> 6, l = tid & 63;
    uint32_t x = hash32(w * 2654435761u + seed);
    for (int i = 0; i < per; ++i) {
        x = hash32(x + i);
        unsigned long long* p = t + ((size_t)((x % (slots / 64)) * 64 + l)) * 2;
        __hip_atomic_fetch_add(p, 12345ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(p + 1, 7ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// contended: all threads hammer `hot` distinct slots
__global__ void g_atomic_hot(unsigned long long* t, uint32_t hot, int per, uint32_t seed)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = hash32((tid >> 6) * 2654435761u + seed);
    for (int i = 0; i < per; ++i) {
        x = hash32(x + i);
        unsigned long long* p = t + (size_t)(x % hot) * 2 * 64;     // one line per hot slot, wave-uniform address
        __hip_atomic_fetch_add(p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// 32-bit variant of the scatter
__global__ void g_atomic_scatter32(unsigned int* t, uint32_t slots, int per, uint32_t seed)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = hash32(tid * 2654435761u + seed);
    for (int i = 0; i < per; ++i) {
        x = hash32(x + i);
        unsigned int* p = t + (size_t)(x % slots) * 2;
        __hip_atomic_fetch_add(p, 12345u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(p + 1, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// LDS: 4096-voxel brick accumulators ({num,den} u64 pairs = 64 KiB), random or wave-contended slots, then flush
template <int MODE>
__global__ void __launch_bounds__(256) l_atomic(unsigned long long* out, int per, uint32_t seed)
{
    __shared__ unsigned long long acc[4096 * 2];
    for (int i = threadIdx.x; i < 8192; i += 256) acc[i] = 0;
    __syncthreads();
    uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t x = hash32((MODE == 1 ? (tid >> 6) : tid) * 2654435761u + seed);
    for (int i = 0; i < per; ++i) {
        x = hash32(x + i);
        uint32_t s = x & 4095u;
        atomicAdd(&acc[s * 2], 12345ull);
        atomicAdd(&acc[s * 2 + 1], 7ull);
    }
    __syncthreads();
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < 8192; i += 256) s += acc[i];
    if (s == 0xdeadbeefull) out[0] = s;
}
// plain (non-atomic) scattered 16-byte read-modify-write
__global__ void g_rmw_scatter(ulonglong2* t, uint32_t slots, int per, uint32_t seed)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = hash32(tid * 2654435761u + seed);
    for (int i = 0; i < per; ++i) {
        x = hash32(x + i);
        ulonglong2 v = t[x % slots]; v.x += 12345ull; v.y += 7ull; t[x % slots] = v;
    }
}
template <typename F> static double timeit(F f, int reps = 5)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best * 1e-3;
}
int main()
{
    const size_t bytes = 1ull << 30;
    unsigned long long* t; CK(hipMalloc(&t, bytes)); CK(hipMemset(t, 0, bytes));
    const int threads = 1 << 20, per = 8; const double ops = (double)threads * per;
    for (uint32_t slots : {1u << 16, 1u << 20, 1u << 22, 1u << 25}) {
        double s = timeit([&] { hipLaunchKernelGGL(g_atomic_scatter, dim3(threads / 256), dim3(256), 0, 0, t, slots, per, 1u); });
        printf("global u64 atomic pair, random over %8u slots (%6.1f MB): %7.2f G pairs/s (%.1f us for %.1fM pairs)\n", slots, slots * 16 / 1e6, ops / s / 1e9, s * 1e6, ops / 1e6);
        s = timeit([&] { hipLaunchKernelGGL(g_atomic_coherent, dim3(threads / 256), dim3(256), 0, 0, t, slots, per, 1u); });
        printf("global u64 atomic pair, wave-coherent  %8u slots            : %7.2f G pairs/s\n", slots, ops / s / 1e9);
        s = timeit([&] { hipLaunchKernelGGL(g_atomic_scatter32, dim3(threads / 256), dim3(256), 0, 0, (unsigned*)t, slots, per, 1u); });
        printf("global u32 atomic pair, random over %8u slots            : %7.2f G pairs/s\n", slots, ops / s / 1e9);
        s = timeit([&] { hipLaunchKernelGGL(g_rmw_scatter, dim3(threads / 256), dim3(256), 0, 0, (ulonglong2*)t, slots, per, 1u); });
        printf("plain 16B RMW,          random over %8u slots            : %7.2f G/s\n", slots, ops / s / 1e9);
    }
    for (uint32_t hot : {1u, 8u, 64u, 1024u}) {
        double s = timeit([&] { hipLaunchKernelGGL(g_atomic_hot, dim3(threads / 256), dim3(256), 0, 0, t, hot, per, 1u); });
        printf("global u64 atomic, wave-uniform address, %5u hot lines: %7.3f G lane-ops/s = %.3f G wave-atomics/s (%.1f ns per wave-atomic per line)\n", hot, ops / s / 1e9, ops / 64 / s / 1e9, s * 1e9 / (ops / 64 / hot));
    }
    for (int blocks : {256, 1024, 4096}) {
        const int p2 = 64; double o2 = (double)blocks * 256 * p2;
        double s = timeit([&] { hipLaunchKernelGGL(l_atomic<0>, dim3(blocks), dim3(256), 0, 0, t, p2, 1u); });
        printf("LDS u64 atomic pair, random slot, %5d blocks: %7.2f G pairs/s\n", blocks, o2 / s / 1e9);
        s = timeit([&] { hipLaunchKernelGGL(l_atomic<1>, dim3(blocks), dim3(256), 0, 0, t, p2, 1u); });
        printf("LDS u64 atomic pair, wave-same slot, %5d blocks: %7.2f G pairs/s\n", blocks, o2 / s / 1e9);
    }
    return 0;
}
