// tools/ubench/fetch_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for access patterns with a KNOWN byte count.
// The guide (MI355X_MICROARCH.md, HBM) calibrates one pattern -- a wide coalesced streaming read reports half its bytes -- and calls every
// other width uncalibrated.  The kernels of this package that are not streaming reads are narrow gathers (marching cubes' 19^3 tile: 4-byte
// and 1-byte loads, rows of 16 + 3 entries from three bricks; the ESDF halo; the Octomap leaves), so the factor has to be measured on those.
// Every kernel below touches each address ONCE in a 1 GiB buffer (four times the Infinity Cache) and says how many bytes it asked for and how
// many 64-byte / 128-byte lines that touches.  Run under  rocprofv3 --pmc FETCH_SIZE  and  --pmc WRITE_SIZE  (tools/gpu_fetch_calib.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// coalesced streaming read, 16 B per lane (the guide's calibrated pattern)
__global__ void __launch_bounds__(256) read_stream16(const uint4* __restrict__ p, size_t n16, uint32_t* out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) *out = acc;
}
// coalesced streaming read, 4 B per lane
__global__ void __launch_bounds__(256) read_stream4(const uint32_t* __restrict__ p, size_t n4, uint32_t* out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc ^= p[i];
    if (acc == 0x12345u) *out = acc;
}
// ONE load of BYTES (1 / 4) per lane every STRIDE bytes: a gather that uses a sliver of every line it touches
template <int BYTES>
__global__ void __launch_bounds__(256) read_strided(const uint8_t* __restrict__ p, size_t nloads, size_t stride, uint32_t* out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nloads; i += (size_t)gridDim.x * 256) {
        if (BYTES == 4) acc ^= *reinterpret_cast<const uint32_t*>(p + i * stride); else acc ^= p[i * stride];
    }
    if (acc == (BYTES == 4 ? 0x12345u : 0x45u)) *out = acc;            // (a value the XOR of the loads can take: the loop must not be provably dead)
}
// the marching-cubes tile row: 19 consecutive 4-byte entries of which 16 are one 64-byte row of a brick and 1 + 2 come from the rows of two
// other bricks (here: other 64-byte rows far away), one row per 19 lanes
__global__ void __launch_bounds__(256) read_rows19(const uint32_t* __restrict__ p, size_t nrows, size_t rows_total, uint32_t* out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nrows * 19; i += (size_t)gridDim.x * 256) {
        const size_t r = i / 19; const int e = (int)(i % 19);
        size_t word;
        if (e >= 1 && e <= 16) word = r * 16 + (size_t)(e - 1);                               // the brick's own row
        else if (e == 0) word = ((r + rows_total / 3) % rows_total) * 16 + 15;                // last entry of a row of the brick below
        else word = ((r + 2 * (rows_total / 3)) % rows_total) * 16 + (size_t)(e - 17);        // first two entries of a row of the brick above
        acc ^= p[word];
    }
    if (acc == 0x12345u) *out = acc;
}
// coalesced streaming write, 16 B per lane; and one 4-byte store every STRIDE bytes
__global__ void __launch_bounds__(256) write_stream16(uint4* __restrict__ p, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void __launch_bounds__(256) write_strided4(uint8_t* __restrict__ p, size_t nstores, size_t stride)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nstores; i += (size_t)gridDim.x * 256) *reinterpret_cast<uint32_t*>(p + i * stride) = (uint32_t)i;
}

int main()
{
    const size_t N = (size_t)1 << 30;
    uint8_t* buf = nullptr; uint32_t* out = nullptr;
    CK(hipMalloc(&buf, N)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, N)); CK(hipMemset(out, 0, 64));
    CK(hipDeviceSynchronize());
    const dim3 g(8192), b(256);
    // name, bytes asked for, 64-byte lines touched, 128-byte lines touched (printed; the counters come from rocprofv3)
    printf("kernel,requested_bytes,lines64,lines128\n");
    hipLaunchKernelGGL(read_stream16, g, b, 0, 0, (const uint4*)buf, N / 16, out);                     printf("read_stream16,%zu,%zu,%zu\n", N, N / 64, N / 128);
    hipLaunchKernelGGL(read_stream4, g, b, 0, 0, (const uint32_t*)buf, N / 4, out);                    printf("read_stream4,%zu,%zu,%zu\n", N, N / 64, N / 128);
    hipLaunchKernelGGL(read_strided<4>, g, b, 0, 0, (const uint8_t*)buf, N / 64, (size_t)64, out);     printf("read_strided<4>/64,%zu,%zu,%zu\n", N / 64 * 4, N / 64, N / 128);
    hipLaunchKernelGGL(read_strided<4>, g, b, 0, 0, (const uint8_t*)buf, N / 128, (size_t)128, out);   printf("read_strided<4>/128,%zu,%zu,%zu\n", N / 128 * 4, N / 128, N / 128);
    hipLaunchKernelGGL(read_strided<4>, g, b, 0, 0, (const uint8_t*)buf, N / 256, (size_t)256, out);   printf("read_strided<4>/256,%zu,%zu,%zu\n", N / 256 * 4, N / 256, N / 256);
    hipLaunchKernelGGL(read_strided<1>, g, b, 0, 0, (const uint8_t*)buf, N / 64, (size_t)64, out);     printf("read_strided<1>/64,%zu,%zu,%zu\n", N / 64, N / 64, N / 128);
    hipLaunchKernelGGL(read_strided<1>, g, b, 0, 0, (const uint8_t*)buf, N / 16, (size_t)16, out);     printf("read_strided<1>/16,%zu,%zu,%zu\n", N / 16, N / 64, N / 128);
    { const size_t rows = N / 64; hipLaunchKernelGGL(read_rows19, g, b, 0, 0, (const uint32_t*)buf, rows / 4, rows, out);
      printf("read_rows19,%zu,%zu,%zu\n", rows / 4 * 19 * 4, rows / 4 * 3, rows / 4 * 3 / 2); }        // (rows of consecutive bricks' rows pair up in 128-byte lines)
    hipLaunchKernelGGL(write_stream16, g, b, 0, 0, (uint4*)buf, N / 16);                               printf("write_stream16,%zu,%zu,%zu\n", N, N / 64, N / 128);
    hipLaunchKernelGGL(write_strided4, g, b, 0, 0, buf, N / 64, (size_t)64);                           printf("write_strided4/64,%zu,%zu,%zu\n", N / 64 * 4, N / 64, N / 128);
    hipLaunchKernelGGL(write_strided4, g, b, 0, 0, buf, N / 128, (size_t)128);                         printf("write_strided4/128,%zu,%zu,%zu\n", N / 128 * 4, N / 128, N / 128);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    return 0;
}
