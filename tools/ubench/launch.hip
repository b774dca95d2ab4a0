// Developer microbenchmark: sustained kernel-launch throughput of dependent chains on 1..8 streams (MI355X).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_work(int* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) atomicAdd(&p[(i * 2654435761u) & 0xffff], 1); }
int main()
{
    int* d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    for (int blocks : {1, 300}) for (int ns : {1, 2, 4, 8}) {
        std::vector<hipStream_t> st(ns);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        const int per = 2000;
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < per; ++i) for (int s = 0; s < ns; ++s) {
                if (blocks == 1) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st[s], d + 1024 * s);
                else hipLaunchKernelGGL(k_work, dim3(blocks), dim3(256), 0, st[s], d + 65536 * (s % 4), blocks * 256);
            }
            auto t1 = std::chrono::steady_clock::now();
            hipDeviceSynchronize();
            auto t2 = std::chrono::steady_clock::now();
            if (rep) printf("blocks %4d streams %d: host %.2f us/launch, gpu %.2f us/launch aggregate (%.2f us per launch per stream)\n", blocks, ns,
                            std::chrono::duration<double, std::micro>(t1 - t0).count() / (per * ns),
                            std::chrono::duration<double, std::micro>(t2 - t0).count() / (per * ns),
                            std::chrono::duration<double, std::micro>(t2 - t0).count() / per);
        }
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}
