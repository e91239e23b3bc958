// Dependent-load latency as a short kernel sees it: one wave per workgroup chases a random cycle through a buffer.
//   usage: latency_probe [buffer MiB] [blocks] [hops] [idle microseconds between launches]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <thread>
#include <vector>

__global__ void chase(const unsigned* __restrict__ next, int hops, unsigned stride, unsigned long long* out, unsigned seed)
{
    // every lane its own chain, a different one in every launch
    unsigned at = (((blockIdx.x * 64u + threadIdx.x) * 2654435761u + seed * 40503u) % (stride * 32u)) * 32u;
    at = next[at];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < hops; ++i) at = next[at];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], t1 - t0 + (at == 0xffffffffu ? 1 : 0));
        atomicAdd(&out[1], 1ull);
    }
}

int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? atoi(argv[1]) : 64;
    const int blocks = argc > 2 ? atoi(argv[2]) : 256;
    const int hops = argc > 3 ? atoi(argv[3]) : 8;
    const int idle_us = argc > 4 ? atoi(argv[4]) : 0;
    const size_t n = mib * 1024 * 1024 / 4;
    std::vector<unsigned> perm(n / 32), next(n, 0);   // one entry per 128-byte line
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    for (size_t i = 0; i < perm.size(); ++i) next[(size_t)perm[i] * 32] = perm[(i + 1) % perm.size()] * 32;
    for (size_t i = 0; i < n; ++i) if (i % 32) next[i] = next[i - i % 32];
    unsigned* d; unsigned long long* out;
    hipMalloc(&d, n * 4); hipMalloc(&out, 16);
    hipMemcpy(d, next.data(), n * 4, hipMemcpyHostToDevice);
    const unsigned stride = (unsigned)(n / 1024 / 32) * 32;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(out, 0, 16);
        hipDeviceSynchronize();
        const int launches = 200;
        auto t0 = std::chrono::steady_clock::now();
        for (int l = 0; l < launches; ++l) {
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, d, hops, stride, out, (unsigned)(rep * 1000 + l));
            if (idle_us >= 0) {
                hipDeviceSynchronize();
                if (idle_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(idle_us));
            }
        }
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        unsigned long long h[2];
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        if (idle_us < 0) printf("  (%.3f ticks per ns)\n", (double)h[0] / h[1] * launches / (ms * 1e6));
        printf("buffer %zu MiB, %d waves, %d hops, idle %d us: %.0f cycles per hop, %.1f us per launch\n", mib, blocks, hops, idle_us,
               (double)h[0] / h[1] / hops, ms * 1e3 / launches);
    }
    return 0;
}
