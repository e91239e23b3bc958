// Cycles until two rows of 1 KiB per 8-lane group have arrived, for two lane-to-byte mappings.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ void rows(const float4* __restrict__ base, int nrows, unsigned seed, unsigned long long* out, float* sink)
{
    const int lane = threadIdx.x & 63, sub = lane & 7;
    const unsigned g = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const unsigned r0 = ((g * 2654435761u + seed * 40503u) >> 7) % nrows, r1 = ((g * 40503u + seed * 2654435761u) >> 9) % nrows;
    const float4* p0 = base + (size_t)r0 * 64, *p1 = base + (size_t)r1 * 64;
    float4 a[8], b[8];
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int at = MODE == 0 ? sub * 8 + u : u * 8 + sub;
        a[u] = p0[at];
        b[u] = p1[at];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(a[u].x), "+v"(b[u].x));
    asm volatile("s_waitcnt vmcnt(0)");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += a[u].x + b[u].y;
    if (s == 12345.678f) *sink = s;
    if (lane == 0) {
        atomicAdd(&out[0], t1 - t0);
        atomicAdd(&out[1], 1ull);
    }
}

int main(int argc, char** argv)
{
    const int nrows = argc > 1 ? atoi(argv[1]) : 4096;
    const int blocks = argc > 2 ? atoi(argv[2]) : 256;
    float4* d; unsigned long long* out; float* sink;
    hipMalloc(&d, (size_t)nrows * 1024); hipMalloc(&out, 16); hipMalloc(&sink, 4);
    hipMemset(d, 0, (size_t)nrows * 1024);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(out, 0, 16);
            hipDeviceSynchronize();
            for (int l = 0; l < 100; ++l) {
                if (mode == 0) hipLaunchKernelGGL(rows<0>, dim3(blocks), dim3(256), 0, 0, d, nrows, (unsigned)l, out, sink);
                else hipLaunchKernelGGL(rows<1>, dim3(blocks), dim3(256), 0, 0, d, nrows, (unsigned)l, out, sink);
                hipDeviceSynchronize();
            }
            unsigned long long h[2];
            hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            printf("%d rows, %d blocks of 4 waves, mapping %s: %.0f cycles until 2 rows per group are in\n", nrows, blocks,
                   mode == 0 ? "lane = 128 contiguous bytes" : "instruction = whole lines", (double)h[0] / h[1]);
        }
    return 0;
}
