// Cost of running through straight-line code once per kernel launch: the same block of N instructions is executed
// twice by every wave; the first pass fetches it, the second finds it in the instruction cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int N>
__device__ __forceinline__ float block_of_code(float x, float y)
{
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    return x;
}

template <int N>
__global__ void passes(unsigned long long* out, float* sink, float y)
{
    float x = threadIdx.x;
    unsigned long long t[3];
    t[0] = __builtin_readcyclecounter();
#pragma unroll 1
    for (int p = 0; p < 2; ++p) {
        x = block_of_code<N>(x, y);
        asm volatile("" : "+v"(x));
        t[p + 1] = __builtin_readcyclecounter();
    }
    if (x == 12345.f) *sink = x;
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], t[1] - t[0]);
        atomicAdd(&out[1], t[2] - t[1]);
        atomicAdd(&out[2], 1ull);
    }
}

template <int N>
void run(int blocks, unsigned long long* out, float* sink)
{
    hipMemset(out, 0, 32);
    for (int l = 0; l < 50; ++l) {
        hipLaunchKernelGGL(passes<N>, dim3(blocks), dim3(256), 0, 0, out, sink, 0.5f);
        hipDeviceSynchronize();
    }
    unsigned long long h[3];
    hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    printf("%5d instructions (%3d KiB), %d blocks: first pass %.0f cycles, second pass %.0f cycles per wave\n", N, N * 8 / 1024, blocks,
           (double)h[0] / h[2], (double)h[1] / h[2]);
}

int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    unsigned long long* out; float* sink;
    hipMalloc(&out, 32); hipMalloc(&sink, 4);
    run<256>(blocks, out, sink);
    run<1024>(blocks, out, sink);
    run<4096>(blocks, out, sink);
    run<4096>(blocks, out, sink);
    return 0;
}
