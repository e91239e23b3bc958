// Measures the shader clock a kernel actually runs at: clock64() (s_memtime, shader clocks) against
// wall_clock64() (s_memrealtime, constant 100 MHz), (a) with one idle-ish wave on the chip and
// (b) with every SIMD busy issuing int8 MFMA + VALU work like the matching kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void spin(long long iters, int heavy, int valu_per_mfma, long long* out, int* sink)
{
    const long long c0 = clock64();
    const long long w0 = wall_clock64();
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    int x = threadIdx.x;
    for (long long i = 0; i < iters; ++i) {
        if (heavy) {
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
            for (int v = 0; v < valu_per_mfma; ++v) x = min(x * 3 + 1, acc[v & 15] ^ x);
        } else {
            x = x * 3 + 1;
        }
    }
    const long long c1 = clock64();
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
    }
    if (x == 0x7fffffff) sink[0] = x + acc[0];
}

int main()
{
    long long* out;
    int* sink;
    hipMalloc(&out, 16);
    hipMalloc(&sink, 4);
    long long h[2];
    struct { const char* name; int blocks, threads, heavy, vpm; long long iters; } cases[] = {
        {"one wave, scalar-ish loop", 1, 64, 0, 0, 4000000},
        {"full chip, MFMA only", 2048, 256, 1, 0, 200000},
        {"full chip, MFMA + 9 VALU per MFMA", 2048, 256, 1, 9, 60000},
        {"full chip, MFMA + 36 VALU per MFMA", 2048, 256, 1, 36, 20000},
    };
    for (auto& cs : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(spin, dim3(cs.blocks), dim3(cs.threads), 0, 0, cs.iters, cs.heavy, cs.vpm, out, sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("%-40s shader clocks %lld, 100 MHz ticks %lld -> %.0f MHz (%.2f ms)\n", cs.name, h[0], h[1],
               100.0 * h[0] / h[1], h[1] / 1e5);
    }
    return 0;
}
