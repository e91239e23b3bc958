// Issue-rate calibration for the instructions of the matching kernels' inner loops (gfx950):
// cycles per wave64 instruction per SIMD, measured with s_memtime around long unrolled runs of
// independent instructions, at 1, 2 and 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define REP8(x) x x x x x x x x
#define OPS_PER_ITER 64

template <int KIND>
__global__ void probe(int iters, long long* out, int* sink, int negscale)
{
    int r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * (i + 3);
    int a = threadIdx.x ^ 0x55, b = blockIdx.x + 7;
    v16i acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
    v4i fa = {(int)threadIdx.x, 1, 2, 3}, fb = {4, 5, 6, (int)blockIdx.x};
    __syncthreads();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            REP8(asm volatile("v_min3_i32 %0, %0, %8, %9\n v_min3_i32 %1, %1, %8, %9\n v_min3_i32 %2, %2, %8, %9\n v_min3_i32 %3, %3, %8, %9\n"
                              "v_min3_i32 %4, %4, %8, %9\n v_min3_i32 %5, %5, %8, %9\n v_min3_i32 %6, %6, %8, %9\n v_min3_i32 %7, %7, %8, %9\n"
                              : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b));)
        } else if (KIND == 1) {
            REP8(asm volatile("v_med3_i32 %0, %0, %8, %9\n v_med3_i32 %1, %1, %8, %9\n v_med3_i32 %2, %2, %8, %9\n v_med3_i32 %3, %3, %8, %9\n"
                              "v_med3_i32 %4, %4, %8, %9\n v_med3_i32 %5, %5, %8, %9\n v_med3_i32 %6, %6, %8, %9\n v_med3_i32 %7, %7, %8, %9\n"
                              : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b));)
        } else if (KIND == 2) {
            REP8(asm volatile("v_mad_i32_i24 %0, %8, %10, %0\n v_mad_i32_i24 %1, %8, %10, %1\n v_mad_i32_i24 %2, %8, %10, %2\n v_mad_i32_i24 %3, %8, %10, %3\n"
                              "v_mad_i32_i24 %4, %9, %10, %4\n v_mad_i32_i24 %5, %9, %10, %5\n v_mad_i32_i24 %6, %9, %10, %6\n v_mad_i32_i24 %7, %9, %10, %7\n"
                              : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b), "s"(negscale));)
        } else if (KIND == 3) {
            REP8(asm volatile("v_min_i32 %0, %0, %8\n v_min_i32 %1, %1, %8\n v_min_i32 %2, %2, %8\n v_min_i32 %3, %3, %8\n"
                              "v_max_i32 %4, %4, %9\n v_max_i32 %5, %5, %9\n v_max_i32 %6, %6, %9\n v_max_i32 %7, %7, %9\n"
                              : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b));)
        } else if (KIND == 4) {   // MFMA only: 8 per iteration, two independent chains
            for (int k = 0; k < 4; ++k) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb, fa, acc1, 0, 0, 0);
            }
        } else if (KIND == 5) {   // 8 MFMA + 64 VALU (min3) per iteration, independent
            for (int k = 0; k < 4; ++k) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc0, 0, 0, 0);
                asm volatile("v_min3_i32 %0, %0, %8, %9\n v_min3_i32 %1, %1, %8, %9\n v_min3_i32 %2, %2, %8, %9\n v_min3_i32 %3, %3, %8, %9\n"
                             "v_min3_i32 %4, %4, %8, %9\n v_min3_i32 %5, %5, %8, %9\n v_min3_i32 %6, %6, %8, %9\n v_min3_i32 %7, %7, %8, %9\n"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b));
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb, fa, acc1, 0, 0, 0);
                asm volatile("v_min3_i32 %0, %0, %8, %9\n v_min3_i32 %1, %1, %8, %9\n v_min3_i32 %2, %2, %8, %9\n v_min3_i32 %3, %3, %8, %9\n"
                             "v_min3_i32 %4, %4, %8, %9\n v_min3_i32 %5, %5, %8, %9\n v_min3_i32 %6, %6, %8, %9\n v_min3_i32 %7, %7, %8, %9\n"
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b));
            }
        }
    }
    const long long c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = c1 - c0;
    int s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    if (s == 0x12345678 || acc0[5] == 0x7654321) sink[0] = s + acc0[0] + acc1[3];
}

template <int KIND>
void run(const char* name, int ops_per_iter, int mfma_per_iter, long long* out, int* sink)
{
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        // 256 CUs x 4 SIMDs; a block of 256 threads puts one wave on every SIMD of a CU
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<KIND>, dim3(256 * wps), dim3(256), 0, 0, 10, out, sink, -512);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<KIND>, dim3(256 * wps), dim3(256), 0, 0, iters, out, sink, -512);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        long long h = 0;
        (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
        const double per_wave = (double)h / iters;
        printf("%-28s waves/SIMD %d: %7.1f clk per iteration per wave", name, wps, per_wave);
        if (ops_per_iter) printf("  -> %.2f clk per VALU op per SIMD", per_wave / ops_per_iter / wps * (mfma_per_iter ? 0 : 1));
        if (mfma_per_iter) printf("  (%d MFMA + %d VALU per iteration: %.1f clk per SIMD per iteration)", mfma_per_iter, ops_per_iter, per_wave / wps);
        printf("  [kernel %.3f ms]\n", ms);
    }
}

int main()
{
    long long* out;
    int* sink;
    (void)hipMalloc(&out, 16);
    (void)hipMalloc(&sink, 4);
    run<0>("v_min3_i32", 64, 0, out, sink);
    run<1>("v_med3_i32", 64, 0, out, sink);
    run<2>("v_mad_i32_i24 (sgpr scale)", 64, 0, out, sink);
    run<3>("v_min_i32 / v_max_i32", 64, 0, out, sink);
    run<4>("mfma_i32_32x32x32_i8", 0, 8, out, sink);
    run<5>("8 mfma + 64 min3", 64, 8, out, sink);
    return 0;
}
