// Where do workgroups land, and do they overlap in time?  Every block records its XCC id, hardware
// id (SE / CU / SIMD of wave 0) and s_memtime at start and end of a fixed amount of MFMA work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct Rec { long long t0, t1; unsigned hwid, xcc; };

__global__ void work(int iters, Rec* rec, int* sink)
{
    v16i acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
    v4i fa = {(int)threadIdx.x, 1, 2, 3}, fb = {4, 5, 6, (int)blockIdx.x};
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb, fa, acc1, 0, 0, 0);
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        Rec r;
        r.t0 = t0; r.t1 = t1;
        r.hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_ID, 32 bits
        r.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));    // XCC_ID
        rec[blockIdx.x] = r;
    }
    if (acc0[5] == 0x7654321) sink[0] = acc0[0] + acc1[3];
}

int main()
{
    const int nblocks = 2048;
    Rec* d;
    int* sink;
    (void)hipMalloc(&d, sizeof(Rec) * nblocks);
    (void)hipMalloc(&sink, 4);
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL(work, dim3(nblocks), dim3(threads), 0, 0, 20000, d, sink);
        (void)hipDeviceSynchronize();
        std::vector<Rec> h(nblocks);
        (void)hipMemcpy(h.data(), d, sizeof(Rec) * nblocks, hipMemcpyDeviceToHost);
        long long tmin = h[0].t0, tmax = h[0].t1;
        for (auto& r : h) { tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); }
        std::map<unsigned, int> per_cu;
        // blocks alive at the midpoint of block 0's life, by (xcc, se, cu)
        const long long mid = (h[0].t0 + h[0].t1) / 2;
        int alive = 0;
        for (auto& r : h) {
            if (r.t0 <= mid && r.t1 >= mid) {
                ++alive;
                const unsigned cu = (r.hwid >> 8) & 15, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 7;
                per_cu[((r.xcc & 15) << 12) | (se << 8) | (sh << 4) | cu]++;
            }
        }
        std::map<int, int> hist;
        for (auto& kv : per_cu) hist[kv.second]++;
        printf("threads %d: %d blocks, total %.3f ms, block 0 lived %.3f ms; alive at its midpoint: %d blocks on %zu CUs;",
               threads, nblocks, (tmax - tmin) / 1e5, (h[0].t1 - h[0].t0) / 1e5, alive, per_cu.size());
        for (auto& kv : hist) printf(" [%d CUs hold %d]", kv.second, kv.first);
        printf("\n  first blocks: ");
        for (int b = 0; b < 12; ++b)
            printf("b%d xcc%u se%u cu%u simd%u t0=%lld | ", b, h[b].xcc & 15, (h[b].hwid >> 13) & 7, (h[b].hwid >> 8) & 15,
                   (h[b].hwid >> 4) & 3, h[b].t0 - tmin);
        printf("\n");
    }
    return 0;
}
