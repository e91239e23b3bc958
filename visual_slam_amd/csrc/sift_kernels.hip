// SIFT-128 u8 brute-force 2-NN on int8 MFMA (gfx950), with the top-2 selection and the
// train-side nearest-query reduction fused into the epilogue: the N x M distance matrix
// never exists in memory.
//
// Stands in the position of matching(...) .. get_list_matching_pairs in the reference's
// per-pair loop (Others/SuperPoint_data/main.py:241-273); the mutual check and the -1
// convention follow Others/SuperPoint_data/models/superglue.py:276-286.
#include "vsm_internal.h"

namespace vsm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define VSM_LDS __attribute__((address_space(3)))
#define VSM_GLB __attribute__((address_space(1)))

// ---------------------------------------------------------------------------------
// u8 rows -> fragment-major int8 + shifted norms
// ---------------------------------------------------------------------------------
// One thread per 16-byte chunk of a row (8 threads per row, 8 rows per wave).
__global__ void sift_transform_kernel(const uint8_t* __restrict__ des, int n, int n_pad,
                                      int8_t* __restrict__ frag, int32_t* __restrict__ nrm8)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int row = (int)(gid >> 3);
    const int chunk = (int)(gid & 7);
    if (row >= n_pad) return;
    uint4 v = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
    if (row < n) v = *reinterpret_cast<const uint4*>(des + (size_t)row * kDescDim + chunk * 16);
    v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
    int sum = 0;
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            int e = (int)(int8_t)((w[i] >> (8 * b)) & 0xff);
            sum += e * e;
        }
    }
    // bytes k = 16*chunk .. +15 : k-step s = chunk >> 1, half h = chunk & 1
    const int tile = row >> 5, r = row & 31, s = chunk >> 1, h = chunk & 1;
    *reinterpret_cast<uint4*>(frag + (size_t)tile * kTileBytes + s * 1024 + (h * 32 + r) * 16) = v;
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    sum += __shfl_xor(sum, 4);
    if (chunk == 0) nrm8[row] = (row < n) ? (sum << 8) : kNormPad;
}

void launch_sift_transform(const uint8_t* des, int n, int n_pad, int8_t* frag, int32_t* nrm8,
                           hipStream_t s)
{
    const long long threads = (long long)n_pad * 8;
    const int block = 256;
    const int grid = (int)((threads + block - 1) / block);
    hipLaunchKernelGGL(sift_transform_kernel, dim3(grid), dim3(block), 0, s, des, n, n_pad, frag, nrm8);
}

// ---------------------------------------------------------------------------------
// main kernel
// ---------------------------------------------------------------------------------
// Workgroup = 8 waves = one 256-row strip of the query set against one chunk (<= 8192
// columns) of the train set.  Wave w keeps the MFMA A-operand of its 32 query rows in
// registers for the whole loop; each step streams a 128-column train tile (16 KiB, already
// in operand order in HBM) into LDS by LDS-DMA, double buffered, and every wave multiplies
// its rows by all four 32-column blocks.
//
// Accumulator layout of v_mfma_i32_32x32x32_i8: lane -> column (lane & 31), register ->
// row (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  So per lane a query row meets four new
// columns per step (one per block), and a column meets 16 query rows per block.
//
// Keys are packed so that one signed 32-bit min carries value and index together:
//   query side:  x = (nrmB[j] - 2 dot) * 256 + (step * 4 + block)      -> running top-2 / row
//   train side:  y = (nrmA[i] - 2 dot) * 256 + row-in-strip            -> min over rows
// (|value| < 2^23, so the product with 256 stays inside int32.)  Ties resolve to the lower
// index inside a lane; across lanes / strips the finishing code compares (value, index).
constexpr int kLdsB = 2 * 16384;
constexpr int kLdsNorm = 2 * 512;
constexpr int kLdsColmin = kChunkCols * 4;
constexpr int kLdsTotal = kLdsB + kLdsNorm + kLdsColmin;  // 66560
static_assert(kLdsTotal >= 256 * 32 * 8, "row merge buffer reuses the front of the LDS block");

__device__ __forceinline__ int min3i(int a, int b, int c)
{
    int r;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int med3i(int a, int b, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_dst)
{
    __builtin_amdgcn_global_load_lds((const VSM_GLB void*)gsrc, (VSM_LDS void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const void* gsrc, char* lds_dst)
{
    __builtin_amdgcn_global_load_lds((const VSM_GLB void*)gsrc, (VSM_LDS void*)lds_dst, 4, 0, 0);
}

__global__ __launch_bounds__(512, 2)
void sift_knn_kernel(const SiftPair* __restrict__ pairs, const SiftItem* __restrict__ items,
                     int32_t* __restrict__ rr_d1, int32_t* __restrict__ rr_j1,
                     int32_t* __restrict__ rr_d2, int32_t* __restrict__ colpart, int negscale)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // Blocks b and b + 8 share an XCD (and its L2).  Give every XCD a contiguous range of
    // work items so the 32 strips of one pair stream the same train set through one L2.
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r8 = nb & 7, xcd = bid & 7;
    const int L = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const SiftItem it = items[L];
    const SiftPair pw = pairs[it.pair];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane5 = lane & 31;
    const int half = lane >> 5;

    const int c0 = it.chunk * kChunkCols;
    const int ncols = min(kChunkCols, pw.nB_pad - c0);
    const int nsteps = ncols / kStepCols;

    int* colmin = reinterpret_cast<int*>(smem + kLdsB + kLdsNorm);
    for (int t = tid; t < ncols; t += 512) colmin[t] = INT32_MAX;

    // persistent A operand and per-row constants
    const int rowbase = it.strip * kStripRows + wave * 32;
    const int8_t* Abase = pw.fragA + (size_t)(rowbase >> 5) * kTileBytes + lane * 16;
    v4i a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = *reinterpret_cast<const v4i*>(Abase + s * 1024);
    int rc[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int rin = (reg & 3) + 8 * (reg >> 2) + 4 * half;
        rc[reg] = pw.nrmA[rowbase + rin] + (wave * 32 + rin);
    }
    int k1[16], k2[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) { k1[reg] = INT32_MAX; k2[reg] = INT32_MAX; }

    const int8_t* Bsrc = pw.fragB + (size_t)(c0 >> 5) * kTileBytes + wave * 2048 + lane * 16;
    const int32_t* Nsrc = pw.nrmB + c0 + (wave & 1) * 64 + lane;

    // prologue: stage step 0
    {
        char* dst = smem + wave * 2048;
        dma16(Bsrc, dst);
        dma16(Bsrc + 1024, dst + 1024);
        if (wave < 2) dma4(Nsrc, smem + kLdsB + wave * 256);
    }
    __syncthreads();

    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        if (st + 1 < nsteps) {
            char* dst = smem + (cur ^ 1) * 16384 + wave * 2048;
            const int8_t* src = Bsrc + (size_t)(st + 1) * 16384;
            dma16(src, dst);
            dma16(src + 1024, dst + 1024);
            if (wave < 2) dma4(Nsrc + (st + 1) * kStepCols, smem + kLdsB + (cur ^ 1) * 512 + wave * 256);
        }
        const char* Bt = smem + cur * 16384 + lane * 16;
        const int* nl = reinterpret_cast<const int*>(smem + kLdsB + cur * 512);

        v16i acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            v4i b0 = *reinterpret_cast<const v4i*>(Bt + cb * 4096);
            v4i b1 = *reinterpret_cast<const v4i*>(Bt + cb * 4096 + 1024);
            v4i b2 = *reinterpret_cast<const v4i*>(Bt + cb * 4096 + 2048);
            v4i b3 = *reinterpret_cast<const v4i*>(Bt + cb * 4096 + 3072);
            v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            z = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b0, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[1], b1, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2], b2, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[3], b3, z, 0, 0, 0);
            acc[cb] = z;
        }

        int cj[4], cm[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            cj[cb] = nl[cb * 32 + lane5] + (st * 4 + cb);
            asm volatile("" : "+v"(cj[cb]));  // keep the sum formed once, outside the mads
            cm[cb] = INT32_MAX;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int x0 = __mul24(acc[0][reg], negscale) + cj[0];
            const int x1 = __mul24(acc[1][reg], negscale) + cj[1];
            const int x2 = __mul24(acc[2][reg], negscale) + cj[2];
            const int x3 = __mul24(acc[3][reg], negscale) + cj[3];
            // running two smallest of {k1, k2, x0..x3}
            const int ua = med3i(k1[reg], x0, x1);
            const int ka = min3i(k1[reg], x0, x1);
            const int ub = med3i(ka, x2, x3);
            k1[reg] = min3i(ka, x2, x3);
            k2[reg] = min3i(k2[reg], ua, ub);
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
            for (int reg = 0; reg < 16; reg += 2) {
                const int y0 = __mul24(acc[cb][reg], negscale) + rc[reg];
                const int y1 = __mul24(acc[cb][reg + 1], negscale) + rc[reg + 1];
                cm[cb] = min3i(cm[cb], y0, y1);
            }
            atomicMin(&colmin[st * kStepCols + cb * 32 + lane5], cm[cb]);
        }
        __syncthreads();
    }

    // train-side partial minima of this strip
    {
        int32_t* dst = colpart + pw.col_off + (size_t)it.strip * pw.nB_pad + c0;
        for (int t = tid; t < ncols; t += 512) dst[t] = colmin[t];
    }
    __syncthreads();

    // query-side: merge the 32 per-lane candidates of every row
    int2* rowbuf = reinterpret_cast<int2*>(smem);  // [256][32]
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int rin = (reg & 3) + 8 * (reg >> 2) + 4 * half;
        rowbuf[(wave * 32 + rin) * 32 + lane5] = make_int2(k1[reg], k2[reg]);
    }
    __syncthreads();
    {
        const int row = tid >> 1, part = tid & 1;
        long long best = INT64_MAX;
        int second = kDistPad;
        for (int e = 0; e < 16; ++e) {
            const int l5 = part * 16 + e;
            const int2 kk = rowbuf[row * 32 + l5];
            const int idx8 = kk.x & 255;
            const int col = c0 + (idx8 >> 2) * kStepCols + (idx8 & 3) * 32 + l5;
            const int d = kk.x >> 8;
            const long long comp = (long long)d * 4294967296LL + (unsigned)col;
            if (comp < best) {
                second = min(second, (int)(best >> 32));
                best = comp;
            } else {
                second = min(second, d);
            }
            second = min(second, kk.y >> 8);
        }
        // combine the two half-rows (neighbouring lanes)
        const long long obest = __shfl_xor(best, 1);
        const int osecond = __shfl_xor(second, 1);
        const long long lo = best < obest ? best : obest;
        const long long hi = best < obest ? obest : best;
        second = min(min(second, osecond), (int)(hi >> 32));
        if (part == 0) {
            const int grow = it.strip * kStripRows + row;
            const int na = pw.nrmA[grow] >> 8;
            const size_t o = pw.row_off + (size_t)it.chunk * pw.nA_pad + grow;
            const int d1 = (int)(lo >> 32);
            rr_d1[o] = d1 >= kDistPad ? INT32_MAX : d1 + na;
            rr_j1[o] = (int)(unsigned)(lo & 0xffffffffu);
            rr_d2[o] = second >= kDistPad ? INT32_MAX : second + na;
        }
    }
}

void launch_sift_knn(const SiftPair* pairs, const SiftItem* items, int nitems,
                     int32_t* rr_d1, int32_t* rr_j1, int32_t* rr_d2, int32_t* colpart,
                     hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(sift_knn_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
        attr_set = true;
    }
    hipLaunchKernelGGL(sift_knn_kernel, dim3(nitems), dim3(512), kLdsTotal, s,
                       pairs, items, rr_d1, rr_j1, rr_d2, colpart, -512);
}

// ---------------------------------------------------------------------------------
// finishing kernel: chunk merge, train-side strip merge, tests, compaction
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void sift_finalize_kernel(const SiftPair* __restrict__ pairs,
                          const int32_t* __restrict__ rr_d1, const int32_t* __restrict__ rr_j1,
                          const int32_t* __restrict__ rr_d2, const int32_t* __restrict__ colpart,
                          MatchRule rule, SiftOutputs out)
{
    __shared__ int wsum[16];
    __shared__ int base_s;
    const SiftPair pw = pairs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (tid == 0) base_s = 0;
    __syncthreads();

    for (int i0 = 0; i0 < pw.nA; i0 += 1024) {
        const int i = i0 + tid;
        int j1 = -1, d1 = INT32_MAX, d2 = INT32_MAX;
        bool ok = false;
        if (i < pw.nA) {
            // merge chunks (chunk order = ascending column index, so a strict compare
            // keeps the lowest column on ties)
            for (int c = 0; c < pw.nchunks; ++c) {
                const size_t o = pw.row_off + (size_t)c * pw.nA_pad + i;
                const int e1 = rr_d1[o], ej = rr_j1[o], e2 = rr_d2[o];
                if (e1 < d1) {
                    d2 = min(d1, e2);
                    d1 = e1;
                    j1 = ej;
                } else {
                    d2 = min(d2, e1);
                }
            }
            if (j1 >= pw.nB) j1 = -1;  // only padding was seen (cannot happen for nB >= 1)
            ok = j1 >= 0;
            if (out.knn_j1) {
                out.knn_j1[pw.out0_off + i] = j1;
                out.knn_d1[pw.out0_off + i] = d1;
                out.knn_d2[pw.out0_off + i] = d2;
            }
            if (ok && rule.max_dist >= 0 && (long long)d1 > rule.max_dist) ok = false;
            if (ok && rule.num2 > 0 && d2 != INT32_MAX) {
                if (!((long long)d1 * rule.den2 < (long long)d2 * rule.num2)) ok = false;
            }
            if (ok && rule.mutual) {
                // nearest query row of column j1: min over strips of (value, row)
                long long best = INT64_MAX;
                for (int s = 0; s < pw.nstrips; ++s) {
                    const int key = colpart[pw.col_off + (size_t)s * pw.nB_pad + j1];
                    const long long comp = (long long)(key >> 8) * 4294967296LL +
                                           (unsigned)(s * kStripRows + (key & 255));
                    best = comp < best ? comp : best;
                }
                ok = (int)(unsigned)(best & 0xffffffffu) == i;
            }
            if (out.dense0) out.dense0[pw.out0_off + i] = ok ? j1 : -1;
        }
        // block-wide exclusive scan of ok
        const unsigned long long bal = __ballot(ok);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            const int c = wsum[w];
            if (w < wave) woff += c;
            tot += c;
        }
        const int base = base_s;
        if (ok) {
            const size_t o = pw.out0_off + base + woff + before;
            out.slot_i0[o] = i;
            out.slot_i1[o] = j1;
            if (out.slot_d) out.slot_d[o] = d1;
        }
        __syncthreads();
        if (tid == 0) base_s = base + tot;
        __syncthreads();
    }
    if (tid == 0) out.counts[blockIdx.x] = base_s;
}

// dense1[j] = lowest query row matched to train row j (a second pass keeps it simple and
// deterministic; with the mutual check at most one query row maps to each j anyway)
__global__ void sift_dense1_kernel(const SiftPair* __restrict__ pairs, const int32_t* __restrict__ counts,
                                   const int32_t* __restrict__ slot_i0, const int32_t* __restrict__ slot_i1,
                                   int32_t* __restrict__ dense1)
{
    const SiftPair pw = pairs[blockIdx.x];
    const int cnt = counts[blockIdx.x];
    for (int t = threadIdx.x; t < pw.nB; t += blockDim.x) dense1[pw.out1_off + t] = INT32_MAX;
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += blockDim.x)
        atomicMin(&dense1[pw.out1_off + slot_i1[pw.out0_off + t]], slot_i0[pw.out0_off + t]);
    __syncthreads();
    for (int t = threadIdx.x; t < pw.nB; t += blockDim.x)
        if (dense1[pw.out1_off + t] == INT32_MAX) dense1[pw.out1_off + t] = -1;
}

void launch_sift_finalize(const SiftPair* pairs, int npairs,
                          const int32_t* rr_d1, const int32_t* rr_j1, const int32_t* rr_d2,
                          const int32_t* colpart, MatchRule rule, SiftOutputs out, hipStream_t s)
{
    int32_t* dense1 = out.dense1;
    out.dense1 = nullptr;  // filled by the second pass below
    hipLaunchKernelGGL(sift_finalize_kernel, dim3(npairs), dim3(1024), 0, s,
                       pairs, rr_d1, rr_j1, rr_d2, colpart, rule, out);
    if (dense1)
        hipLaunchKernelGGL(sift_dense1_kernel, dim3(npairs), dim3(1024), 0, s,
                           pairs, out.counts, out.slot_i0, out.slot_i1, dense1);
}

// ---------------------------------------------------------------------------------
// counts -> offsets, slots -> tight arrays
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void scan_counts_kernel(const int32_t* __restrict__ counts, int npairs, int64_t* __restrict__ offsets)
{
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int p0 = 0; p0 < npairs; p0 += 1024) {
        const int p = p0 + tid;
        long long v = p < npairs ? counts[p] : 0;
        long long incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const long long o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        long long woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) woff += wsum[w];
            tot += wsum[w];
        }
        const long long carry = carry_s;
        if (p < npairs) offsets[p] = carry + woff + incl - v;
        __syncthreads();
        if (tid == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (tid == 0) offsets[npairs] = carry_s;
}

__global__ void pack_kernel(const SiftPair* __restrict__ pairs, const int32_t* __restrict__ counts,
                            const int64_t* __restrict__ offsets,
                            const int32_t* __restrict__ slot_i0, const int32_t* __restrict__ slot_i1,
                            int32_t* __restrict__ tight_i0, int32_t* __restrict__ tight_i1)
{
    const int p = blockIdx.x;
    const int cnt = counts[p];
    const int64_t src = pairs[p].out0_off, dst = offsets[p];
    for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
        tight_i0[dst + t] = slot_i0[src + t];
        tight_i1[dst + t] = slot_i1[src + t];
    }
}

void launch_pack(const SiftPair* pairs, int npairs, const int32_t* counts, int64_t* offsets,
                 const int32_t* slot_i0, const int32_t* slot_i1,
                 int32_t* tight_i0, int32_t* tight_i1, hipStream_t s)
{
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, s, counts, npairs, offsets);
    hipLaunchKernelGGL(pack_kernel, dim3(npairs), dim3(256), 0, s,
                       pairs, counts, offsets, slot_i0, slot_i1, tight_i0, tight_i1);
}

}  // namespace vsm
