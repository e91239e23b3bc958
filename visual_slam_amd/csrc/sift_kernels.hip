// SIFT-128 u8 brute-force matching on int8 MFMA (gfx950).  The N x M distance matrix never
// exists in memory: the nearest / runner-up selection is fused into the MFMA epilogue.
//
// Stands in the position of matching(...) .. get_list_matching_pairs in the reference's
// per-pair loop (Others/SuperPoint_data/main.py:241-273); the mutual check and the -1
// convention follow Others/SuperPoint_data/models/superglue.py:276-286.
//
// Pipeline per submission of image pairs:
//   1. query pass   (MFMA)  every image-0 row: nearest image-1 row, distance, runner-up distance
//                            (tile-minimum form: smallest half-distance, the 8-row group holding it, the smallest
//                            half-distance outside that group)
//   2. select                distance cap + ratio test -> compact candidate list per pair
//                            (tile-minimum form: optimistic tests, then resolve / rescan of the candidates: exact
//                            row, distance and tests; then the list of image-0 rows the mutual check must look at)
//   3. verify pass  (MFMA)  every candidate's image-1 row: nearest image-0 row (mutual check);
//                            only the candidates' columns are revisited, not the whole matrix
//   4. confirm               keep candidates whose nearest image-0 row is the candidate itself
#include "vsm_internal.h"

#include <algorithm>
#include <type_traits>

namespace vsm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define VSM_LDS __attribute__((address_space(3)))
#define VSM_GLB __attribute__((address_space(1)))

// ---------------------------------------------------------------------------------
// u8 rows -> fragment-major int8 + shifted norms
// ---------------------------------------------------------------------------------
// One thread per 16-byte chunk of a row (8 threads per row, 8 rows per wave).
// Besides nrm8c the kernel writes, at nrm8c[n_pad + row], the constant the tile-minimum query pass starts
// its accumulators from: floor(nrm / 2) + (sum of the row's int8 values), kYPad for padding rows.
__global__ void sift_transform_kernel(const uint8_t* __restrict__ des, int n, int n_pad,
                                      int8_t* __restrict__ frag, int32_t* __restrict__ nrm8c)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int row = (int)(gid >> 3);
    const int chunk = (int)(gid & 7);
    if (row >= n_pad) return;
    uint4 v = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
    if (row < n) v = *reinterpret_cast<const uint4*>(des + (size_t)row * kDescDim + chunk * 16);
    v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
    int sum = 0, lin = 0;
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            int e = (int)(int8_t)((w[i] >> (8 * b)) & 0xff);
            sum += e * e;
            lin += e;
        }
    }
    // bytes k = 16*chunk .. +15 : k-step s = chunk >> 1, half h = chunk & 1
    const int tile = row >> 5, r = row & 31, s = chunk >> 1, h = chunk & 1;
    *reinterpret_cast<uint4*>(frag + (size_t)tile * kTileBytes + s * 1024 + (h * 32 + r) * 16) = v;
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    sum += __shfl_xor(sum, 4);
    lin += __shfl_xor(lin, 1);
    lin += __shfl_xor(lin, 2);
    lin += __shfl_xor(lin, 4);
    if (chunk == 0) {
        nrm8c[row] = ((row < n) ? (sum << 8) : kNormPad) | (row & 255);
        nrm8c[n_pad + row] = row < n ? (sum >> 1) + lin : kYPad;
    }
}

void launch_sift_transform(const uint8_t* des, int n, int n_pad, int8_t* frag, int32_t* nrm8c,
                           hipStream_t s)
{
    const long long threads = (long long)n_pad * 8;
    const int block = 256;
    const int grid = (int)((threads + block - 1) / block);
    hipLaunchKernelGGL(sift_transform_kernel, dim3(grid), dim3(block), 0, s, des, n, n_pad, frag, nrm8c);
}

// ---------------------------------------------------------------------------------
// streaming MFMA passes
// ---------------------------------------------------------------------------------
// Workgroup = 4 waves.  Each wave keeps the MFMA operands of a few blocks of 32 "persistent"
// rows in registers; the other set streams through LDS in 256-row steps (32 KiB, already in
// operand order in HBM, copied by LDS-DMA, double buffered), and every wave multiplies its
// blocks by all eight 32-row tiles of the step, reading each tile from LDS once.
//
// The persistent rows sit on the accumulator's LANE axis (they are the B operand): lane l owns
// persistent row (l & 31), and its 16 accumulator registers hold streamed rows
// (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5) of the tile.  Everything a lane sees therefore
// belongs to ONE persistent row, so the running minimum lives in two registers per lane and
// needs no cross-lane traffic until the very end (lanes l and l + 32 hold disjoint streamed
// rows of the same persistent row).
//
// Keys are packed so one signed 32-bit min carries value and index:
//     key = (nrm[streamed row] - 2 dot) * 256 + (streamed row & 255)
// (|value| < 2^23, so the product with 256 stays inside int32; nrm8c already holds
// nrm * 256 + (row & 255)).  The packed minimum is exact inside a group of 256 streamed rows
// and resolves ties to the lower row; after every group it is folded into an unpacked running
// best with a strict compare, which keeps the lower row across groups as well.
constexpr int kTilesPerStep = kStepRows / 32;            // 8 tiles of 32 streamed rows
constexpr int kLdsTile = kStepRows * kDescDim;           // 32 KiB of descriptors per step
constexpr int kLdsNorm = kStepRows * 4;                  // 1 KiB of packed norms per step
constexpr int kLdsStream = 2 * kLdsTile + 2 * kLdsNorm;  // double buffered
static_assert(kStepRows == 256, "a step is exactly one 256-row index group");

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

// Blocks b and b + 8 share an XCD (and its L2).  Give every XCD a contiguous range of work
// items so that the strips of one image pair stream the same set through one L2.
__device__ __forceinline__ int xcd_contiguous_item(int bid, int nb)
{
    const int q = nb >> 3, r8 = nb & 7, xcd = bid & 7;
    return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
}

// one step's LDS-DMA: 32 KiB of operand-order descriptors (4 KiB per wave) + 256 packed norms.
// frag_src / nrm_src already include this wave's and lane's offsets.
template <int NWAVES>
__device__ __forceinline__ void stage_step(char* smem, int buf, const int8_t* frag_src, const int32_t* nrm_src,
                                           int wave)
{
    constexpr int kPieces = kLdsTile / 1024 / NWAVES;   // 1 KiB LDS-DMA pieces per wave
    char* dst = smem + buf * kLdsTile + wave * (kPieces * 1024);
#pragma unroll
    for (int p = 0; p < kPieces; ++p) dma16(frag_src + p * 1024, dst + p * 1024);
    if (wave < 4) dma4(nrm_src, smem + 2 * kLdsTile + buf * kLdsNorm + wave * 256);
}

// the 16 packed norms of the streamed rows this lane's registers hold (4 x 16-byte LDS reads)
__device__ __forceinline__ void tile_consts(const int* nl_tile, int half, int (&c)[16])
{
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        const v4i v = *reinterpret_cast<const v4i*>(nl_tile + 8 * qd + 4 * half);
        c[4 * qd + 0] = v[0];
        c[4 * qd + 1] = v[1];
        c[4 * qd + 2] = v[2];
        c[4 * qd + 3] = v[3];
    }
}

// ---- 1. query pass ------------------------------------------------------------------
// A workgroup is kQueryWaves waves; each wave owns kQueryRB blocks of 32 query rows, which share
// every streamed operand and norm read (a quarter of the LDS traffic per MFMA of one block per
// wave).  Two workgroups fit a CU, so one computes while the other waits at its step barrier.
constexpr int kQueryWaves = 4;
constexpr int kQueryRB = kStripRows / (32 * kQueryWaves);

// SECOND = false: the caller has no use for the runner-up distance (no ratio test, no raw 2-NN output); the
// epilogue is then 1.5 instead of 2.25 vector instructions per distance and rr_d2 receives "none".
template <bool SECOND>
__global__ __launch_bounds__(kQueryWaves * 64, 2)
void sift_query_kernel(const SiftPair* __restrict__ pairs, const SiftItem* __restrict__ items,
                       int32_t* __restrict__ rr_d1, int32_t* __restrict__ rr_j1,
                       int32_t* __restrict__ rr_d2, int negscale)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SiftItem it = items[xcd_contiguous_item(blockIdx.x, gridDim.x)];
    const SiftPair pw = pairs[it.pair];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane5 = lane & 31;
    const int half = lane >> 5;

    const int c0 = it.chunk * pw.chunk_cols;
    const int ncols = min(pw.chunk_cols, pw.nB_pad - c0);
    const int nsteps = ncols / kStepRows;  // sets are padded to whole steps

    // persistent operand: this wave's query rows
    const int qtile = (it.strip * (kStripRows / 32)) + wave * kQueryRB;
    v4i pf[kQueryRB][4];
#pragma unroll
    for (int rb = 0; rb < kQueryRB; ++rb) {
        const int8_t* src = pw.fragA + (size_t)(qtile + rb) * kTileBytes + lane * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s) pf[rb][s] = *reinterpret_cast<const v4i*>(src + s * 1024);
    }

    const int8_t* Bsrc = pw.fragB + (size_t)(c0 >> 5) * kTileBytes + wave * (kLdsTile / kQueryWaves) + lane * 16;
    const int32_t* Nsrc = pw.nrmB + c0 + (wave & 3) * 64 + lane;

    // unpacked running result per query row
    int bestd[kQueryRB], bestcol[kQueryRB], second[kQueryRB];
#pragma unroll
    for (int rb = 0; rb < kQueryRB; ++rb) { bestd[rb] = kDistPad; bestcol[rb] = INT32_MAX; second[rb] = kDistPad; }

    stage_step<kQueryWaves>(smem, 0, Bsrc, Nsrc, wave);
    __syncthreads();

    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        if (st + 1 < nsteps)
            stage_step<kQueryWaves>(smem, cur ^ 1, Bsrc + (size_t)(st + 1) * kLdsTile, Nsrc + (st + 1) * kStepRows, wave);
        const char* Bt = smem + cur * kLdsTile + lane * 16;
        const int* nl = reinterpret_cast<const int*>(smem + 2 * kLdsTile + cur * kLdsNorm);

        int k1[kQueryRB], k2[kQueryRB];  // packed two smallest of this 256-row group
#pragma unroll
        for (int rb = 0; rb < kQueryRB; ++rb) { k1[rb] = INT32_MAX; k2[rb] = INT32_MAX; }

#pragma unroll
        for (int t = 0; t < kTilesPerStep; ++t) {
            const char* tile = Bt + t * 4096;
            const v4i b0 = *reinterpret_cast<const v4i*>(tile);
            const v4i b1 = *reinterpret_cast<const v4i*>(tile + 1024);
            const v4i b2 = *reinterpret_cast<const v4i*>(tile + 2048);
            const v4i b3 = *reinterpret_cast<const v4i*>(tile + 3072);
            int c[16];
            tile_consts(nl + t * 32, half, c);
            v16i acc[kQueryRB];
#pragma unroll
            for (int rb = 0; rb < kQueryRB; ++rb) {
                v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                z = __builtin_amdgcn_mfma_i32_32x32x32_i8(b0, pf[rb][0], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_i32_32x32x32_i8(b1, pf[rb][1], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_i32_32x32x32_i8(b2, pf[rb][2], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_i32_32x32x32_i8(b3, pf[rb][3], z, 0, 0, 0);
                acc[rb] = z;
            }
#pragma unroll
            for (int rb = 0; rb < kQueryRB; ++rb) {
                int uprev = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int x0 = __mul24(acc[rb][2 * e], negscale) + c[2 * e];
                    const int x1 = __mul24(acc[rb][2 * e + 1], negscale) + c[2 * e + 1];
                    if (SECOND) {
                        const int u = med3i(k1[rb], x0, x1);   // runner-up of {k1, x0, x1}
                        k1[rb] = min3i(k1[rb], x0, x1);
                        if (e & 1) k2[rb] = min3i(k2[rb], uprev, u);
                        uprev = u;
                    } else {
                        k1[rb] = min3i(k1[rb], x0, x1);
                    }
                }
            }
        }
#pragma unroll
        for (int rb = 0; rb < kQueryRB; ++rb) {
            // fold the finished 256-row group into the running result (strict compare: on equal
            // distances the earlier group, i.e. the lower row, stays)
            const int d = k1[rb] >> 8;
            const int col = c0 + st * kStepRows + (k1[rb] & 255);
            if (d < bestd[rb]) {
                if (SECOND) second[rb] = min(second[rb], bestd[rb]);
                bestd[rb] = d;
                bestcol[rb] = col;
            } else if (SECOND) {
                second[rb] = min(second[rb], d);
            }
            if (SECOND) second[rb] = min(second[rb], k2[rb] >> 8);
        }
        __syncthreads();
    }

#pragma unroll
    for (int rb = 0; rb < kQueryRB; ++rb) {
        // lanes l and l + 32 saw disjoint columns of the same query row
        int bd = bestd[rb], bc = bestcol[rb], sc = second[rb];
        const int od = __shfl_xor(bd, 32);
        const int oc = __shfl_xor(bc, 32);
        const int os = __shfl_xor(sc, 32);
        const bool other_wins = od < bd || (od == bd && oc < bc);
        const int loser = other_wins ? bd : od;
        if (other_wins) {
            bd = od;
            bc = oc;
        }
        sc = min(min(sc, os), loser);
        if (half == 0) {
            const int grow = (qtile + rb) * 32 + lane5;
            const int na = pw.nrmA[grow] >> 8;
            const size_t o = pw.row_off + (size_t)it.chunk * pw.nA_pad + grow;
            rr_d1[o] = bd >= kDistPad ? INT32_MAX : bd + na;
            rr_j1[o] = bc;
            rr_d2[o] = !SECOND || sc >= kDistPad ? INT32_MAX : sc + na;
        }
    }
}

// Pair record and query-pass work items of a one-pair submission, written on the device (the record
// comes as a kernel argument), for callers that may not copy from host memory.
__global__ void sift_plan_single_kernel(SiftPair sp, SiftPair* __restrict__ pair_out, SiftItem* __restrict__ items,
                                        int32_t* __restrict__ zero_pair)
{
    if (threadIdx.x == 0) *pair_out = sp;
    if (zero_pair && threadIdx.x < 2) zero_pair[threadIdx.x] = 0;   // two counters the caller wants cleared: no fill command
    const int n = sp.nstrips * sp.nchunks;
    for (int k = threadIdx.x; k < n; k += blockDim.x)
        items[k] = SiftItem{0, (int16_t)(k / sp.nchunks), (int16_t)(k % sp.nchunks)};
}

void launch_sift_plan_single(const SiftPair& sp, SiftPair* pair_out, SiftItem* items, hipStream_t s, int32_t* zero_pair)
{
    hipLaunchKernelGGL(sift_plan_single_kernel, dim3(1), dim3(256), 0, s, sp, pair_out, items, zero_pair);
}

void launch_sift_query(const SiftPair* pairs, const SiftItem* items, int nitems,
                       int32_t* rr_d1, int32_t* rr_j1, int32_t* rr_d2, bool want_second, hipStream_t s)
{
    static DeviceFlags attr_two, attr_one;   // the attribute is per device and per instantiation
    if (want_second) {
        ensure_dynamic_lds(attr_two, reinterpret_cast<const void*>(sift_query_kernel<true>), kLdsStream);
        hipLaunchKernelGGL(sift_query_kernel<true>, dim3(nitems), dim3(kQueryWaves * 64), kLdsStream, s,
                           pairs, items, rr_d1, rr_j1, rr_d2, -512);
    } else {
        ensure_dynamic_lds(attr_one, reinterpret_cast<const void*>(sift_query_kernel<false>), kLdsStream);
        hipLaunchKernelGGL(sift_query_kernel<false>, dim3(nitems), dim3(kQueryWaves * 64), kLdsStream, s,
                           pairs, items, rr_d1, rr_j1, rr_d2, -512);
    }
}

__device__ __forceinline__ int dot16(const v4i a, const v4i b, int acc)
{
#pragma unroll
    for (int w = 0; w < 4; ++w) acc = __builtin_amdgcn_sdot4(a[w], b[w], acc, false);
    return acc;
}


__device__ __forceinline__ bool ratio_ok(const MatchRule& rule, long long d1, long long d2)
{
    return rule.num2 <= 0 || d1 * rule.den2 < d2 * rule.num2;
}

// minimum over each group of 8 lanes, valid in all 8 (three DPP steps, no LDS round trip)
__device__ __forceinline__ int group8_min(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));   // row_half_mirror: lane k <-> 7 - k
    return v;
}


// ---- 1b. query pass, tile-minimum form ---------------------------------------------------
// The form above spends 2.25 vector instructions per distance (scale + pack, running two smallest), which
// makes the vector issue port, not the matrix pipe, its limit.  This form spends 1.0:
//   * the accumulators start from the train rows' constants cq and the query operand is ~p, so the MFMA
//     result is the half-distance y = floor(nrm(q) / 2) - p.q itself (no per-element arithmetic);
//   * per lane and 32-row tile only the minimum m of the lane's 8 values of a query row is taken (four
//     v_min / v_min3) and packed with the tile's number within the chunk, key = m * 256 + tile (one
//     v_lshl_add), then k2 = med3(k1, k2, key), k1 = min(k1, key): among equal minima the first tile wins, and
//     k2 >> 8 is the second smallest of the minima whatever tile it came from.
// A lane therefore knows the smallest y of its query row (k1 >> 8), the first tile that holds it (k1 & 255), and the
// second smallest of the per-(tile, lane quarter) minima (k2): an upper bound of the runner-up that misses only
// rows of the winner's own 8-row group.  What this leaves open -- which row of the tile, the parity bit
// that y drops, a runner-up inside the winner's group -- is settled for the candidates only, from that one
// tile (sift_resolve_kernel), and in the rare undecidable case by a scan of the whole train set
// (sift_rescan_kernel).  Results are bit-identical to the packed form's.
// The pass runs on v_mfma_i32_16x16x64_i8: on the same loop the chip sustains about 10 % more operations per
// second with it than with 32x32x32 (it holds a higher clock), measured on this kernel as 10.8 against 11.9 ms.
// A wave's 128 query rows are 8 column blocks of 16 (lane l: column l & 15); a 32-row tile is 2 row blocks
// of 16, K = 128 two steps of 64 (lane quarter q = l >> 4 holds k = 64 c + 16 q .. + 15 of both operands,
// and accumulator registers 0..3 hold rows 4 q .. 4 q + 3 of the row block).  Operands are read out of the
// same fragment-major arrays: bytes k = 32 s + 16 h .. + 15 of row r of a tile lie at s * 1024 + (h * 32 + r) * 16.
template <bool FUSED>
__global__ __launch_bounds__(kQueryWaves * 64, 2) __attribute__((amdgpu_waves_per_eu(2, 2)))
void sift_query_tile_kernel(const SiftPair* __restrict__ pairs, const SiftItem* __restrict__ items,
                              int32_t* __restrict__ rr_y1, int32_t* __restrict__ rr_tile,
                              int32_t* __restrict__ rr_k2, MatchRule rule, SiftCandidates cand,
                              int32_t* __restrict__ seg_counts)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SiftItem it = items[xcd_contiguous_item(blockIdx.x, gridDim.x)];
    const SiftPair pw = pairs[it.pair];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, quad = lane >> 4;

    const int c0 = it.chunk * pw.chunk_cols;
    const int ncols = min(pw.chunk_cols, pw.nB_pad - c0);
    const int nsteps = ncols / kStepRows;

    // byte offset, inside a tile, of this lane's 16 operand bytes of k-step c (0, 1) and row block rb (0, 1)
    const int lane_off = ((quad >> 1) * 1024) + ((quad & 1) * 32 + l16) * 16;
    constexpr int kCB = 8;
    const int qtile = (it.strip * (kStripRows / 32)) + wave * kQueryRB;
    v4i pf[kCB][2];
#pragma unroll
    for (int cb = 0; cb < kCB; ++cb) {
        const int8_t* src = pw.fragA + (size_t)(qtile + (cb >> 1)) * kTileBytes + lane_off + (cb & 1) * 256;
#pragma unroll
        for (int c = 0; c < 2; ++c) pf[cb][c] = ~*reinterpret_cast<const v4i*>(src + c * 2048);
    }

    const int8_t* Bsrc = pw.fragB + (size_t)(c0 >> 5) * kTileBytes + wave * (kLdsTile / kQueryWaves) + lane * 16;
    const int32_t* Csrc = pw.nrmB + pw.nB_pad + c0 + (wave & 3) * 64 + lane;

    static_assert(kChunkCols / 32 <= 256, "the tile within a chunk must fit the low byte of a key");
    constexpr int kKeyPad = kYPad * 256;
    int k1[kCB], k2[kCB];   // packed keys: y * 256 + tile within the chunk
#pragma unroll
    for (int cb = 0; cb < kCB; ++cb) { k1[cb] = kKeyPad; k2[cb] = kKeyPad; }

    // Software pipeline in program order, pinned by scheduling barriers (nothing moves across one):
    //   * column blocks go in pairs: eight MFMAs (first k-step of both row blocks of both column blocks, then
    //     the second k-step), so an accumulator is read again four MFMAs after it was written;
    //   * behind every MFMA go two instructions of the epilogue of the PREVIOUS pair, whose accumulators were
    //     finished at least four MFMAs ago, and, with the first six MFMAs of a tile, one LDS read of the
    //     operands and constants of the next tile.
    // zp of pair 3 starts as "nothing": its first (pending) epilogue changes no state.
    v4i z[kCB][2];
    z[6][0] = z[6][1] = z[7][0] = z[7][1] = v4i{kYPad, kYPad, kYPad, kYPad};
    int pend_tile = 0;
    int mm[2];   // running minima of the pair whose epilogue is in progress

    auto epi = [&](int stage, int cb0, int tileno) {
        // the empty volatile statements keep each piece where it is written: being pure arithmetic whose
        // results are needed only at the end, the pieces would otherwise all sink behind the step barrier
        if (stage < 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cb = cb0 + u;
                if (stage == 0) mm[u] = min(min(z[cb][0][0], z[cb][0][1]), min(z[cb][0][2], z[cb][0][3]));
                else mm[u] = min(min(mm[u], z[cb][1][0]), min(min(z[cb][1][1], z[cb][1][2]), z[cb][1][3]));
                asm volatile("" : "+v"(mm[u]));
            }
        } else {
            // one column block per piece: three instructions
            const int u = stage - 2, cb = cb0 + u;
            int tn = tileno;
            asm volatile("" : "+s"(tn));   // one scalar register: the packing is then a single v_lshl_add_u32
            const int key = mm[u] * 256 + tn, b1 = k1[cb];
            k2[cb] = max(min(b1, k2[cb]), min(max(b1, k2[cb]), key));   // v_med3_i32: the runner-up of {k1, k2, key}
            k1[cb] = min(b1, key);                                       // equal minima: the lower tile number stays
            asm volatile("" : "+v"(k2[cb]), "+v"(k1[cb]));
        }
    };

    stage_step<kQueryWaves>(smem, 0, Bsrc, Csrc, wave);
    __syncthreads();
    v4i a[2][2][2], cv[2][2];   // [tile parity][row block][k-step]
    {
        const char* tile = smem + lane_off;
        const int* ctile = reinterpret_cast<const int*>(smem + 2 * kLdsTile) + quad * 4;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            a[0][rb][0] = *reinterpret_cast<const v4i*>(tile + rb * 256);
            a[0][rb][1] = *reinterpret_cast<const v4i*>(tile + rb * 256 + 2048);
            cv[0][rb] = *reinterpret_cast<const v4i*>(ctile + rb * 16);
        }
    }

    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        if (st + 1 < nsteps)
            stage_step<kQueryWaves>(smem, cur ^ 1, Bsrc + (size_t)(st + 1) * kLdsTile, Csrc + (st + 1) * kStepRows, wave);
        const char* Bt = smem + cur * kLdsTile + lane_off;
        const int* nl = reinterpret_cast<const int*>(smem + 2 * kLdsTile + cur * kLdsNorm) + quad * 4;
        const int tile0 = st * kTilesPerStep;   // tile numbers count from the chunk's first tile
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < kTilesPerStep; ++t) {
            const int p = t & 1, q = p ^ 1;
            // where tile t + 1 lies; after the last tile of a step, the first tile of the next step's buffer is
            // read behind the barrier below instead
            const char* nt = Bt + (t + 1) * 4096;
            const int* nc = nl + (t + 1) * 32;
            const bool pre = t + 1 < kTilesPerStep;
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const int cb = 2 * pr;
                const int ecb = pr == 0 ? 6 : cb - 2;
                const int etile = pr == 0 ? (t == 0 ? pend_tile : tile0 + t - 1) : tile0 + t;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int u = (i >> 1) & 1, rb = i & 1, ks = i >> 2;
                    z[cb + u][rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[p][rb][ks], pf[cb + u][ks],
                                                                          ks == 0 ? cv[p][rb] : z[cb + u][rb], 0, 0, 0);
                    if (pr == 0 && pre) {
                        // behind the MFMA, not in front of it: the wait for this tile's operands in front of its
                        // first MFMA would otherwise also wait for the read just issued
                        __builtin_amdgcn_sched_barrier(0);
                        if (i == 0) a[q][0][0] = *reinterpret_cast<const v4i*>(nt);
                        if (i == 1) a[q][1][0] = *reinterpret_cast<const v4i*>(nt + 256);
                        if (i == 2) a[q][0][1] = *reinterpret_cast<const v4i*>(nt + 2048);
                        if (i == 3) a[q][1][1] = *reinterpret_cast<const v4i*>(nt + 256 + 2048);
                        if (i == 4) cv[q][0] = *reinterpret_cast<const v4i*>(nc);
                        if (i == 5) cv[q][1] = *reinterpret_cast<const v4i*>(nc + 16);
                    }
                    if ((i & 1) == 1) epi(i >> 1, ecb, etile);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        pend_tile = tile0 + kTilesPerStep - 1;
        __syncthreads();
        // the next step's first tile, now that the barrier has made it visible
        if (st + 1 < nsteps) {
            const char* tile = smem + (cur ^ 1) * kLdsTile + lane_off;
            const int* ctile = reinterpret_cast<const int*>(smem + 2 * kLdsTile + (cur ^ 1) * kLdsNorm) + quad * 4;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                a[0][rb][0] = *reinterpret_cast<const v4i*>(tile + rb * 256);
                a[0][rb][1] = *reinterpret_cast<const v4i*>(tile + rb * 256 + 2048);
                cv[0][rb] = *reinterpret_cast<const v4i*>(ctile + rb * 16);
            }
        }
    }
#pragma unroll
    for (int stage = 0; stage < 4; ++stage) epi(stage, 6, pend_tile);

    // FUSED (pairs whose train set is one chunk): the select and resolve passes of this strip's 512 rows run here,
    // while the train image the strip has just streamed is still in the XCD's L2.  Candidates go to the strip's own
    // segment of the pair's candidate arrays (out0_off + strip * 512 + k, ascending rows); a gather pass closes the
    // gaps between the segments afterwards.
    int4* s_cand = reinterpret_cast<int4*>(smem);                       // [4 waves][128]: (row, group, y1, y2)
    int* s_cnt = reinterpret_cast<int*>(smem + 4 * 128 * sizeof(int4));  // [4]
    int wcount = 0;
#pragma unroll
    for (int cb = 0; cb < kCB; ++cb) {
        // the four lane quarters saw disjoint 8-row groups of every tile for the same query row; what goes on to
        // the resolve pass is the winner's GROUP: tile * 4 + lane quarter (rows 4 q .. 4 q + 3 of both row blocks)
        int y1 = k1[cb] >> 8, y2 = k2[cb] >> 8, tl = ((c0 >> 5) + (k1[cb] & 255)) * 4 + quad;
#pragma unroll
        for (int d = 16; d <= 32; d <<= 1) {
            const int o1 = __shfl_xor(y1, d);
            const int o2 = __shfl_xor(y2, d);
            const int ot = __shfl_xor(tl, d);
            const bool other = o1 < y1 || (o1 == y1 && ot < tl);
            y2 = min(max(y1, o1), min(y2, o2));
            tl = other ? ot : tl;
            y1 = min(y1, o1);
        }
        const int grow = qtile * 32 + cb * 16 + l16;
        if (quad == 0) {
            const size_t o = pw.row_off + (size_t)it.chunk * pw.nA_pad + grow;
            rr_y1[o] = y1;
            if (!FUSED) {
                rr_tile[o] = tl;
                rr_k2[o] = y2;
            }
        }
        if (FUSED) {
            // the select pass's optimistic tests (d1 at its lowest, d2 at its highest)
            bool ok = quad == 0 && grow < pw.nA && y1 < kYPad;
            if (ok) {
                const long long na = pw.nrmA[grow] >> 8;
                const long long d1lo = 2ll * y1 + na;
                if (rule.max_dist >= 0 && d1lo > rule.max_dist) ok = false;
                if (ok && rule.num2 > 0 && y2 < kYPad) {
                    const long long d2hi = 2ll * y2 + 1 + na;
                    if (!(d1lo * rule.den2 < d2hi * rule.num2)) ok = false;
                }
            }
            const unsigned long long mask = __ballot(ok);
            if (ok) s_cand[wave * 128 + wcount + __popcll(mask & ((1ull << lane) - 1ull))] = make_int4(grow, tl, y1, y2);
            wcount += __popcll(mask);
        }
    }
    if (FUSED) {
        if (lane == 0) s_cnt[wave] = wcount;
        __syncthreads();
        int pre[kQueryWaves + 1];
        pre[0] = 0;
#pragma unroll
        for (int v = 0; v < kQueryWaves; ++v) pre[v + 1] = pre[v] + s_cnt[v];
        const int total = pre[kQueryWaves];
        const int k8 = lane & 7, grp = lane >> 3;
        const int8_t VSM_GLB* fragA = (const int8_t VSM_GLB*)pw.fragA;
        const int8_t VSM_GLB* fragB = (const int8_t VSM_GLB*)pw.fragB;
        const int32_t VSM_GLB* nrmB = (const int32_t VSM_GLB*)pw.nrmB;
        const int seg0 = it.strip * kStripRows;
        // Eight lanes per candidate, as in sift_resolve_kernel.  Two batches of eight candidates per wave are in flight:
        // the loads of both are issued before either is reduced (with every row of a strip a candidate -- rules
        // without a test -- the tail otherwise waits for one round trip to memory per batch, sixteen in a row).
        struct Batch {
            int c, i, g, y1, y2, nrmb;
            bool have;
            long long na;
            v4i q[8], t[8];
        };
        auto issue = [&](int c0c, Batch& b) {
            b.c = c0c + grp;
            b.have = b.c < total;
            const int cc = b.have ? b.c : wave * 8;   // (a candidate that exists: wave * 8 < total whenever the loop runs)
            int v = 0;
#pragma unroll
            for (int u = 1; u < kQueryWaves; ++u) v += cc >= pre[u] ? 1 : 0;
            const int4 rec = s_cand[v * 128 + (cc - pre[v])];
            b.i = rec.x; b.g = rec.y; b.y1 = rec.z; b.y2 = rec.w;
            const int tl = b.g >> 2, qd = b.g & 3;
            const int r = (k8 >> 2) * 16 + qd * 4 + (k8 & 3);     // row of the tile
            const int8_t VSM_GLB* q = fragA + (size_t)(b.i >> 5) * kTileBytes + (b.i & 31) * 16;
            const int8_t VSM_GLB* t = fragB + (size_t)tl * kTileBytes + r * 16;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    b.q[2 * s + h] = *reinterpret_cast<const v4i VSM_GLB*>(q + s * 1024 + h * 512);
                    b.t[2 * s + h] = *reinterpret_cast<const v4i VSM_GLB*>(t + s * 1024 + h * 512);
                }
            }
            const int row = tl * 32 + r;
            b.nrmb = row < pw.nB ? nrmB[row] >> 8 : INT32_MAX;
            b.na = pw.nrmA[b.i] >> 8;
        };
        auto settle = [&](const Batch& b) {
            const int i = b.i, y1 = b.y1, y2 = b.y2;
            const long long na = b.na;
            const int tl = b.g >> 2, qd = b.g & 3;
            int dot = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) dot = dot16(b.q[u], b.t[u], dot);
            const int key = b.nrmb != INT32_MAX ? (b.nrmb - 2 * dot) * 8 + k8 : INT32_MAX;
            const int best = group8_min(key);
            const int second = group8_min(k8 == (best & 7) ? INT32_MAX : key);
            const int kb = best & 7, x1 = best >> 3;
            const long long d1 = (long long)x1 + na;
            bool amb = y2 == y1 || (x1 >> 1) != y1;
            bool pass = false;
            if (!amb) {
                const bool has_in = second != INT32_MAX, has_out = y2 < kYPad;
                bool pr = true;
                if (has_in || has_out) {
                    const long long in_x = has_in ? (long long)(second >> 3) : INT64_MAX / 4;
                    const long long out_lo = has_out ? 2ll * y2 : INT64_MAX / 4;
                    const bool pa = ratio_ok(rule, d1, min(in_x, out_lo) + na);
                    const bool pb = ratio_ok(rule, d1, min(in_x, out_lo + 1) + na);
                    if (pa != pb) amb = true;
                    pr = pa;
                }
                pass = pr && (rule.max_dist < 0 || d1 <= rule.max_dist);
            }
            if (b.have && k8 == 0) {
                const int cpos = seg0 + b.c;
                const size_t o = pw.out0_off + cpos;
                cand.i0[o] = i;
                cand.vbest[o] = ~0ull;
                if (amb) {
                    const int e = atomicAdd(&cand.rescan_cnt[it.pair], 1);
                    cand.rescan_list[pw.out0_off + e] = cpos;
                    cand.j1[o] = -1;
                    cand.d1[o] = 0;
                } else {
                    cand.j1[o] = pass ? tl * 32 + (kb >> 2) * 16 + qd * 4 + (kb & 3) : -1;
                    cand.d1[o] = (int)d1;
                    if (pass && cand.dmax) atomicMax(&cand.dmax[it.pair], (int)d1);
                }
            }
        };
        for (int c0c = wave * 8; c0c < total; c0c += 2 * kQueryWaves * 8) {
            Batch b0, b1;
            issue(c0c, b0);
            issue(c0c + kQueryWaves * 8, b1);
            settle(b0);
            settle(b1);
        }
        if (tid == 0) seg_counts[it.pair * (kChunkCols / kStripRows) + it.strip] = total;
    }
}

// Closes the gaps between the strips' candidate segments of the fused query pass: one workgroup per pair moves the
// segments to the front in strip order (a segment only ever moves towards lower positions, and the segments before it
// have moved already) and leaves the pair's candidate count.  Candidates the resolve or rescan pass rejected (no row)
// are dropped on the way: the mutual check then gathers train rows and runs matrix blocks for the accepted ones
// only (on data full of duplicates every row is a candidate at first and nearly none is accepted).
__global__ __launch_bounds__(kStripRows)
void sift_gather_segments_kernel(const SiftPair* __restrict__ pairs, const int32_t* __restrict__ seg_counts,
                                 SiftCandidates cand)
{
    __shared__ int wsum[kStripRows / 64];
    const SiftPair pw = pairs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int base = 0;
    const int nstrips = pw.nchunks > 0 ? pw.nstrips : 0;   // a pair with an empty set has no work items: nothing wrote its counts
    for (int sidx = 0; sidx < nstrips; ++sidx) {
        const int n = seg_counts[blockIdx.x * (kChunkCols / kStripRows) + sidx];
        const size_t src = pw.out0_off + (size_t)sidx * kStripRows + tid;
        int i = 0, j = -1, d = 0;
        if (tid < n) {
            i = cand.i0[src];
            j = cand.j1[src];
            d = cand.d1[src];
        }
        const bool keep = j >= 0;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();   // every read of this segment is done, and the wave counts are in place
        int before = __popcll(bal & ((1ull << lane) - 1ull)), total = 0;
#pragma unroll
        for (int w = 0; w < kStripRows / 64; ++w) {
            const int c = wsum[w];
            before += w < wave ? c : 0;
            total += c;
        }
        if (keep) {
            const size_t dst = pw.out0_off + base + before;
            cand.i0[dst] = i;
            cand.j1[dst] = j;
            cand.d1[dst] = d;
            cand.vbest[dst] = ~0ull;
        }
        __syncthreads();
        base += total;
    }
    if (tid == 0) cand.count[blockIdx.x] = base;
}

void launch_sift_query_tile(const SiftPair* pairs, const SiftItem* items, int nitems,
                            int32_t* rr_y1, int32_t* rr_tile, int32_t* rr_k2, hipStream_t s)
{
    static DeviceFlags attr_set;   // the attribute is per device
    ensure_dynamic_lds(attr_set, reinterpret_cast<const void*>(sift_query_tile_kernel<false>), kLdsStream);
    hipLaunchKernelGGL(sift_query_tile_kernel<false>, dim3(nitems), dim3(kQueryWaves * 64), kLdsStream, s,
                       pairs, items, rr_y1, rr_tile, rr_k2, MatchRule{}, SiftCandidates{}, nullptr);
}

// ---- 2b. select, tile-minimum form --------------------------------------------------------
// Chunk merge and OPTIMISTIC tests: the nearest row's exact distance is 2 y1 + (0 or 1) + nrm(a), the
// runner-up's is at most 2 k2 + 1 + nrm(a), so a row that fails with d1 at its lowest and d2 at its highest
// fails for certain; everything else becomes a candidate for the resolve pass.
//
// The chunk merge of one row: chunks in ascending order, the first of equal minima keeps its tile.
__device__ __forceinline__ void merge_chunk_row(int e1, int et, int e2, int& y1, int& tl, int& y2)
{
    if (e1 < y1) {       // a tie between chunks leaves y2 == y1: the resolve pass rescans
        y2 = min(y1, e2);
        y1 = e1;
        tl = et;
    } else {
        y2 = min(y2, e1);
    }
}

// Pairs cut into several chunks (few pairs of many rows): the select pass has one workgroup per pair, whose
// walk over chunks x rows is then a long chain of dependent loads.  This pass merges the chunks with the whole
// grid, four rows per thread, and leaves the result in chunk 0's records; the select pass then reads those only.
__global__ __launch_bounds__(256)
void sift_merge_chunks_kernel(const SiftPair* __restrict__ pairs, int32_t* __restrict__ rr_y1,
                              int32_t* __restrict__ rr_tile, int32_t* __restrict__ rr_k2)
{
    const SiftPair pw = pairs[blockIdx.y];
    const int ib = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (ib >= pw.nA || pw.nchunks < 2) return;
    int4 y1 = make_int4(kYPad, kYPad, kYPad, kYPad), y2 = y1, tl = make_int4(0, 0, 0, 0);
    for (int c = 0; c < pw.nchunks; ++c) {
        const size_t o = pw.row_off + (size_t)c * pw.nA_pad + ib;
        const int4 v1 = *reinterpret_cast<const int4*>(rr_y1 + o);
        const int4 vt = *reinterpret_cast<const int4*>(rr_tile + o);
        const int4 v2 = *reinterpret_cast<const int4*>(rr_k2 + o);
        merge_chunk_row(v1.x, vt.x, v2.x, y1.x, tl.x, y2.x);
        merge_chunk_row(v1.y, vt.y, v2.y, y1.y, tl.y, y2.y);
        merge_chunk_row(v1.z, vt.z, v2.z, y1.z, tl.z, y2.z);
        merge_chunk_row(v1.w, vt.w, v2.w, y1.w, tl.w, y2.w);
    }
    const size_t o = pw.row_off + ib;
    *reinterpret_cast<int4*>(rr_y1 + o) = y1;
    *reinterpret_cast<int4*>(rr_tile + o) = tl;
    *reinterpret_cast<int4*>(rr_k2 + o) = y2;
}

// One workgroup per (pair, slice of its rows); slices exist so that a pair of many rows is not walked by a single
// workgroup (65536 rows: 63 us on one CU).  Candidates must come out in ascending row order, so a sliced launch
// runs twice: COUNT = true leaves the number of candidates of every slice in slice_counts[pair * slices + slice],
// COUNT = false starts each slice's output behind the slices before it.  With one slice per pair only the second
// form runs.  Slice s of a pair holds rows [s * span, (s + 1) * span), span = the pair's rows over the slices,
// rounded up to whole rounds of the workgroup.
template <bool COUNT>
__global__ __launch_bounds__(1024)
void sift_select_tile_kernel(const SiftPair* __restrict__ pairs,
                             int32_t* __restrict__ rr_y1, const int32_t* __restrict__ rr_tile,
                             const int32_t* __restrict__ rr_k2, int merged, MatchRule rule, SiftCandidates cand,
                             int32_t* __restrict__ slice_counts)
{
    __shared__ int wsum[16];
    __shared__ int base_s;
    constexpr int R = 8;
    const int pair = blockIdx.y, sl = blockIdx.x, nsl = gridDim.x;
    const SiftPair pw = pairs[pair];
    const int tid = threadIdx.x;
    const int span = ((pw.nA + nsl - 1) / nsl + 1024 * R - 1) / (1024 * R) * (1024 * R);
    const int lo = min(sl * span, pw.nA), hi = min(lo + span, pw.nA);
    if (tid == 0) {
        int base = 0;
        if (!COUNT)
            for (int k = 0; k < sl; ++k) base += slice_counts[pair * nsl + k];
        base_s = base;
        if (!COUNT && sl == 0) {
            if (cand.dmax) cand.dmax[pair] = INT32_MIN;
            cand.rescan_cnt[pair] = 0;
        }
    }
    __syncthreads();

    for (int i0 = lo; i0 < hi; i0 += 1024 * R) {
        const int ib = i0 + tid * R;
        int tl[R], y1[R], y2[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { tl[r] = 0; y1[r] = kYPad; y2[r] = kYPad; }
        if (ib < pw.nA) {
            const int nchunks = merged ? min(pw.nchunks, 1) : pw.nchunks;
            for (int c = 0; c < nchunks; ++c) {
                const size_t o = pw.row_off + (size_t)c * pw.nA_pad + ib;
                int e1[R], et[R], e2[R];
#pragma unroll
                for (int q = 0; q < R / 4; ++q) {
                    const int4 v1 = *reinterpret_cast<const int4*>(rr_y1 + o + 4 * q);
                    const int4 vt = *reinterpret_cast<const int4*>(rr_tile + o + 4 * q);
                    const int4 v2 = *reinterpret_cast<const int4*>(rr_k2 + o + 4 * q);
                    e1[4 * q] = v1.x; e1[4 * q + 1] = v1.y; e1[4 * q + 2] = v1.z; e1[4 * q + 3] = v1.w;
                    et[4 * q] = vt.x; et[4 * q + 1] = vt.y; et[4 * q + 2] = vt.z; et[4 * q + 3] = vt.w;
                    e2[4 * q] = v2.x; e2[4 * q + 1] = v2.y; e2[4 * q + 2] = v2.z; e2[4 * q + 3] = v2.w;
                }
#pragma unroll
                for (int r = 0; r < R; ++r) merge_chunk_row(e1[r], et[r], e2[r], y1[r], tl[r], y2[r]);
            }
            if (!COUNT && nchunks > 1) {
                // the merged half-distances stay in chunk 0's records: sift_verify_rows_kernel reads them
                // (not in the counting launch: the second launch merges the same records again)
                int32_t* dst = rr_y1 + pw.row_off + ib;
                *reinterpret_cast<int4*>(dst) = make_int4(y1[0], y1[1], y1[2], y1[3]);
                *reinterpret_cast<int4*>(dst + 4) = make_int4(y1[4], y1[5], y1[6], y1[7]);
            }
        }
        unsigned okbits = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = ib + r;
            if (i >= pw.nA) continue;
            bool ok = y1[r] < kYPad;
            const long long na = pw.nrmA[i] >> 8;
            const long long d1lo = 2ll * y1[r] + na;
            if (ok && rule.max_dist >= 0 && d1lo > rule.max_dist) ok = false;
            if (ok && rule.num2 > 0 && y2[r] < kYPad) {
                const long long d2hi = 2ll * y2[r] + 1 + na;
                if (!(d1lo * rule.den2 < d2hi * rule.num2)) ok = false;
            }
            okbits |= (ok ? 1u : 0u) << r;
        }
        int slot = block_compact_range(__popc(okbits), wsum, &base_s);
        if (COUNT) continue;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((okbits >> r) & 1u) {
                const size_t o = pw.out0_off + slot++;
                cand.i0[o] = ib + r;
                cand.j1[o] = tl[r];
                cand.d1[o] = y1[r];
                cand.aux[o] = y2[r];
                cand.vbest[o] = ~0ull;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (COUNT) slice_counts[pair * nsl + sl] = base_s;
        else if (sl == nsl - 1) cand.count[pair] = base_s;
    }
}

int sift_select_slices(int max_rows) { return std::max(1, std::min(kMaxRowSlices, (max_rows + 8191) / 8192)); }
int sift_verify_rows_slices(int max_rows) { return sift_select_slices(max_rows); }   // 8192 rows: two rounds of a workgroup

void launch_sift_select_tile(const SiftPair* pairs, int npairs, int max_rows, int max_chunks, int32_t* rr_y1,
                             int32_t* rr_tile, int32_t* rr_k2, MatchRule rule, SiftCandidates cand,
                             int32_t* slice_counts, hipStream_t s)
{
    // a merged record of a one-chunk pair is the record itself; a first tie between chunks that the merge
    // turned into y2 == y1 stays one when the select pass reads the merged record
    const bool merge = max_chunks > 1 && max_rows >= 4096;
    if (merge)
        hipLaunchKernelGGL(sift_merge_chunks_kernel, dim3((max_rows + 1023) / 1024, npairs), dim3(256), 0, s,
                           pairs, rr_y1, rr_tile, rr_k2);
    const dim3 grid(sift_select_slices(max_rows), npairs);
    if (grid.x > 1)
        hipLaunchKernelGGL(sift_select_tile_kernel<true>, grid, dim3(1024), 0, s,
                           pairs, rr_y1, rr_tile, rr_k2, merge ? 1 : 0, rule, cand, slice_counts);
    hipLaunchKernelGGL(sift_select_tile_kernel<false>, grid, dim3(1024), 0, s,
                       pairs, rr_y1, rr_tile, rr_k2, merge ? 1 : 0, rule, cand, slice_counts);
}

// ---- 2c. resolve / rescan ------------------------------------------------------------------
// Eight lanes per candidate, lane k = row k of the winner's 8-row group (rows 4 q .. 4 q + 3 of both 16-row
// blocks of its tile), the whole 128-byte dot product per lane: the 8 exact distances give the nearest row
// (lowest row among equal distances), the exact d1 and the runner-up inside the group; every row outside the
// group has y >= y2 with equality somewhere, i.e. the runner-up outside is 2 y2 or 2 y2 + 1.  The tests are
// decided unless that parity bit matters or the minimum is tied with another group (y2 == y1): those
// candidates go to the rescan pass.  Candidate records are read 64 at a time (lane k: the k-th of the batch).
constexpr int kResolveWaves = 4;
__global__ __launch_bounds__(kResolveWaves * 64)
void sift_resolve_kernel(const SiftPair* __restrict__ pairs, int slices, MatchRule rule, SiftCandidates cand)
{
    const int item = xcd_contiguous_item(blockIdx.x, gridDim.x);
    const int p = item / slices, sl = item % slices;
    const SiftPair pw = pairs[p];
    const int ncand = cand.count[p];
    const int lane = threadIdx.x & 63, k8 = lane & 7, grp = lane >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stride = slices * kResolveWaves;
    const int8_t VSM_GLB* fragA = (const int8_t VSM_GLB*)pw.fragA;
    const int8_t VSM_GLB* fragB = (const int8_t VSM_GLB*)pw.fragB;
    const int32_t VSM_GLB* nrmA = (const int32_t VSM_GLB*)pw.nrmA;
    const int32_t VSM_GLB* nrmB = (const int32_t VSM_GLB*)pw.nrmB;
    // a wave takes candidates first, first + stride, ...
    const int first = sl * kResolveWaves + wave;
    for (int base = first; base < ncand; base += 64 * stride) {
        const int cl = base + lane * stride;
        const bool live = cl < ncand;
        const size_t ol = pw.out0_off + (live ? cl : base);
        const int rec_i = cand.i0[ol], rec_g = cand.j1[ol], rec_y1 = cand.d1[ol], rec_y2 = cand.aux[ol];
        const int rec_na = nrmA[rec_i] >> 8;
        const int nb = min(64, (ncand - base + stride - 1) / stride);   // candidates in this batch
        for (int k0 = 0; k0 < nb; k0 += 8) {
            const int k = k0 + grp;              // this group's candidate of the batch
            const bool have = k < nb;
            const int src = have ? k : k0;
            const int i = __shfl(rec_i, src), g = __shfl(rec_g, src);
            const int y1 = __shfl(rec_y1, src), y2 = __shfl(rec_y2, src);
            const long long na = __shfl(rec_na, src);
            const int tl = g >> 2, quad = g & 3;
            const int r = (k8 >> 2) * 16 + quad * 4 + (k8 & 3);     // row of the tile
            const int8_t VSM_GLB* q = fragA + (size_t)(i >> 5) * kTileBytes + (i & 31) * 16;
            const int8_t VSM_GLB* t = fragB + (size_t)tl * kTileBytes + r * 16;
            int dot = 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    dot = dot16(*reinterpret_cast<const v4i VSM_GLB*>(q + s * 1024 + h * 512),
                                *reinterpret_cast<const v4i VSM_GLB*>(t + s * 1024 + h * 512), dot);
            }
            const int row = tl * 32 + r;
            // |x| < 2^23: value and position in the group in one key (positions ascend with the rows)
            const int key = row < pw.nB ? ((nrmB[row] >> 8) - 2 * dot) * 8 + k8 : INT32_MAX;
            const int best = group8_min(key);
            const int second = group8_min(k8 == (best & 7) ? INT32_MAX : key);
            const int kb = best & 7, x1 = best >> 3;
            const long long d1 = (long long)x1 + na;
            bool amb = y2 == y1 || (x1 >> 1) != y1;   // minimum tied with another group: the nearest row may lie there
            bool pass = false;
            if (!amb) {
                const bool has_in = second != INT32_MAX, has_out = y2 < kYPad;
                bool pr = true;
                if (has_in || has_out) {
                    const long long in_x = has_in ? (long long)(second >> 3) : INT64_MAX / 4;
                    const long long out_lo = has_out ? 2ll * y2 : INT64_MAX / 4;
                    const bool pa = ratio_ok(rule, d1, min(in_x, out_lo) + na);
                    const bool pb = ratio_ok(rule, d1, min(in_x, out_lo + 1) + na);
                    if (pa != pb) amb = true;   // the parity bit of the runner-up decides
                    pr = pa;
                }
                pass = pr && (rule.max_dist < 0 || d1 <= rule.max_dist);
            }
            if (have && k8 == 0) {
                const int c = base + k * stride;
                const size_t o = pw.out0_off + c;
                if (amb) {
                    const int e = atomicAdd(&cand.rescan_cnt[p], 1);
                    cand.rescan_list[pw.out0_off + e] = c;
                    cand.j1[o] = -1;
                } else {
                    cand.j1[o] = pass ? tl * 32 + (kb >> 2) * 16 + quad * 4 + (kb & 3) : -1;
                    cand.d1[o] = (int)d1;
                    if (pass && cand.dmax) atomicMax(&cand.dmax[p], (int)d1);
                }
            }
        }
    }
}

// ---- 2d. rescan ------------------------------------------------------------------------------
// The candidates one group could not settle get the packed form of the query pass, which is exact by itself: their
// rows of image 0 are gathered, 512 per work item (32 per block of a wave, as in the query pass), and the item's
// chunk of the train set streams past them.  On ordinary data a pair has none or a handful of them (one item whose
// waves skip the blocks without rows); on data full of exact duplicates every row of every pair comes here, and the
// pass then costs what the packed query pass costs, which bounds the tile-minimum path at the sum of the two.
constexpr int kRescanGrid = 512;    // two workgroups per CU: the whole grid is resident and walks the item list

// rescan_cnt[pair] -> (pair, block of kStripRows list slots, chunk) items; nitems[0] = their number, nitems[1] = the
// candidates of the whole submission
__global__ __launch_bounds__(1024)
void sift_rescan_plan_kernel(const SiftPair* __restrict__ pairs, int npairs, const int32_t* __restrict__ counts,
                             SiftItem* __restrict__ items, int32_t* __restrict__ nitems)
{
    __shared__ int wsum[16], csum[16];
    __shared__ int carry_s, cand_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { carry_s = 0; cand_s = 0; }
    __syncthreads();
    for (int p0 = 0; p0 < npairs; p0 += 1024) {
        const int p = p0 + tid;
        const int cnt = p < npairs ? counts[p] : 0;
        const int nch = p < npairs ? pairs[p].nchunks : 0;
        const int blocks = (cnt + kStripRows - 1) / kStripRows;
        const int v = blocks * nch;
        int incl = v, cincl = cnt;
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d), oc = __shfl_up(cincl, d);
            if (lane >= d) { incl += o; cincl += oc; }
        }
        if (lane == 63) { wsum[wave] = incl; csum[wave] = cincl; }
        __syncthreads();
        int woff = 0, tot = 0, ctot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) woff += wsum[w];
            tot += wsum[w];
            ctot += csum[w];
        }
        int at = carry_s + woff + incl - v;
        for (int b = 0; b < blocks; ++b)
            for (int ch = 0; ch < nch; ++ch) items[at++] = SiftItem{p, (int16_t)b, (int16_t)ch};
        __syncthreads();
        if (tid == 0) { carry_s += tot; cand_s += ctot; }
        __syncthreads();
    }
    if (tid == 0) {
        nitems[0] = carry_s;
        nitems[1] = cand_s;
    }
}

// The packed query pass (see sift_query_kernel) over gathered rows: list slot k of the pair names candidate
// rescan_list[k], whose row of image 0 is the persistent operand.  Results per (chunk, slot), as the query pass leaves
// them per (chunk, row): exact distance of the nearest row, the row (lowest among equals), exact runner-up distance.
__global__ __launch_bounds__(kQueryWaves * 64, 2)
void sift_rescan_query_kernel(const SiftPair* __restrict__ pairs, SiftCandidates cand, int negscale)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane5 = lane & 31;
    const int half = lane >> 5;
    const int total = cand.rescan_nitems[0];
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
    const SiftItem it = cand.rescan_items[w];
    const SiftPair pw = pairs[it.pair];
    const int cnt = cand.rescan_cnt[it.pair];
    const int c0 = it.chunk * pw.chunk_cols;
    const int ncols = min(pw.chunk_cols, pw.nB_pad - c0);
    const int nsteps = ncols / kStepRows;

    const int slot0 = it.strip * kStripRows + wave * (kQueryRB * 32);
    // blocks of 32 slots this wave has rows in (the same for every lane of the wave)
    const int nlive = __builtin_amdgcn_readfirstlane(min(kQueryRB, max(0, (cnt - slot0 + 31) / 32)));
    int qrow[kQueryRB];
    v4i pf[kQueryRB][4];
#pragma unroll
    for (int rb = 0; rb < kQueryRB; ++rb) {
        const int slot = slot0 + rb * 32 + lane5;
        qrow[rb] = slot < cnt ? cand.i0[pw.out0_off + cand.rescan_list[pw.out0_off + slot]] : 0;
        const int8_t* src = pw.fragA + (size_t)(qrow[rb] >> 5) * kTileBytes + (half * 32 + (qrow[rb] & 31)) * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s) pf[rb][s] = *reinterpret_cast<const v4i*>(src + s * 1024);
    }

    const int8_t* Bsrc = pw.fragB + (size_t)(c0 >> 5) * kTileBytes + wave * (kLdsTile / kQueryWaves) + lane * 16;
    const int32_t* Nsrc = pw.nrmB + c0 + (wave & 3) * 64 + lane;

    int bestd[kQueryRB], bestcol[kQueryRB], second[kQueryRB];
#pragma unroll
    for (int rb = 0; rb < kQueryRB; ++rb) { bestd[rb] = kDistPad; bestcol[rb] = INT32_MAX; second[rb] = kDistPad; }

    stage_step<kQueryWaves>(smem, 0, Bsrc, Nsrc, wave);
    __syncthreads();

    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        if (st + 1 < nsteps)
            stage_step<kQueryWaves>(smem, cur ^ 1, Bsrc + (size_t)(st + 1) * kLdsTile, Nsrc + (st + 1) * kStepRows, wave);
        const char* Bt = smem + cur * kLdsTile + lane * 16;
        const int* nl = reinterpret_cast<const int*>(smem + 2 * kLdsTile + cur * kLdsNorm);

        int k1[kQueryRB], k2[kQueryRB];
#pragma unroll
        for (int rb = 0; rb < kQueryRB; ++rb) { k1[rb] = INT32_MAX; k2[rb] = INT32_MAX; }

        // all four blocks of the wave in use: the instruction stream of the query pass; else only the blocks with rows
        auto tiles = [&](auto full) {
            constexpr bool FULL = decltype(full)::value;
#pragma unroll
            for (int t = 0; t < kTilesPerStep; ++t) {
                const char* tile = Bt + t * 4096;
                const v4i b0 = *reinterpret_cast<const v4i*>(tile);
                const v4i b1 = *reinterpret_cast<const v4i*>(tile + 1024);
                const v4i b2 = *reinterpret_cast<const v4i*>(tile + 2048);
                const v4i b3 = *reinterpret_cast<const v4i*>(tile + 3072);
                int c[16];
                tile_consts(nl + t * 32, half, c);
                v16i acc[kQueryRB];
#pragma unroll
                for (int rb = 0; rb < kQueryRB; ++rb) {
                    if (FULL || rb < nlive) {
                        v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        z = __builtin_amdgcn_mfma_i32_32x32x32_i8(b0, pf[rb][0], z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_i32_32x32x32_i8(b1, pf[rb][1], z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_i32_32x32x32_i8(b2, pf[rb][2], z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_i32_32x32x32_i8(b3, pf[rb][3], z, 0, 0, 0);
                        acc[rb] = z;
                    }
                }
#pragma unroll
                for (int rb = 0; rb < kQueryRB; ++rb) {
                    if (FULL || rb < nlive) {
                        int uprev = 0;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int x0 = __mul24(acc[rb][2 * e], negscale) + c[2 * e];
                            const int x1 = __mul24(acc[rb][2 * e + 1], negscale) + c[2 * e + 1];
                            const int u = med3i(k1[rb], x0, x1);   // runner-up of {k1, x0, x1}
                            k1[rb] = min3i(k1[rb], x0, x1);
                            if (e & 1) k2[rb] = min3i(k2[rb], uprev, u);
                            uprev = u;
                        }
                    }
                }
            }
        };
        if (nlive == kQueryRB) tiles(std::true_type{});
        else if (nlive > 0) tiles(std::false_type{});
#pragma unroll
        for (int rb = 0; rb < kQueryRB; ++rb) {
            // fold the finished 256-row group (strict compare: the earlier group, i.e. the lower row, stays)
            const int d = k1[rb] >> 8;
            const int col = c0 + st * kStepRows + (k1[rb] & 255);
            if (d < bestd[rb]) {
                second[rb] = min(second[rb], bestd[rb]);
                bestd[rb] = d;
                bestcol[rb] = col;
            } else {
                second[rb] = min(second[rb], d);
            }
            second[rb] = min(second[rb], k2[rb] >> 8);
        }
        __syncthreads();
    }

#pragma unroll
    for (int rb = 0; rb < kQueryRB; ++rb) {
        int bd = bestd[rb], bc = bestcol[rb], sc = second[rb];
        const int od = __shfl_xor(bd, 32);
        const int oc = __shfl_xor(bc, 32);
        const int os = __shfl_xor(sc, 32);
        const bool other_wins = od < bd || (od == bd && oc < bc);
        const int loser = other_wins ? bd : od;
        if (other_wins) {
            bd = od;
            bc = oc;
        }
        sc = min(min(sc, os), loser);
        const int slot = slot0 + rb * 32 + lane5;
        if (half == 0 && slot < cnt) {
            const int na = pw.nrmA[qrow[rb]] >> 8;
            const size_t o = pw.row_off + (size_t)it.chunk * pw.nA_pad + slot;
            cand.rs_d1[o] = bd >= kDistPad ? INT32_MAX : bd + na;
            cand.rs_j1[o] = bc;
            cand.rs_d2[o] = sc >= kDistPad ? INT32_MAX : sc + na;
        }
    }
    }   // next work item (the last step's barrier separates it from this one's LDS reads)
}

// Merges the chunks of every rescanned candidate (ascending column ranges: a strict compare keeps the lowest row),
// applies the exact tests and writes the candidate's record.
__global__ __launch_bounds__(kStripRows)
void sift_rescan_finish_kernel(const SiftPair* __restrict__ pairs, MatchRule rule, SiftCandidates cand)
{
    const int total = cand.rescan_nitems[0];
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        const SiftItem it = cand.rescan_items[w];
        if (it.chunk != 0) continue;
        const SiftPair pw = pairs[it.pair];
        const int slot = it.strip * kStripRows + threadIdx.x;
        if (slot >= cand.rescan_cnt[it.pair]) continue;
        int d1 = INT32_MAX, j1 = -1, d2 = INT32_MAX;
        for (int c = 0; c < pw.nchunks; ++c) {
            const size_t o = pw.row_off + (size_t)c * pw.nA_pad + slot;
            const int e1 = cand.rs_d1[o], ej = cand.rs_j1[o], e2 = cand.rs_d2[o];
            if (e1 < d1) {
                d2 = min(d1, e2);
                d1 = e1;
                j1 = ej;
            } else {
                d2 = min(d2, e1);
            }
        }
        if (j1 >= pw.nB) j1 = -1;
        bool pass = j1 >= 0;
        if (pass && d2 != INT32_MAX) pass = ratio_ok(rule, d1, d2);
        if (pass && rule.max_dist >= 0 && (long long)d1 > rule.max_dist) pass = false;
        const size_t o = pw.out0_off + cand.rescan_list[pw.out0_off + slot];
        cand.j1[o] = pass ? j1 : -1;
        cand.d1[o] = d1;
        if (pass && cand.dmax) atomicMax(&cand.dmax[it.pair], d1);
    }
}

void launch_sift_rescan(const SiftPair* pairs, int npairs, int max_items, MatchRule rule, SiftCandidates cand, hipStream_t s)
{
    if (npairs <= 0 || max_items <= 0) return;
    hipLaunchKernelGGL(sift_rescan_plan_kernel, dim3(1), dim3(1024), 0, s, pairs, npairs, cand.rescan_cnt, cand.rescan_items,
                       cand.rescan_nitems);
    static DeviceFlags attr_set;   // the attribute is per device
    ensure_dynamic_lds(attr_set, reinterpret_cast<const void*>(sift_rescan_query_kernel), kLdsStream);
    const int grid = std::min(kRescanGrid, max_items);
    hipLaunchKernelGGL(sift_rescan_query_kernel, dim3(grid), dim3(kQueryWaves * 64), kLdsStream, s, pairs, cand, -512);
    hipLaunchKernelGGL(sift_rescan_finish_kernel, dim3(grid), dim3(kStripRows), 0, s, pairs, rule, cand);
}

// The query pass with the select and resolve passes of every strip in its tail (every pair's sets must be one chunk
// of rows at most); cand.rescan_cnt and cand.dmax are initialised here.  launch_sift_fused_finish follows it: the
// rescan pass over what the tails could not settle, and the gather pass over their segments.
void launch_sift_query_tile_fused(const SiftPair* pairs, int npairs, const SiftItem* items, int nitems,
                                  int32_t* rr_y1, MatchRule rule, SiftCandidates cand, int32_t* seg_counts,
                                  hipStream_t s)
{
    static DeviceFlags attr_set;
    ensure_dynamic_lds(attr_set, reinterpret_cast<const void*>(sift_query_tile_kernel<true>), kLdsStream);
    (void)hipMemsetAsync(cand.rescan_cnt, 0, sizeof(int32_t) * npairs, s);
    if (cand.dmax) (void)hipMemsetAsync(cand.dmax, 0x80, sizeof(int32_t) * npairs, s);   // below every distance
    hipLaunchKernelGGL(sift_query_tile_kernel<true>, dim3(nitems), dim3(kQueryWaves * 64), kLdsStream, s,
                       pairs, items, rr_y1, nullptr, nullptr, rule, cand, seg_counts);
}

void launch_sift_fused_finish(const SiftPair* pairs, int npairs, int max_items, MatchRule rule, SiftCandidates cand,
                              const int32_t* seg_counts, hipStream_t s)
{
    // (the rescan pass names candidates by their position in the strips' segments, which the gather pass then moves)
    launch_sift_rescan(pairs, npairs, max_items, rule, cand, s);
    hipLaunchKernelGGL(sift_gather_segments_kernel, dim3(npairs), dim3(kStripRows), 0, s, pairs, seg_counts, cand);
}

// The image-0 rows that can matter to the mutual check of a pair (see SiftCandidates::dmax), in ascending order.
// Sliced like sift_select_tile_kernel: a counting launch first when a pair's rows are cut into several slices.
template <bool COUNT>
__global__ __launch_bounds__(1024)
void sift_verify_rows_kernel(const SiftPair* __restrict__ pairs, const int32_t* __restrict__ rr_y1, SiftCandidates cand,
                             int32_t* __restrict__ slice_counts)
{
    __shared__ int wsum[16];
    __shared__ int base_s;
    constexpr int R = 4;
    const int pair = blockIdx.y, sl = blockIdx.x, nsl = gridDim.x;
    const SiftPair pw = pairs[pair];
    const int tid = threadIdx.x;
    // a pair with an empty train set has no query-pass records at all (its row_off is the next pair's): no rows to look at
    const int nrows = pw.nchunks > 0 ? pw.nA : 0;
    const int span = ((nrows + nsl - 1) / nsl + 1024 * R - 1) / (1024 * R) * (1024 * R);
    const int lo = min(sl * span, nrows), hi = min(lo + span, nrows);
    if (tid == 0) {
        int base = 0;
        if (!COUNT)
            for (int k = 0; k < sl; ++k) base += slice_counts[pair * nsl + k];
        base_s = base;
    }
    __syncthreads();
    const long long dmax = cand.dmax[pair];
    for (int i0 = lo; i0 < hi; i0 += 1024 * R) {
        const int ib = i0 + tid * R;
        unsigned okbits = 0;
        int nn_keep[R] = {0, 0, 0, 0};
        if (ib < pw.nA) {
            const int4 y = *reinterpret_cast<const int4*>(rr_y1 + pw.row_off + ib);
            const int4 na = *reinterpret_cast<const int4*>(pw.nrmA + ib);
            const int yy[R] = {y.x, y.y, y.z, y.w}, nn[R] = {na.x, na.y, na.z, na.w};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                nn_keep[r] = nn[r];
                if (ib + r < pw.nA && yy[r] < kYPad && 2ll * yy[r] + (nn[r] >> 8) <= dmax) okbits |= 1u << r;
            }
        }
        int slot = block_compact_range(__popc(okbits), wsum, &base_s);
        if (COUNT) continue;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if ((okbits >> r) & 1u) {
                // the packed norm the verify pass streams for this position: the position is the row code
                cand.vrows[pw.out0_off + slot] = ib + r;
                cand.vcodes[pw.out0_off + slot] = (nn_keep[r] & ~255) | (slot & 255);
                ++slot;
            }
    }
    __syncthreads();
    if (tid == 0) {
        if (COUNT) slice_counts[pair * nsl + sl] = base_s;
        else if (sl == nsl - 1) cand.nvrows[pair] = base_s;
    }
}

void launch_sift_verify_rows(const SiftPair* pairs, int npairs, int max_rows, const int32_t* rr_y1, SiftCandidates cand,
                             int32_t* slice_counts, hipStream_t s)
{
    if (npairs <= 0) return;
    const dim3 grid(sift_verify_rows_slices(max_rows), npairs);
    if (grid.x > 1)
        hipLaunchKernelGGL(sift_verify_rows_kernel<true>, grid, dim3(1024), 0, s, pairs, rr_y1, cand, slice_counts);
    hipLaunchKernelGGL(sift_verify_rows_kernel<false>, grid, dim3(1024), 0, s, pairs, rr_y1, cand, slice_counts);
}

void launch_sift_resolve(const SiftPair* pairs, int npairs, int max_rows, int max_items, MatchRule rule,
                         SiftCandidates cand, hipStream_t s)
{
    if (npairs <= 0) return;
    // enough waves per pair for its candidates (at most one per row), few enough blocks for short lists
    const int slices = std::max(1, std::min(64, (max_rows + 511) / 512));
    hipLaunchKernelGGL(sift_resolve_kernel, dim3(npairs * slices), dim3(kResolveWaves * 64), 0, s, pairs, slices, rule, cand);
    launch_sift_rescan(pairs, npairs, max_items, rule, cand, s);
}

// ---- 3. verify pass -------------------------------------------------------------------
// Persistent rows = the image-1 rows named by a block of 256 candidates (gathered from the
// operand-order layout: kVerifyRB blocks of 32 per wave, sharing every streamed operand read);
// streamed set = all of image 0.  Per candidate: the nearest image-0 row.
constexpr int kVerifyWaves = 4;
constexpr int kVerifyGrid = 1024;      // workgroups of the verify pass
constexpr int kVerifyMaxSplit = 64;    // most ranges of image-0 steps a block of candidates is cut into
constexpr int kVerifyRB = kVerifyBlock / (32 * kVerifyWaves);

__global__ __launch_bounds__(kVerifyWaves * 64, 2)
void sift_verify_kernel(const SiftPair* __restrict__ pairs, const SiftItem* __restrict__ vitems,
                        const int* __restrict__ nvitems, SiftCandidates cand, int negscale)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane5 = lane & 31;
    const int half = lane >> 5;
    // The item list is built on the device (only blocks that hold candidates); a fixed grid walks it
    // round-robin, so every workgroup leaves after at most ceil(work / gridDim.x) items.  When the
    // list is short (one large pair), every block is also cut into `nsplit` ranges of image-0 steps,
    // and the ranges meet in a 64-bit atomic minimum of (distance, row).
    const int total = nvitems[0], nsplit = nvitems[1];
    for (int w = blockIdx.x; w < total * nsplit; w += gridDim.x) {
    const SiftItem it = vitems[w / nsplit];
    const int part = w % nsplit;
    const SiftPair pw = pairs[it.pair];
    // with a row list the streamed set is the listed rows of image 0, gathered step by step
    const int32_t* __restrict__ vrows = cand.vrows ? cand.vrows + pw.out0_off : nullptr;
    const int nv = vrows ? cand.nvrows[it.pair] : pw.nA_pad;
    const int allsteps = (nv + kStepRows - 1) / kStepRows;
    const int st0 = (int)((long long)allsteps * part / nsplit), st1 = (int)((long long)allsteps * (part + 1) / nsplit);
    if (st0 == st1) continue;
    const int ncand = cand.count[it.pair];
    const int cbase = it.strip * kVerifyBlock;

    int ci[kVerifyRB];
    v4i pf[kVerifyRB][4];
#pragma unroll
    for (int rb = 0; rb < kVerifyRB; ++rb) {
        ci[rb] = cbase + (wave * kVerifyRB + rb) * 32 + lane5;
        const int j = ci[rb] < ncand ? max(cand.j1[pw.out0_off + ci[rb]], 0) : 0;   // -1: rejected by the resolve pass
        const int8_t* src = pw.fragB + (size_t)(j >> 5) * kTileBytes + (half * 32 + (j & 31)) * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s) pf[rb][s] = *reinterpret_cast<const v4i*>(src + s * 1024);
    }

    const int8_t* Asrc = pw.fragA + wave * (kLdsTile / kVerifyWaves) + lane * 16;
    const int32_t* Nsrc = pw.nrmA + (wave & 3) * 64 + lane;

    int bestd[kVerifyRB], bestrow[kVerifyRB];
#pragma unroll
    for (int rb = 0; rb < kVerifyRB; ++rb) { bestd[rb] = kDistPad; bestrow[rb] = INT32_MAX; }

    // One step of listed rows: every lane fetches the 16-byte pieces of its own row of the wave's two tiles out
    // of whichever tile of image 0 holds it (LDS-DMA takes a global address per lane), and the packed norm of
    // the row at its position of the step, with the position as the row code.
    // The list entries a step needs (two rows per lane, one packed norm) are fetched a step before the staging that
    // uses them, so that no staging waits for a load it has just issued.
    const int32_t* __restrict__ vcodes = cand.vcodes ? cand.vcodes + pw.out0_off : nullptr;
    int id_row[2] = {0, 0}, id_code = kNormPad;
    auto fetch_ids = [&](int st) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int pos = st * kStepRows + (wave * 2 + tt) * 32 + lane5;
            id_row[tt] = pos < nv ? vrows[pos] : -1;
        }
        const int pos = st * kStepRows + wave * 64 + lane;
        id_code = pos < nv ? vcodes[pos] : (kNormPad | (pos & 255));
    };
    auto stage_rows = [&](int buf) {
        char* dst = smem + buf * kLdsTile + wave * (kLdsTile / kVerifyWaves);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int row = id_row[tt];
            const int8_t* src = (row >= 0 ? pw.fragA + (size_t)(row >> 5) * kTileBytes + (half * 32 + (row & 31)) * 16
                                          : cand.zero_tile + (half * 32 + lane5) * 16);
#pragma unroll
            for (int s = 0; s < 4; ++s) dma16(src + s * 1024, dst + (tt * 4 + s) * 1024);
        }
        reinterpret_cast<int*>(smem + 2 * kLdsTile + buf * kLdsNorm)[wave * 64 + lane] = id_code;
    };
    auto stage = [&](int buf, int st) {
        if (vrows) stage_rows(buf);
        else stage_step<kVerifyWaves>(smem, buf, Asrc + (size_t)st * kLdsTile, Nsrc + st * kStepRows, wave);
    };

    if (vrows) fetch_ids(st0);
    stage(0, st0);
    if (vrows && st0 + 1 < st1) fetch_ids(st0 + 1);
    __syncthreads();

    for (int st = st0; st < st1; ++st) {
        const int cur = (st - st0) & 1;
        if (st + 1 < st1) {
            stage(cur ^ 1, st + 1);
            if (vrows && st + 2 < st1) fetch_ids(st + 2);
        }
        const char* At = smem + cur * kLdsTile + lane * 16;
        const int* nl = reinterpret_cast<const int*>(smem + 2 * kLdsTile + cur * kLdsNorm);
        int m[kVerifyRB];
#pragma unroll
        for (int rb = 0; rb < kVerifyRB; ++rb) m[rb] = INT32_MAX;
#pragma unroll
        for (int t = 0; t < kTilesPerStep; ++t) {
            const char* tile = At + t * 4096;
            const v4i b0 = *reinterpret_cast<const v4i*>(tile);
            const v4i b1 = *reinterpret_cast<const v4i*>(tile + 1024);
            const v4i b2 = *reinterpret_cast<const v4i*>(tile + 2048);
            const v4i b3 = *reinterpret_cast<const v4i*>(tile + 3072);
            int c[16];
            tile_consts(nl + t * 32, half, c);
            v16i acc[kVerifyRB];
#pragma unroll
            for (int rb = 0; rb < kVerifyRB; ++rb) {
                v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                z = __builtin_amdgcn_mfma_i32_32x32x32_i8(b0, pf[rb][0], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_i32_32x32x32_i8(b1, pf[rb][1], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_i32_32x32x32_i8(b2, pf[rb][2], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_i32_32x32x32_i8(b3, pf[rb][3], z, 0, 0, 0);
                acc[rb] = z;
            }
#pragma unroll
            for (int rb = 0; rb < kVerifyRB; ++rb) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int y0 = __mul24(acc[rb][2 * e], negscale) + c[2 * e];
                    const int y1 = __mul24(acc[rb][2 * e + 1], negscale) + c[2 * e + 1];
                    m[rb] = min3i(m[rb], y0, y1);
                }
            }
        }
#pragma unroll
        for (int rb = 0; rb < kVerifyRB; ++rb) {
            const int d = m[rb] >> 8;
            if (d < bestd[rb]) {
                bestd[rb] = d;
                bestrow[rb] = st * kStepRows + (m[rb] & 255);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int rb = 0; rb < kVerifyRB; ++rb) {
        const int od = __shfl_xor(bestd[rb], 32);
        const int orow = __shfl_xor(bestrow[rb], 32);
        if (od < bestd[rb] || (od == bestd[rb] && orow < bestrow[rb])) {
            bestd[rb] = od;
            bestrow[rb] = orow;
        }
        if (half == 0 && ci[rb] < ncand) {
            // positions of the row list ascend with the rows: the lowest position is the lowest row
            const int row = vrows && bestrow[rb] < nv ? vrows[bestrow[rb]] : bestrow[rb];
            atomicMin(&cand.vbest[pw.out0_off + ci[rb]],
                      ((unsigned long long)((unsigned)bestd[rb] ^ 0x80000000u) << 32) | (unsigned)row);
        }   // sign bit flipped: the distances are relative (the candidate's own norm is left out) and may be negative
    }
    }   // next work item (the last step's barrier separates it from this one's LDS reads)
}

// counts[pair] -> compact list of (pair, block of kVerifyBlock candidates); nvitems[0] = its length,
// nvitems[1] = ranges of image-0 steps every block is cut into (so that a short list still fills the grid)
__global__ __launch_bounds__(1024)
void sift_verify_plan_kernel(const int32_t* __restrict__ counts, int npairs, SiftItem* __restrict__ vitems,
                             int* __restrict__ nvitems)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int p0 = 0; p0 < npairs; p0 += 1024) {
        const int p = p0 + tid;
        const int v = p < npairs ? (counts[p] + kVerifyBlock - 1) / kVerifyBlock : 0;
        int incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) woff += wsum[w];
            tot += wsum[w];
        }
        const int first = carry_s + woff + incl - v;
        for (int b = 0; b < v; ++b) vitems[first + b] = SiftItem{p, (int16_t)b, 0};
        __syncthreads();
        if (tid == 0) carry_s += tot;
        __syncthreads();
    }
    if (tid == 0) {
        const int total = carry_s;
        nvitems[0] = total;
        nvitems[1] = total > 0 && total * 2 <= kVerifyGrid ? min(kVerifyGrid / total, kVerifyMaxSplit) : 1;
    }
}


void launch_sift_verify(const SiftPair* pairs, int npairs, SiftItem* vitems, int max_vitems, int* nvitems,
                        SiftCandidates cand, hipStream_t s)
{
    if (npairs <= 0 || max_vitems <= 0) return;
    hipLaunchKernelGGL(sift_verify_plan_kernel, dim3(1), dim3(1024), 0, s, cand.count, npairs, vitems, nvitems);
    static DeviceFlags attr_set;   // the attribute is per device
    ensure_dynamic_lds(attr_set, reinterpret_cast<const void*>(sift_verify_kernel), kLdsStream);
    hipLaunchKernelGGL(sift_verify_kernel, dim3(kVerifyGrid), dim3(kVerifyWaves * 64), kLdsStream, s,
                       pairs, vitems, nvitems, cand, -512);
}

// ---- 2. select ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void sift_select_kernel(const SiftPair* __restrict__ pairs,
                        const int32_t* __restrict__ rr_d1, const int32_t* __restrict__ rr_j1,
                        const int32_t* __restrict__ rr_d2, MatchRule rule,
                        SiftCandidates cand, SiftOutputs out)
{
    __shared__ int wsum[16];
    __shared__ int base_s;
    const SiftPair pw = pairs[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid == 0) base_s = 0;
    __syncthreads();

    // eight consecutive rows per thread and round: few rounds (and barriers) even for a very large set,
    // and the candidates still come out in ascending row order
    constexpr int R = 8;
    for (int i0 = 0; i0 < pw.nA; i0 += 1024 * R) {
        const int ib = i0 + tid * R;   // nA_pad is a multiple of 512 and so is row_off: 32-byte aligned
        int j1[R], d1[R], d2[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { j1[r] = -1; d1[r] = INT32_MAX; d2[r] = INT32_MAX; }
        if (ib < pw.nA) {
            // merge chunks (ascending column ranges: a strict compare keeps the lowest column)
            for (int c = 0; c < pw.nchunks; ++c) {
                const size_t o = pw.row_off + (size_t)c * pw.nA_pad + ib;
                int e1[R], ej[R], e2[R];
#pragma unroll
                for (int q = 0; q < R / 4; ++q) {
                    const int4 v1 = *reinterpret_cast<const int4*>(rr_d1 + o + 4 * q);
                    const int4 vj = *reinterpret_cast<const int4*>(rr_j1 + o + 4 * q);
                    const int4 v2 = *reinterpret_cast<const int4*>(rr_d2 + o + 4 * q);
                    e1[4 * q] = v1.x; e1[4 * q + 1] = v1.y; e1[4 * q + 2] = v1.z; e1[4 * q + 3] = v1.w;
                    ej[4 * q] = vj.x; ej[4 * q + 1] = vj.y; ej[4 * q + 2] = vj.z; ej[4 * q + 3] = vj.w;
                    e2[4 * q] = v2.x; e2[4 * q + 1] = v2.y; e2[4 * q + 2] = v2.z; e2[4 * q + 3] = v2.w;
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (e1[r] < d1[r]) {
                        d2[r] = min(d1[r], e2[r]);
                        d1[r] = e1[r];
                        j1[r] = ej[r];
                    } else {
                        d2[r] = min(d2[r], e1[r]);
                    }
                }
            }
        }
        unsigned okbits = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = ib + r;
            if (i >= pw.nA) continue;
            if (j1[r] >= pw.nB) j1[r] = -1;
            bool ok = j1[r] >= 0;
            if (out.knn_j1) {
                out.knn_j1[pw.out0_off + i] = j1[r];
                out.knn_d1[pw.out0_off + i] = d1[r];
                out.knn_d2[pw.out0_off + i] = d2[r];
            }
            if (ok && rule.max_dist >= 0 && (long long)d1[r] > rule.max_dist) ok = false;
            if (ok && rule.num2 > 0 && d2[r] != INT32_MAX) {
                if (!((long long)d1[r] * rule.den2 < (long long)d2[r] * rule.num2)) ok = false;
            }
            okbits |= (ok ? 1u : 0u) << r;
        }
        int slot = block_compact_range(__popc(okbits), wsum, &base_s);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((okbits >> r) & 1u) {
                const size_t o = pw.out0_off + slot++;
                cand.i0[o] = ib + r;
                cand.j1[o] = j1[r];
                cand.d1[o] = d1[r];
                cand.vbest[o] = ~0ull;
            }
        }
    }
    __syncthreads();
    if (tid == 0) cand.count[blockIdx.x] = base_s;
}

void launch_sift_select(const SiftPair* pairs, int npairs, const int32_t* rr_d1, const int32_t* rr_j1,
                        const int32_t* rr_d2, MatchRule rule, SiftCandidates cand, SiftOutputs out,
                        hipStream_t s)
{
    hipLaunchKernelGGL(sift_select_kernel, dim3(npairs), dim3(1024), 0, s,
                       pairs, rr_d1, rr_j1, rr_d2, rule, cand, out);
}

// ---- 4. confirm --------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void sift_confirm_kernel(const SiftPair* __restrict__ pairs, SiftCandidates cand, int mutual, SiftOutputs out)
{
    __shared__ int wsum[16];
    __shared__ int base_s;
    const SiftPair pw = pairs[blockIdx.x];
    const int tid = threadIdx.x;
    const int ncand = cand.count[blockIdx.x];
    if (tid == 0) base_s = 0;
    if (out.dense0)
        for (int i = tid; i < pw.nA; i += 1024) out.dense0[pw.out0_off + i] = -1;
    if (out.dense1)
        for (int jj = tid; jj < pw.nB; jj += 1024) out.dense1[pw.out1_off + jj] = INT32_MAX;
    __syncthreads();

    for (int c0 = 0; c0 < ncand; c0 += 1024) {
        const int c = c0 + tid;
        int i = 0, j = 0, d = 0;
        bool ok = false;
        if (c < ncand) {
            const size_t o = pw.out0_off + c;
            i = cand.i0[o];
            j = cand.j1[o];
            d = cand.d1[o];
            ok = j >= 0 && (!mutual || (int)(unsigned)(cand.vbest[o] & 0xffffffffull) == i);
        }
        const int slot = block_compact_slot(ok, wsum, &base_s);
        if (ok) {
            const size_t o = pw.out0_off + slot;
            out.slot_i0[o] = i;
            out.slot_i1[o] = j;
            if (out.slot_d) out.slot_d[o] = d;
            if (out.dense0) out.dense0[pw.out0_off + i] = j;
            // without the mutual check several rows may name one train row: report the lowest
            if (out.dense1) atomicMin(&out.dense1[pw.out1_off + j], i);
        }
    }
    __syncthreads();
    if (tid == 0) out.counts[blockIdx.x] = base_s;
    if (out.dense1)
        for (int jj = tid; jj < pw.nB; jj += 1024)
            if (out.dense1[pw.out1_off + jj] == INT32_MAX) out.dense1[pw.out1_off + jj] = -1;
}

void launch_sift_confirm(const SiftPair* pairs, int npairs, SiftCandidates cand, int mutual,
                         SiftOutputs out, hipStream_t s)
{
    hipLaunchKernelGGL(sift_confirm_kernel, dim3(npairs), dim3(1024), 0, s, pairs, cand, mutual, out);
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
