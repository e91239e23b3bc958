// Image-pair shortlist from a visual vocabulary: word histograms, weights, histogram-intersection
// scores, best partners of every image, ordered pair list.  All integer arithmetic, so the result does
// not depend on the order in which the device adds things up.
//
// The reference chooses pairs by position in the sequence only (generate_imagePairs, main.py:157-169);
// this is the content-based counterpart for unordered collections.
#include "vsm_internal.h"

#include <algorithm>

namespace vsm {

namespace {

constexpr int kScoreTile = 64;   // images per side of a score tile (256 threads, 4 x 4 scores each)
constexpr int kScoreStep = 32;   // words per LDS step

// One block column per image of the submission: words[offsets[p] .. offsets[p+1]) are the nearest
// vocabulary rows of image first_image + p.
__global__ __launch_bounds__(256) void bow_count_kernel(const int64_t* __restrict__ offsets,
                                                        const int32_t* __restrict__ words, int first_image,
                                                        int nwords, uint32_t* __restrict__ hist)
{
    const int p = blockIdx.y;
    const int64_t lo = offsets[p], hi = offsets[p + 1];
    uint32_t* h = hist + (size_t)(first_image + p) * nwords;
    for (int64_t r = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; r < hi; r += (int64_t)gridDim.x * 256) {
        const int w = words[r];
        if (w >= 0 && w < nwords) atomicAdd(&h[w], 1u);
    }
}

// df[w] = number of images that hold word w (zeroed by the caller); grid (words / 256, slices of 32 images).
__global__ __launch_bounds__(256) void bow_df_kernel(const uint32_t* __restrict__ hist, int nimg, int nwords,
                                                     uint32_t* __restrict__ df)
{
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nwords) return;
    const int i0 = blockIdx.y * 32, i1 = min(i0 + 32, nimg);
    uint32_t n = 0;
    for (int i = i0; i < i1; ++i) n += hist[(size_t)i * nwords + w] != 0u;
    if (n) atomicAdd(&df[w], n);
}

// The same with the histogram of an image in LDS (vocabularies of at most kLdsWords words): one
// workgroup per image, plain stores of the finished row.
constexpr int kLdsWords = 16384;
__global__ __launch_bounds__(1024) void bow_count_lds_kernel(const int64_t* __restrict__ offsets,
                                                             const int32_t* __restrict__ words, int first_image,
                                                             int nwords, uint32_t* __restrict__ hist)
{
    extern __shared__ uint32_t s_hist[];
    const int p = blockIdx.x, tid = threadIdx.x;
    for (int w = tid; w < nwords; w += 1024) s_hist[w] = 0u;
    __syncthreads();
    const int64_t lo = offsets[p], hi = offsets[p + 1];
    for (int64_t r = lo + tid; r < hi; r += 1024) {
        const int w = words[r];
        if (w >= 0 && w < nwords) atomicAdd(&s_hist[w], 1u);
    }
    __syncthreads();
    uint32_t* h = hist + (size_t)(first_image + p) * nwords;
    for (int w = tid; w < nwords; w += 1024) h[w] = s_hist[w];
}

// weight[w] = 0 for a word no image holds, else 1 + floor(log2(nimg / df)) with the integer quotient;
// the histograms are weighted in place (weight * min(a, b) = min(weight * a, weight * b)).
__global__ __launch_bounds__(256) void bow_scale_kernel(uint32_t* __restrict__ hist, const uint32_t* __restrict__ df,
                                                        int nimg, int nwords, size_t total)
{
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (size_t)gridDim.x * 256) {
        const uint32_t d = df[k % (size_t)nwords];
        hist[k] *= d ? (uint32_t)(32 - __clz(nimg / (int)d)) : 0u;
    }
}

// scores[i][j] = sum over words of min(g[i][w], g[j][w]) for the weighted histograms g; tiles on and
// above the diagonal are computed and added to both triangles.  blockIdx.z takes a slice of
// words_per_slice words (a multiple of kScoreStep), so that a few hundred images still give the chip
// enough workgroups; the slices meet in integer atomic adds on scores, which the caller has zeroed.
__global__ __launch_bounds__(256) void bow_score_kernel(const uint32_t* __restrict__ g, int nimg, int nwords,
                                                        int words_per_slice, int32_t* __restrict__ scores)
{
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const int k_lo = blockIdx.z * words_per_slice, k_hi = min(nwords, k_lo + words_per_slice);
    __shared__ __align__(16) uint32_t sa[kScoreStep][kScoreTile + 4];   // rows of 68 words: 16-byte reads stay aligned
    __shared__ __align__(16) uint32_t sb[kScoreStep][kScoreTile + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int kk = tid & 31, r0 = tid >> 5;
    uint32_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0u;
    for (int k0 = k_lo; k0 < k_hi; k0 += kScoreStep) {
        const bool kin = k0 + kk < k_hi;
#pragma unroll
        for (int r = r0; r < kScoreTile; r += 8) {
            const int ia = bi * kScoreTile + r, ib = bj * kScoreTile + r;
            sa[kk][r] = (kin && ia < nimg) ? g[(size_t)ia * nwords + k0 + kk] : 0u;
            sb[kk][r] = (kin && ib < nimg) ? g[(size_t)ib * nwords + k0 + kk] : 0u;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < kScoreStep; ++k) {
            const uint4 qa = *reinterpret_cast<const uint4*>(&sa[k][ty * 4]);
            const uint4 qb = *reinterpret_cast<const uint4*>(&sb[k][tx * 4]);
            const uint32_t va[4] = {qa.x, qa.y, qa.z, qa.w}, vb[4] = {qb.x, qb.y, qb.z, qb.w};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] += min(va[a], vb[b]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = bi * kScoreTile + ty * 4 + a, j = bj * kScoreTile + tx * 4 + b;
            if (i < nimg && j < nimg && acc[a][b] != 0u) {
                atomicAdd(&scores[(size_t)i * nimg + j], (int32_t)acc[a][b]);
                if (bi != bj) atomicAdd(&scores[(size_t)j * nimg + i], (int32_t)acc[a][b]);   // a diagonal tile holds both (i, j) and (j, i) itself
            }
        }
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, d), hi = __shfl_xor((unsigned)(v >> 32), d);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

// One workgroup per image: its topk partners by (score descending, image ascending), scores of 0
// excluded; every choice marks the unordered pair.
__global__ __launch_bounds__(256) void bow_topk_kernel(const int32_t* __restrict__ scores, int nimg, int topk,
                                                       uint8_t* __restrict__ mark)
{
    __shared__ unsigned long long s_part[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const int32_t* row = scores + (size_t)i * nimg;
    unsigned long long prev = ~0ull;
    for (int r = 0; r < topk; ++r) {
        unsigned long long best = 0ull;
        for (int j = tid; j < nimg; j += 256) {
            const int32_t s = row[j];
            if (j == i || s <= 0) continue;
            const unsigned long long key = ((unsigned long long)(uint32_t)s << 32) | (0xFFFFFFFFu - (uint32_t)j);
            if (key < prev && key > best) best = key;
        }
        best = wave_max_u64(best);
        __syncthreads();   // the previous round's s_part has been read by every thread
        if ((tid & 63) == 0) s_part[tid >> 6] = best;
        __syncthreads();
        best = s_part[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) best = s_part[w] > best ? s_part[w] : best;
        if (best == 0ull) break;   // the same value in every thread
        if (tid == 0) {
            const int j = (int)(0xFFFFFFFFu - (uint32_t)best);
            const int a = i < j ? i : j, b = i < j ? j : i;
            mark[(size_t)a * nimg + b] = 1;
        }
        prev = best;
    }
}

__global__ __launch_bounds__(256) void bow_rowcount_kernel(const uint8_t* __restrict__ mark, int nimg,
                                                           int32_t* __restrict__ counts)
{
    __shared__ int s_part[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    int n = 0;
    for (int j = i + 1 + tid; j < nimg; j += 256) n += mark[(size_t)i * nimg + j];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d);
    if ((tid & 63) == 0) s_part[tid >> 6] = n;
    __syncthreads();
    if (tid == 0) counts[i] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// offsets[0..nimg] = exclusive sums of counts; one workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void bow_offsets_kernel(const int32_t* __restrict__ counts, int nimg,
                                                           int64_t* __restrict__ offsets)
{
    __shared__ int64_t s_wave[16];
    __shared__ int64_t s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nimg; i0 += 1024) {
        const int i = i0 + tid;
        const int64_t v = i < nimg ? counts[i] : 0;
        int64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int64_t woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) woff += s_wave[w];
            tot += s_wave[w];
        }
        const int64_t base = s_base;
        if (i < nimg) offsets[i] = base + woff + incl - v;
        __syncthreads();
        if (tid == 0) s_base = base + tot;
        __syncthreads();
    }
    if (tid == 0) offsets[nimg] = s_base;
}

// Row i of the marks, in ascending j, into pairs[offsets[i] ..].
__global__ __launch_bounds__(1024) void bow_emit_kernel(const uint8_t* __restrict__ mark, int nimg,
                                                        const int64_t* __restrict__ offsets, int32_t* __restrict__ pairs)
{
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int i = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) base_s = 0;
    __syncthreads();
    const int64_t out = offsets[i];
    if (offsets[i + 1] == out) return;   // the same in every thread
    for (int j0 = i + 1; j0 < nimg; j0 += 1024) {
        const int j = j0 + tid;
        const bool flag = j < nimg && mark[(size_t)i * nimg + j] != 0;
        const int slot = block_compact_slot(flag, wsum, &base_s);
        if (flag) {
            pairs[2 * (out + slot)] = i;
            pairs[2 * (out + slot) + 1] = j;
        }
    }
}

// ---- vocabulary training (Lloyd iterations on u8 descriptors, integers only) ----------------------

// byte k of row r of a fragment-major int8 set, as the u8 it was uploaded as
__device__ __forceinline__ unsigned frag_byte(const int8_t* __restrict__ frag, int r, int k)
{
    const size_t o = (size_t)(r >> 5) * kTileBytes + (size_t)(k >> 5) * 1024 + (size_t)((((k >> 4) & 1) * 32 + (r & 31)) * 16) + (k & 15);
    return (unsigned)(uint8_t)frag[o] ^ 0x80u;
}

// One wave per word: the row picked as its first value.
__global__ __launch_bounds__(256) void kmeans_seed_kernel(const int8_t* const* __restrict__ frags,
                                                          const int32_t* __restrict__ rows, int nwords,
                                                          uint8_t* __restrict__ words)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= nwords) return;
    const int8_t* frag = frags[w];
    const int r = rows[w];
    words[(size_t)w * kDescDim + lane] = (uint8_t)frag_byte(frag, r, lane);
    words[(size_t)w * kDescDim + 64 + lane] = (uint8_t)frag_byte(frag, r, 64 + lane);
}

// Sums of the rows of every word.  Workgroup (x, y) takes the words [x * kSumWords, (x + 1) * kSumWords)
// and the images y, y + gridDim.y, ... of the submission: it walks the word of every row of those images,
// adds the rows of its words into sums kept in LDS (one wave per row, two components per lane), and adds
// the finished partial sums to memory.  Pair p of the submission is (image, vocabulary) with every row
// accepted, so words[offsets[p] + r] is the word of row r.
constexpr int kSumWords = 128;   // 128 words x 128 components x 4 bytes = 64 KiB of LDS
__global__ __launch_bounds__(512) void kmeans_accumulate_kernel(const SiftPair* __restrict__ pairs, int nimages,
                                                                const int64_t* __restrict__ offsets,
                                                                const int32_t* __restrict__ words, int nwords,
                                                                unsigned* __restrict__ sums,
                                                                unsigned* __restrict__ counts)
{
    extern __shared__ unsigned s_sum[];   // [kSumWords][128] then [kSumWords] counts
    unsigned* s_cnt = s_sum + kSumWords * kDescDim;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w0 = blockIdx.x * kSumWords, w1 = min(nwords, w0 + kSumWords);
    for (int k = tid; k < kSumWords * kDescDim + kSumWords; k += 512) s_sum[k] = 0u;
    __syncthreads();
    for (int p = blockIdx.y; p < nimages; p += gridDim.y) {
        const int8_t* __restrict__ frag = pairs[p].fragA;
        const int n = pairs[p].nA;
        const int32_t* __restrict__ wp = words + offsets[p];
        for (int r0 = wave * 64; r0 < n; r0 += 8 * 64) {
            const int r = r0 + lane;
            const int w = r < n ? wp[r] : -1;
            unsigned long long mine = __ballot(w >= w0 && w < w1);
            while (mine) {   // four rows at a time, so that their loads are in flight together
                int src[4], wr[4];
                unsigned lo[4], hi[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    src[u] = mine ? __ffsll((long long)mine) - 1 : -1;
                    mine &= mine - 1;   // 0 stays 0
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (src[u] >= 0) {
                        wr[u] = __shfl(w, src[u]) - w0;
                        lo[u] = frag_byte(frag, r0 + src[u], lane);
                        hi[u] = frag_byte(frag, r0 + src[u], 64 + lane);
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (src[u] >= 0) {
                        atomicAdd(&s_sum[wr[u] * kDescDim + lane], lo[u]);
                        atomicAdd(&s_sum[wr[u] * kDescDim + 64 + lane], hi[u]);
                        if (lane == 0) atomicAdd(&s_cnt[wr[u]], 1u);
                    }
            }
        }
    }
    __syncthreads();
    for (int k = tid; k < (w1 - w0) * kDescDim; k += 512)
        if (s_sum[k]) atomicAdd(&sums[(size_t)w0 * kDescDim + k], s_sum[k]);
    for (int k = tid; k < w1 - w0; k += 512)
        if (s_cnt[k]) atomicAdd(&counts[w0 + k], s_cnt[k]);
}

// The same for large vocabularies, straight into memory: one wave per descriptor row adds the row to the sums of its word.  Pair p of the submission is
// (image, vocabulary) with every row accepted, so words[offsets[p] + r] is the word of row r.
__global__ __launch_bounds__(256) void kmeans_accumulate_rows_kernel(const SiftPair* __restrict__ pairs,
                                                                const int64_t* __restrict__ offsets,
                                                                const int32_t* __restrict__ words, int nwords,
                                                                unsigned* __restrict__ sums,
                                                                unsigned* __restrict__ counts)
{
    const int p = blockIdx.y, lane = threadIdx.x & 63;
    const int8_t* __restrict__ frag = pairs[p].fragA;
    const int n = pairs[p].nA;
    const int64_t off = offsets[p];
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += gridDim.x * 4) {
        const int w = words[off + r];
        if (w < 0 || w >= nwords) continue;
        atomicAdd(&sums[(size_t)w * kDescDim + lane], frag_byte(frag, r, lane));
        atomicAdd(&sums[(size_t)w * kDescDim + 64 + lane], frag_byte(frag, r, 64 + lane));
        if (lane == 0) atomicAdd(&counts[w], 1u);
    }
}

// word = round-half-up of the mean of its rows; a word without rows keeps its value.
__global__ __launch_bounds__(256) void kmeans_update_kernel(const unsigned* __restrict__ sums,
                                                            const unsigned* __restrict__ counts, int nwords,
                                                            uint8_t* __restrict__ words)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= (size_t)nwords * kDescDim) return;
    const unsigned long long c = counts[k / kDescDim];
    if (c) words[k] = (uint8_t)((2ull * sums[k] + c) / (2ull * c));   // sums < 2^32: at most 2^24 rows of bytes
}

}  // namespace

void launch_kmeans_seed(const int8_t* const* frags, const int32_t* rows, int nwords, uint8_t* words, hipStream_t s)
{
    if (nwords <= 0) return;
    hipLaunchKernelGGL(kmeans_seed_kernel, dim3((nwords + 3) / 4), dim3(256), 0, s, frags, rows, nwords, words);
}

void launch_kmeans_accumulate(const SiftPair* pairs, int nimages, const int64_t* offsets, const int32_t* words,
                              int nwords, unsigned* sums, unsigned* counts, hipStream_t s)
{
    if (nimages <= 0) return;
    const int lds = (kSumWords * kDescDim + kSumWords) * 4;
    static DeviceFlags attr_set;   // the attribute is per device
    ensure_dynamic_lds(attr_set, reinterpret_cast<const void*>(kmeans_accumulate_kernel), lds);
    const int ranges = (nwords + kSumWords - 1) / kSumWords;
    if (ranges > 64) {   // every workgroup reads the words of its images once per range: too much for large vocabularies
        hipLaunchKernelGGL(kmeans_accumulate_rows_kernel, dim3(64, nimages), dim3(256), 0, s, pairs, offsets, words, nwords,
                           sums, counts);
        return;
    }
    // enough workgroups to fill the chip, few enough that the partial sums they add to memory stay small
    const int slices = std::max(1, std::min(nimages, (1024 + ranges - 1) / ranges));
    hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3(ranges, slices), dim3(512), lds, s, pairs, nimages, offsets, words,
                       nwords, sums, counts);
}

void launch_kmeans_update(const unsigned* sums, const unsigned* counts, int nwords, uint8_t* words,
                          hipStream_t s)
{
    if (nwords <= 0) return;
    const size_t total = (size_t)nwords * kDescDim;
    hipLaunchKernelGGL(kmeans_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, sums, counts, nwords, words);
}

void launch_bow_count(const int64_t* offsets, const int32_t* words, int first_image, int nimages, int nwords,
                      uint32_t* hist, hipStream_t s)
{
    if (nimages <= 0) return;
    if (nwords <= kLdsWords)
        hipLaunchKernelGGL(bow_count_lds_kernel, dim3(nimages), dim3(1024), (size_t)nwords * 4, s, offsets, words,
                           first_image, nwords, hist);
    else
        hipLaunchKernelGGL(bow_count_kernel, dim3(32, nimages), dim3(256), 0, s, offsets, words, first_image, nwords, hist);
}

void launch_bow_shortlist(uint32_t* hist, int nimg, int nwords, int topk, uint32_t* weight, int32_t* scores,
                          uint8_t* mark, int32_t* counts, int64_t* offsets, int32_t* pairs, hipStream_t s)
{
    if (nimg <= 0) return;
    const size_t total = (size_t)nimg * nwords;
    (void)hipMemsetAsync(weight, 0, (size_t)nwords * 4, s);
    hipLaunchKernelGGL(bow_df_kernel, dim3((nwords + 255) / 256, (nimg + 31) / 32), dim3(256), 0, s, hist, nimg, nwords, weight);
    const unsigned sblocks = (unsigned)std::min<size_t>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(bow_scale_kernel, dim3(sblocks), dim3(256), 0, s, hist, weight, nimg, nwords, total);
    const int tiles = (nimg + kScoreTile - 1) / kScoreTile;
    const int upper = tiles * (tiles + 1) / 2, steps = (nwords + kScoreStep - 1) / kScoreStep;
    const int nslices = std::max(1, std::min({(2048 + upper - 1) / upper, steps, 1024}));
    const int words_per_slice = (steps + nslices - 1) / nslices * kScoreStep;
    (void)hipMemsetAsync(scores, 0, (size_t)nimg * nimg * 4, s);
    hipLaunchKernelGGL(bow_score_kernel, dim3(tiles, tiles, (nwords + words_per_slice - 1) / words_per_slice), dim3(256), 0, s,
                       hist, nimg, nwords, words_per_slice, scores);
    hipLaunchKernelGGL(bow_topk_kernel, dim3(nimg), dim3(256), 0, s, scores, nimg, topk, mark);
    hipLaunchKernelGGL(bow_rowcount_kernel, dim3(nimg), dim3(256), 0, s, mark, nimg, counts);
    hipLaunchKernelGGL(bow_offsets_kernel, dim3(1), dim3(1024), 0, s, counts, nimg, offsets);
    hipLaunchKernelGGL(bow_emit_kernel, dim3(nimg), dim3(1024), 0, s, mark, nimg, offsets, pairs);
}

}  // namespace vsm
