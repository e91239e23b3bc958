// Accept tests for float-descriptor pairs, on the device: the tail of the reference's mutual block
// (Others/SuperPoint_data/models/superglue.py:276-293: argmax both ways, mutual agreement, score
// threshold, -1 for "no match") plus the optional distance cap and Lowe ratio test of the L2 ranking.
#include "vsm_internal.h"
#include "f32_kernels.h"

#include <math.h>

namespace vsm {

// Four consecutive rows per thread and round: one round (and one set of barriers) for a pair of 4096 rows.
// copy_out (single-pair calls): when the block is done it copies the listed device words end to end into dst, which
// is pinned host memory, so the caller needs neither a copy command nor another kernel behind this one.
__global__ __launch_bounds__(1024)
void f32_accept_kernel(const SiftPair* __restrict__ pairs, const F32PairRef* __restrict__ refs, F32Rule rule,
                       const int32_t* __restrict__ j1, const float* __restrict__ k1, const float* __restrict__ k2,
                       SiftOutputs out, WordSegments copy_out, uint32_t* __restrict__ copy_dst, F32LeaveClean clean,
                       F32PairArgs one)
{
    __shared__ int wsum[16];
    __shared__ int base_s;
    constexpr int R = 4;
    // a one-pair call hands the pair's records over as arguments: one trip to memory less
    const SiftPair pw = one.given ? one.pair : pairs[blockIdx.x];
    const F32PairRef rf = one.given ? one.ref : refs[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid == 0) base_s = 0;
    // the searches before this kernel are done with the words their splits shared: zero again for the next call
    for (int i = tid; i < clean.thr_words; i += 1024) clean.thr[i] = 0;
    if (out.dense0)
        for (int i = tid; i < pw.nA; i += 1024) out.dense0[pw.out0_off + i] = -1;
    if (out.dense1)
        for (int jj = tid; jj < pw.nB; jj += 1024) out.dense1[pw.out1_off + jj] = INT32_MAX;
    __syncthreads();

    const bool empty = pw.nA == 0 || pw.nB == 0;
    for (int i0 = 0; i0 < pw.nA && !empty; i0 += 1024 * R) {
        const int ib = i0 + tid * R;
        int j[R];
        float best[R];
        unsigned okbits = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = ib + r;
            j[r] = -1;
            best[r] = 0.0f;
            if (i >= pw.nA) continue;
            j[r] = j1[rf.fwd_off + i];
            best[r] = k1[rf.fwd_off + i];
            bool ok = j[r] >= 0;
            if (ok) {
                if (rule.metric == 0) {
                    ok = best[r] > rule.min_score;
                } else {
                    if (rule.max_dist >= 0.0f && best[r] > rule.max_dist) ok = false;
                    if (ok && rule.ratio_num > 0 && pw.nB >= 2) {
                        const double lhs = (double)best[r] * (double)rule.ratio_den * (double)rule.ratio_den;
                        const double rhs = (double)k2[rf.fwd_off + i] * (double)rule.ratio_num * (double)rule.ratio_num;
                        ok = lhs < rhs;
                    }
                }
                if (ok && rule.mutual && rf.bwd_off >= 0 && j1[rf.bwd_off + j[r]] != i) ok = false;
            }
            okbits |= (ok ? 1u : 0u) << r;
        }
        int slot = block_compact_range(__popc(okbits), wsum, &base_s);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!((okbits >> r) & 1u)) continue;
            const int i = ib + r;
            const size_t o = pw.out0_off + slot++;
            out.slot_i0[o] = i;
            out.slot_i1[o] = j[r];
            if (out.slot_d) out.slot_d[o] = __float_as_int(best[r]);
            if (out.dense0) out.dense0[pw.out0_off + i] = j[r];
            // without the mutual check several rows may name one train row: report the lowest
            if (out.dense1) atomicMin(&out.dense1[pw.out1_off + j[r]], i);
        }
    }
    __syncthreads();
    if (tid == 0) out.counts[blockIdx.x] = base_s;
    if (out.dense1)
        for (int jj = tid; jj < pw.nB; jj += 1024)
            if (out.dense1[pw.out1_off + jj] == INT32_MAX) out.dense1[pw.out1_off + jj] = -1;
    if (copy_dst) {
        __syncthreads();   // this block wrote every word the segments name (or an earlier kernel did)
        int base = 0;
        for (int k = 0; k < copy_out.n; ++k) {
            // (of a match list only the matches: fewer words across the bus; its place in copy_dst stays count[k] wide)
            const int words = (copy_out.lists >> k) & 1u ? min(copy_out.count[k], base_s) : copy_out.count[k];
            for (int i = tid; i < words; i += 1024) copy_dst[base + i] = copy_out.src[k][i];
            base += copy_out.count[k];
        }
    }
    if (clean.flag) {
        __syncthreads();   // the flag has been copied out (it is one of the words above)
        if (tid == 0) *clean.flag = 0;
    }
    if (copy_dst && one.done_seq) {
        // every word is on its way to host memory before the word that says so
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(copy_dst + one.done_at, one.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

void launch_f32_accept(const SiftPair* pairs, const F32PairRef* refs, int npairs, F32Rule rule,
                       const int32_t* j1, const float* k1, const float* k2, SiftOutputs out, hipStream_t s,
                       const WordSegments* copy_out, uint32_t* copy_dst, const F32LeaveClean* clean,
                       const F32PairArgs* one)
{
    if (npairs <= 0) return;
    const WordSegments none{};
    const F32LeaveClean nothing{};
    const F32PairArgs not_given{};
    hipLaunchKernelGGL(f32_accept_kernel, dim3(npairs), dim3(1024), 0, s, pairs, refs, rule, j1, k1, k2, out,
                       copy_out && npairs == 1 ? *copy_out : none, copy_out && npairs == 1 ? copy_dst : nullptr,
                       clean && npairs == 1 ? *clean : nothing, one && npairs == 1 ? *one : not_given);
}

__global__ __launch_bounds__(1024)
void f32_confirm_kernel(const SiftPair* __restrict__ pairs, const F32PairRef* __restrict__ refs,
                        const int32_t* __restrict__ j1, SiftOutputs out)
{
    __shared__ int wsum[16];
    __shared__ int base_s;
    const SiftPair pw = pairs[blockIdx.x];
    const F32PairRef rf = refs[blockIdx.x];
    const int tid = threadIdx.x;
    const int ncand = out.counts[blockIdx.x];
    if (tid == 0) base_s = 0;
    if (out.dense0)
        for (int i = tid; i < pw.nA; i += 1024) out.dense0[pw.out0_off + i] = -1;
    if (out.dense1)
        for (int jj = tid; jj < pw.nB; jj += 1024) out.dense1[pw.out1_off + jj] = -1;
    __syncthreads();
    // in place: a kept candidate moves to a slot at or before its own, and a round reads its 1024 candidates
    // before any of them is written
    for (int k0 = 0; k0 < ncand; k0 += 1024) {
        const int k = k0 + tid;
        int i = -1, j = -1, d = 0;
        bool ok = false;
        if (k < ncand) {
            i = out.slot_i0[pw.out0_off + k];
            j = out.slot_i1[pw.out0_off + k];
            if (out.slot_d) d = out.slot_d[pw.out0_off + k];
            ok = rf.bwd_off >= 0 && j1[rf.bwd_off + k] == i;
        }
        const int slot = block_compact_slot(ok, wsum, &base_s);
        if (ok) {
            const size_t o = pw.out0_off + slot;
            out.slot_i0[o] = i;
            out.slot_i1[o] = j;
            if (out.slot_d) out.slot_d[o] = d;
            if (out.dense0) out.dense0[pw.out0_off + i] = j;
            if (out.dense1) out.dense1[pw.out1_off + j] = i;   // mutual: one query row per train row
        }
    }
    __syncthreads();
    if (tid == 0) out.counts[blockIdx.x] = base_s;
}

void launch_f32_confirm(const SiftPair* pairs, const F32PairRef* refs, int npairs, const int32_t* j1,
                        SiftOutputs out, hipStream_t s)
{
    if (npairs <= 0) return;
    hipLaunchKernelGGL(f32_confirm_kernel, dim3(npairs), dim3(1024), 0, s, pairs, refs, j1, out);
}

// list_total (may be NULL): of the segments marked as match lists only *list_total words are copied (a count that
// exists on the device only); every segment keeps its place, count[k] words wide, in dst.
__global__ void gather_words_kernel(WordSegments seg, uint32_t* __restrict__ dst, const int64_t* __restrict__ list_total)
{
    const long long listed = list_total ? *list_total : INT32_MAX;
    int base = 0;
    for (int k = 0; k < seg.n; ++k) {
        const int words = (seg.lists >> k) & 1u ? (int)min((long long)seg.count[k], listed) : seg.count[k];
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x)
            dst[base + i] = seg.src[k][i];
        base += seg.count[k];
    }
}

void launch_gather_words(const WordSegments& seg, uint32_t* dst, hipStream_t s, const int64_t* list_total)
{
    int total = 0;
    for (int k = 0; k < seg.n; ++k) total += seg.count[k];
    if (total <= 0) return;
    hipLaunchKernelGGL(gather_words_kernel, dim3(std::min(64, (total + 255) / 256)), dim3(256), 0, s, seg, dst, list_total);
}

// ---- merge of per-slice 2-NN lists (database cut into contiguous row ranges) ---------------
// packed: [nslices][4][nq] 32-bit words = best key, runner-up key (float bits), best row, runner-up
// row (slice-local, -1 = none).  Order: key, then global row (lower first), as on the host.
__global__ void merge_top2_kernel(const int32_t* __restrict__ packed, const int64_t* __restrict__ base,
                                  int nslices, int nq, int metric, float* __restrict__ ok1,
                                  float* __restrict__ ok2, int64_t* __restrict__ oi1, int64_t* __restrict__ oi2)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const bool dot = metric == 0;
    float b1 = 0.0f, b2 = 0.0f;
    int64_t x1 = -1, x2 = -1;
    for (int s = 0; s < nslices; ++s) {
        const int32_t* sl = packed + (size_t)s * 4 * nq;
        for (int r = 0; r < 2; ++r) {
            const int32_t li = sl[(size_t)(2 + r) * nq + q];
            if (li < 0) continue;
            const float k = __int_as_float(sl[(size_t)r * nq + q]);
            const int64_t id = base[s] + li;
            const bool beats1 = x1 < 0 || (dot ? k > b1 : k < b1) || (k == b1 && id < x1);
            if (beats1) {
                b2 = b1; x2 = x1;
                b1 = k; x1 = id;
            } else if (x2 < 0 || (dot ? k > b2 : k < b2) || (k == b2 && id < x2)) {
                b2 = k; x2 = id;
            }
        }
    }
    ok1[q] = b1; ok2[q] = b2; oi1[q] = x1; oi2[q] = x2;
}

void launch_merge_top2(const int32_t* packed, const int64_t* base, int nslices, int nq, int metric,
                       float* ok1, float* ok2, int64_t* oi1, int64_t* oi2, hipStream_t s)
{
    if (nq <= 0) return;
    hipLaunchKernelGGL(merge_top2_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, packed, base, nslices, nq, metric,
                       ok1, ok2, oi1, oi2);
}

}  // namespace vsm
