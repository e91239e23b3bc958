// Internal declarations shared by the HIP kernels and the C-ABI layer.
#pragma once

#include <hip/hip_runtime.h>

#include "device_flags.h"
#include "f32_kernels.h"
#include <stdint.h>

namespace vsm {

// ---- SIFT-128 u8 path ---------------------------------------------------------
//
// Device layout of a u8 descriptor set ("fragment-major int8"):
//   rows are padded to a multiple of kRowPad; every 32-row tile occupies 4096 bytes,
//   split into four 1024-byte k-steps; inside a k-step lane l = h*32 + r (r = row in
//   tile, h = 0/1) owns the 16 bytes k = 32*s + 16*h .. +15 of row r.  One wave reads
//   a k-step as one perfectly coalesced 16-byte-per-lane load, and the 16 bytes are
//   exactly that lane's operand of v_mfma_i32_32x32x32_i8.  Bytes are stored as
//   int8 = u8 - 128 (u8 ^ 0x80), padding rows as 0.
//   nrm8c[row] = ((sum of squared int8 values) << 8) | (row & 255); padding rows carry
//   kNormPad in the upper 24 bits.
// Squared L2 of two u8 descriptors = nrm(a) + nrm(b) - 2*dot(int8 a, int8 b), exact in
// int32 (< 2^23).
constexpr int kDescDim = 128;
constexpr int kRowPad = 512;
constexpr int kTileBytes = 4096;      // 32 rows * 128 bytes
constexpr int kStripRows = 512;       // query rows per workgroup of the query pass (4 waves * 128)
constexpr int kVerifyBlock = 256;     // candidates per workgroup of the verify pass (4 waves * 64)
constexpr int kStepRows = 256;        // streamed rows per main-loop step
constexpr int kChunkCols = 8192;      // streamed rows per work item of the query-side pass, at most
constexpr int32_t kNormPad = 0x7FFFFF00;
constexpr int32_t kDistPad = 0x7FFFFF;  // (kNormPad >> 8): "no such neighbour"
// Half-distance domain of the tile-minimum query pass: for a query row p and a train row q (both int8),
//     y = floor(nrm(q) / 2) - dot(p, q),   nrm(q) - 2 dot(p, q) = 2 y + (nrm(q) & 1),
// -2^21 <= y <= 2^21 + 2^20 for real rows; padding rows (zero operand bytes, so their accumulators end where they
// start) and "no such neighbour" are kYPad = 2^22, small enough for y * 256 + (tile within the chunk) to fit an int32:
// the query pass ranks such packed keys.
// The int32 array of a u8 set holds nrm8c[n_pad] followed by these constants, cq[n_pad]:
//     cq[row] = floor(nrm / 2) + (sum of the row's int8 values)
// (the query operand enters the MFMA as ~p = -p - 1, so the accumulator that starts at cq ends at y).
constexpr int32_t kYPad = 1 << 22;

struct SiftPair {
    const int8_t* fragA;     // image 0 (queries)
    const int32_t* nrmA;
    const int8_t* fragB;     // image 1 (train)
    const int32_t* nrmB;
    int32_t nA, nB;          // real row counts
    int32_t nA_pad, nB_pad;  // padded to kRowPad
    int32_t nstrips, nchunks;
    int32_t chunk_cols, pad0;   // train rows per work item of the query pass (a multiple of kStepRows)
    int64_t row_off;   // into the per-(chunk,row) partial arrays: nchunks * nA_pad entries
    int64_t out0_off;  // into per-query-row arrays (nA entries): candidates, outputs
    int64_t out1_off;  // into per-train-row outputs (nB entries)
};

struct SiftItem {
    int32_t pair;
    int16_t strip;   // query pass: kStripRows-row strip of image 0; verify pass: block of kVerifyBlock candidates
    int16_t chunk;   // query pass: 8192-row chunk of image 1
};

struct MatchRule {
    int64_t num2, den2;   // ratio^2 as num2/den2; num2 <= 0 disables
    int64_t max_dist;     // < 0 disables
    int32_t mutual;
};

// candidates = rows that passed the distance/ratio tests, compact per pair at out0_off
struct SiftCandidates {
    int32_t* i0;
    int32_t* j1;
    int32_t* d1;
    int32_t* count;   // [npairs]
    // nearest image-0 row of column j1 as (distance with its sign bit flipped << 32 | row): set to all ones by the select pass,
    // lowered by the verify pass with an atomic minimum (lowest row among equal distances)
    unsigned long long* vbest;
    // tile-minimum path only.  Between its select and resolve passes j1 holds the 8-row group of the nearest
    // row (tile * 4 + lane quarter), d1 its half-distance y1, aux the smallest half-distance outside that group;
    // the resolve pass replaces j1 by the row (-1: rejected) and d1 by the exact squared distance.
    int32_t* aux;
    // Candidates the resolve pass could not settle from one group (the minimum tied between groups, or the runner-up's
    // parity bit deciding the ratio test): rescan_cnt[pair] of them, their positions in the pair's candidate list at
    // rescan_list[out0_off + k] in no particular order.  The rescan pass gathers their rows of image 0, 512 per work
    // item, and runs the packed form of the query pass over the whole train set for them; rescan_items (room for as
    // many items as the query pass has) and rescan_nitems ([0] = items, [1] = candidates of the whole submission) are
    // written on the device, rs_d1 / rs_j1 / rs_d2 hold its per-(chunk, list slot) results at
    // row_off + chunk * nA_pad + slot.
    int32_t* rescan_cnt;
    int32_t* rescan_list;
    SiftItem* rescan_items;
    int32_t* rescan_nitems;
    int32_t* rs_d1;
    int32_t* rs_j1;
    int32_t* rs_d2;
    // tile-minimum path only: the image-0 rows the verify pass has to look at.  dmax[pair] = the largest exact
    // distance of a candidate that passed its tests; a row whose own nearest distance (at least
    // 2 y1 + |a|^2) exceeds it is farther than that from EVERY train row, so it cannot come before any
    // candidate in any candidate's column.  vrows (at out0_off, ascending) lists the other rows, nvrows[pair]
    // counts them.  NULL in the packed path and for rules without a ratio test or distance cap (every row is a
    // candidate then and nothing bounds the rows that matter): the verify pass then streams all of image 0.
    int32_t* dmax;
    int32_t* vrows;
    int32_t* vcodes;   // per listed row: its squared norm << 8 | (position in the list & 255), the word the verify pass streams
    int32_t* nvrows;
    // one 32-row tile of zero operands (kTileBytes): what the positions past the end of a row list stream, so
    // that their packed keys stay the padding constant (a real row's products would overflow it)
    const int8_t* zero_tile;
};

struct SiftOutputs {
    // compacted matches, per pair at out0_off (capacity nA)
    int32_t* slot_i0;
    int32_t* slot_i1;
    int32_t* slot_d;   // optional
    int32_t* counts;   // [npairs]
    // optional dense forms
    int32_t* dense0;   // at out0_off, nA entries
    int32_t* dense1;   // at out1_off, nB entries
    // optional raw 2-NN
    int32_t* knn_j1;
    int32_t* knn_d1;
    int32_t* knn_d2;
};

// Block-wide stream compaction step for 1024-thread blocks: returns this thread's output slot
// (meaningful if flag) and advances *base_s by the number of flagged threads.  wsum: 16 ints of LDS.
__device__ __forceinline__ int block_compact_slot(bool flag, int* wsum, int* base_s)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long bal = __ballot(flag);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
        const int c = wsum[w];
        if (w < wave) woff += c;
        tot += c;
    }
    const int base = *base_s;
    __syncthreads();
    if (tid == 0) *base_s = base + tot;
    __syncthreads();
    return base + woff + before;
}

// The same for threads that flag `n` consecutive outputs each: returns the first of this thread's slots.
__device__ __forceinline__ int block_compact_range(int n, int* wsum, int* base_s)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
        const int c = wsum[w];
        if (w < wave) woff += c;
        tot += c;
    }
    const int base = *base_s;
    __syncthreads();
    if (tid == 0) *base_s = base + tot;
    __syncthreads();
    return base + woff + incl - n;
}

// ---- float path: accept tests of a submission of pairs ---------------------------------
struct F32PairRef {
    int64_t fwd_off;   // rows of the image-0 -> image-1 search in the j1 / k1 / k2 arrays
    int64_t bwd_off;   // rows of the image-1 -> image-0 search (mutual check); < 0: none
};
struct F32Rule {
    int32_t metric, mutual;
    int32_t ratio_num, ratio_den;   // num <= 0 disables
    float min_score;                // dot ranking: accept iff best > min_score
    float max_dist;                 // L2 ranking: accept iff best <= max_dist; < 0 disables
};
// One block per pair; only nA, nB, out0_off and out1_off of SiftPair are read.  slot_d receives the
// bits of the best key of every accepted row.
// Several device arrays of 32-bit words copied end to end into one (so that one copy brings a small
// result to the host).
struct WordSegments {
    const uint32_t* src[8];
    int count[8];
    int n;
    unsigned lists;   // bit k: segment k is a match list, of which only the pair's match count of words matter
};
// What a one-pair call leaves zeroed for the next one, so that the next needs no fill command in front of its
// searches: the words the splits of its searches shared, and the survivor-overflow flag once it has been copied out.
struct F32LeaveClean {
    unsigned* thr;
    int thr_words;
    int* flag;
};
// A one-pair call's records as arguments of the accept kernel, and the word of copy_dst (index done_at) that
// receives done_seq (non-zero) after every copied word: the caller may watch that word instead of waiting for
// the stream.
struct F32PairArgs {
    SiftPair pair;
    F32PairRef ref;
    int32_t given;
    uint32_t done_at, done_seq;
};
// copy_out / copy_dst (one pair only, may be NULL): the block finally copies those device words end to end into
// copy_dst, pinned host memory.
void launch_f32_accept(const SiftPair* pairs, const F32PairRef* refs, int npairs, F32Rule rule,
                       const int32_t* j1, const float* k1, const float* k2, SiftOutputs out, hipStream_t s,
                       const WordSegments* copy_out = nullptr, uint32_t* copy_dst = nullptr,
                       const F32LeaveClean* clean = nullptr, const F32PairArgs* one = nullptr);

// The device records of a one-pair float submission as kernel arguments (see bf16_coarse_args_kernel): up to two
// searches (forward, reverse), the pair record and its references, and where the last two go in the block.
struct F32PlanArgs {
    CoarseProblem probs[2];
    SiftPair pair;
    F32PairRef ref;
    int32_t nprobs;
    int32_t pbytes, pair_bytes;   // bytes the searches / the pair records take in the block (256-byte multiples)
};

// Second half of a mutual check whose reverse searches ran on the candidates only: candidate k of a pair
// (slot_i0 / slot_i1 / slot_d at out0_off + k, k < counts[pair]) stays when the reverse search of slot k
// (j1[bwd_off + k]) names its query row.  Compacts the lists in place, rewrites counts, fills the dense forms.
void launch_f32_confirm(const SiftPair* pairs, const F32PairRef* refs, int npairs, const int32_t* j1,
                        SiftOutputs out, hipStream_t s);

// device merge of per-slice 2-NN lists; packed = [nslices][4][nq] words (key1, key2, row1, row2)
void launch_merge_top2(const int32_t* packed, const int64_t* base, int nslices, int nq, int metric,
                       float* ok1, float* ok2, int64_t* oi1, int64_t* oi2, hipStream_t s);

void launch_sift_transform(const uint8_t* des, int n, int n_pad, int8_t* frag, int32_t* nrm8c,
                           hipStream_t s);
// one-pair submission planned on the device: *pair_out = sp (row_off = out0_off = out1_off = 0), items =
// every (strip, chunk) of it in strip-major order
// zero_pair (may be NULL): two ints the kernel also clears
void launch_sift_plan_single(const SiftPair& sp, SiftPair* pair_out, SiftItem* items, hipStream_t s, int32_t* zero_pair = nullptr);
// query-side pass: per (chunk, query row) best train row, its distance, runner-up distance (INT32_MAX
// throughout without want_second: a cheaper epilogue for callers that never look at it)
void launch_sift_query(const SiftPair* pairs, const SiftItem* items, int nitems,
                       int32_t* rr_d1, int32_t* rr_j1, int32_t* rr_d2, bool want_second, hipStream_t s);
// Tile-minimum form of the query pass and its follow-ups (ratio-test submissions): per (chunk, query row)
// the smallest half-distance y1, the 8-row group of a 32-row train tile that holds it (tile * 4 + lane quarter),
// and the smallest half-distance outside that group; select with optimistic tests (a superset of the final
// candidates); resolve from the one group (row, exact distance, exact tests), rescan of the few rows one group
// cannot settle.
void launch_sift_query_tile(const SiftPair* pairs, const SiftItem* items, int nitems,
                            int32_t* rr_y1, int32_t* rr_tile, int32_t* rr_k2, hipStream_t s);
// slice_counts: sift_select_slices(max_rows) ints per pair of scratch (a pair of many rows is cut into slices, each
// with a workgroup of its own; untouched when there is one slice).
constexpr int kMaxRowSlices = 64;
int sift_select_slices(int max_rows);
int sift_verify_rows_slices(int max_rows);
// The tile-minimum query pass with the select and resolve passes in its tail, and (finish) the rescan pass and the pass
// that gathers the strips' candidate segments: for submissions whose every pair has at most kChunkCols rows in either set.
// seg_counts: kChunkCols / kStripRows ints per pair of scratch.
void launch_sift_query_tile_fused(const SiftPair* pairs, int npairs, const SiftItem* items, int nitems,
                                  int32_t* rr_y1, MatchRule rule, SiftCandidates cand, int32_t* seg_counts,
                                  hipStream_t s);
void launch_sift_fused_finish(const SiftPair* pairs, int npairs, int max_items, MatchRule rule, SiftCandidates cand,
                              const int32_t* seg_counts, hipStream_t s);
void launch_sift_select_tile(const SiftPair* pairs, int npairs, int max_rows, int max_chunks, int32_t* rr_y1,
                             int32_t* rr_tile, int32_t* rr_k2, MatchRule rule, SiftCandidates cand,
                             int32_t* slice_counts, hipStream_t s);
// The rescan pass over the candidates that launch_sift_resolve or the fused query pass left unsettled (see
// SiftCandidates::rescan_cnt): plan, gathered packed query pass, merge of its chunks + exact tests.
void launch_sift_rescan(const SiftPair* pairs, int npairs, int max_items, MatchRule rule, SiftCandidates cand, hipStream_t s);
// After the resolve pass: cand.vrows / cand.nvrows from the merged half-distances the select pass left in
// chunk 0's records of rr_y1.
// slice_counts: sift_verify_rows_slices(max_rows) ints per pair of scratch.
void launch_sift_verify_rows(const SiftPair* pairs, int npairs, int max_rows, const int32_t* rr_y1, SiftCandidates cand,
                             int32_t* slice_counts, hipStream_t s);
void launch_sift_resolve(const SiftPair* pairs, int npairs, int max_rows, int max_items, MatchRule rule,
                         SiftCandidates cand, hipStream_t s);
// chunk merge + distance/ratio tests -> candidate lists (+ raw 2-NN outputs)
void launch_sift_select(const SiftPair* pairs, int npairs, const int32_t* rr_d1, const int32_t* rr_j1,
                        const int32_t* rr_d2, MatchRule rule, SiftCandidates cand, SiftOutputs out,
                        hipStream_t s);
// mutual check: nearest image-0 row of every candidate's train row.  vitems (room for max_vitems =
// sum over pairs of ceil(nA / kVerifyBlock) entries) and *nvitems are device scratch: the list of
// candidate blocks is built on the device from the candidate counts.
void launch_sift_verify(const SiftPair* pairs, int npairs, SiftItem* vitems, int max_vitems, int* nvitems,
                        SiftCandidates cand, hipStream_t s);
// keeps the confirmed candidates, writes compact and dense outputs
void launch_sift_confirm(const SiftPair* pairs, int npairs, SiftCandidates cand, int mutual,
                         SiftOutputs out, hipStream_t s);
// counts[npairs] -> offsets[npairs+1] (exclusive scan), then slots -> tight arrays
// list_total (device, may be NULL): segments marked as match lists are copied up to that many words only
void launch_gather_words(const WordSegments& seg, uint32_t* dst, hipStream_t s, const int64_t* list_total = nullptr);

// ---- geometric verification of putative matches -----------------------------------------
struct GeomPair {
    const float* xy0;   // keypoints of image 0: n0 x (x, y)
    const float* xy1;
    int32_t n0, n1;
    int32_t n;          // putative matches of the pair
    int32_t pad;
    int64_t off;        // first match of the pair in the match arrays
};
struct GeomRule {
    float max_sampson;   // inlier threshold on the Sampson distance, pixels
    int hypotheses;      // samples per pair
    unsigned seed;
    int min_inliers;     // pairs with fewer inliers report none
    unsigned long long first_pair;   // sampling key of the call's first pair
};
void launch_geom_verify(const GeomPair* pairs, int npairs, int max_matches, const int32_t* idx0, const int32_t* idx1,
                        const GeomRule& rule, float4* coords, float* Fh, float* Fbest, int32_t* best_count,
                        float* Fout, int32_t* counts, int32_t* slot_i0, int32_t* slot_i1, SiftPair* pack_pairs,
                        hipStream_t s);

// relative pose per pair from its fundamental matrix (Fs[npairs][9]), the two cameras' intrinsics
// (cams[npairs][8] = fx, fy, cx, cy of image 0 then image 1) and its inlier matches (pairs[p].off / .n into
// idx0 / idx1): R[npairs][9], T[npairs][3] (X1 = R X0 + T, |T| = 1), ne[npairs] = matches in front of both
// cameras (0: no pose)
void launch_geom_pose(const GeomPair* pairs, int npairs, const float* cams, const float* Fs, const int32_t* idx0,
                      const int32_t* idx1, float* R, float* T, int32_t* ne, hipStream_t s);

// ---- image-pair shortlist from a visual vocabulary ----------------------------------------
// hist: [nimg][nwords] counts (zeroed by the caller); words / offsets: the tight nearest-row list of a
// submission whose pair p is (image first_image + p, vocabulary) with every row accepted.
void launch_bow_count(const int64_t* offsets, const int32_t* words, int first_image, int nimages, int nwords,
                      uint32_t* hist, hipStream_t s);
// Weights the histograms in place, fills scores [nimg][nimg], marks (zeroed by the caller, [nimg][nimg]
// bytes) the pairs chosen by some image's topk, and writes them in ascending (i, j) order: offsets[nimg]
// is the pair count.
void launch_bow_shortlist(uint32_t* hist, int nimg, int nwords, int topk, uint32_t* df, int32_t* scores,
                          uint8_t* mark, int32_t* counts, int64_t* offsets, int32_t* pairs, hipStream_t s);

// vocabulary training on u8 sets: first values (word w = row rows[w] of the set whose fragment-major
// data is frags[w]), sums of the rows of every word over a submission of (image, vocabulary) pairs with
// every row accepted (sums [nwords][128], counts [nwords], zeroed by the caller; 32-bit sums: at most 2^24
// rows in all), rounded means.
void launch_kmeans_seed(const int8_t* const* frags, const int32_t* rows, int nwords, uint8_t* words, hipStream_t s);
void launch_kmeans_accumulate(const SiftPair* pairs, int nimages, const int64_t* offsets, const int32_t* words,
                              int nwords, unsigned* sums, unsigned* counts, hipStream_t s);
void launch_kmeans_update(const unsigned* sums, const unsigned* counts, int nwords, uint8_t* words,
                          hipStream_t s);

void launch_pack(const SiftPair* pairs, int npairs, const int32_t* counts, int64_t* offsets,
                 const int32_t* slot_i0, const int32_t* slot_i1,
                 int32_t* tight_i0, int32_t* tight_i1, hipStream_t s);

}  // namespace vsm
