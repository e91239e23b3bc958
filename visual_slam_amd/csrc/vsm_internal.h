// Internal declarations shared by the HIP kernels and the C-ABI layer.
#pragma once

#include <hip/hip_runtime.h>
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
//   nrm8[row] = (sum of squared int8 values) << 8, padding rows kNormPad.
// Squared L2 of two u8 descriptors = nrm(a) + nrm(b) - 2*dot(int8 a, int8 b), exact in
// int32 (< 2^23).
constexpr int kDescDim = 128;
constexpr int kRowPad = 256;
constexpr int kTileBytes = 4096;      // 32 rows * 128 bytes
constexpr int kStripRows = 256;       // query rows per workgroup (8 waves * 32)
constexpr int kStepCols = 128;        // train columns per main-loop step
constexpr int kStepsPerChunk = 64;    // 8-bit step/column-block index: 64 steps * 4 blocks
constexpr int kChunkCols = kStepCols * kStepsPerChunk;  // 8192
constexpr int32_t kNormPad = 0x7FFFFF00;
constexpr int32_t kDistPad = 0x7FFFFF;  // (kNormPad >> 8): "no such neighbour"

struct SiftPair {
    const int8_t* fragA;
    const int32_t* nrmA;
    const int8_t* fragB;
    const int32_t* nrmB;
    int32_t nA, nB;          // real row counts
    int32_t nA_pad, nB_pad;  // padded to kRowPad
    int32_t nstrips, nchunks;
    int64_t row_off;   // into the per-(chunk,row) partial arrays: nchunks * nA_pad entries
    int64_t col_off;   // into the per-(strip,col) partial array: nstrips * nB_pad entries
    int64_t out0_off;  // into per-query-row outputs (nA entries)
    int64_t out1_off;  // into per-train-row outputs (nB entries)
};

struct SiftItem {
    int32_t pair;
    int16_t strip;
    int16_t chunk;
};

struct MatchRule {
    int64_t num2, den2;   // ratio^2 as num2/den2; num2 <= 0 disables
    int64_t max_dist;     // < 0 disables
    int32_t mutual;
};

struct SiftOutputs {
    // compacted matches, per pair at out0_off (capacity nA)
    int32_t* slot_i0;
    int32_t* slot_i1;
    int32_t* slot_d;
    int32_t* counts;   // [npairs]
    // optional dense forms
    int32_t* dense0;   // at out0_off, nA entries
    int32_t* dense1;   // at out1_off, nB entries (pre-filled with -1 by the kernel)
    // optional raw 2-NN
    int32_t* knn_j1;
    int32_t* knn_d1;
    int32_t* knn_d2;
};

void launch_sift_transform(const uint8_t* des, int n, int n_pad, int8_t* frag, int32_t* nrm8,
                           hipStream_t s);
void launch_sift_knn(const SiftPair* pairs, const SiftItem* items, int nitems,
                     int32_t* rr_d1, int32_t* rr_j1, int32_t* rr_d2, int32_t* colpart,
                     hipStream_t s);
void launch_sift_finalize(const SiftPair* pairs, int npairs,
                          const int32_t* rr_d1, const int32_t* rr_j1, const int32_t* rr_d2,
                          const int32_t* colpart, MatchRule rule, SiftOutputs out, hipStream_t s);
// counts[npairs] -> offsets[npairs+1] (exclusive scan), then slots -> tight arrays
void launch_pack(const SiftPair* pairs, int npairs, const int32_t* counts, int64_t* offsets,
                 const int32_t* slot_i0, const int32_t* slot_i1,
                 int32_t* tight_i0, int32_t* tight_i1, hipStream_t s);

}  // namespace vsm
