// float32 descriptor path (SuperPoint-256 style): declarations shared with the C-ABI layer.
#pragma once

#include <hip/hip_runtime.h>

#include "device_flags.h"
#include <stdint.h>

namespace vsm {

// Device layout of a float set ("fragment-major f32"): rows padded to kF32RowPad, the
// dimension padded with zeros to dimp in {64, 128, 256}; every 32-row tile holds dimp/8
// groups of 1024 bytes, and in group g lane l = h*32 + r owns the four floats
// x[r][8g + h], x[r][8g + 2 + h], x[r][8g + 4 + h], x[r][8g + 6 + h]: the operands of four
// consecutive v_mfma_f32_32x32x2_f32 steps (k = 2s + h), so the chain runs over k in
// ascending order.  nrm[row] = the k-ascending fmaf chain of x[k]*x[k]; the array has 64
// floats of slack after n_pad.
constexpr int kF32RowPad = 256;
constexpr int kF32StripRows = 256;  // queries per workgroup (8 waves * 32)

__host__ __device__ inline int f32_dimp(int dim) { return dim <= 64 ? 64 : dim <= 128 ? 128 : 256; }

struct F32Problem {
    const float* fragA;  // queries
    const float* nrmA;
    const float* fragB;  // database / train set
    const float* nrmB;
    int32_t nA, nB, nA_pad, nB_pad;
    int32_t dim;
    int32_t metric;
};

// bytes of scratch launch_f32_knn needs
size_t f32_scratch_bytes(const F32Problem& pr, bool want_col);

void launch_f32_transform(const float* src, int n, int n_pad, int dim, bool dim_major,
                          float* frag, float* nrm, hipStream_t s);
// Query side: j1/j2 (indices), k1/k2 (DOT: dot products; L2: the ranking key turned into a squared distance,
// before launch_f32_l2_finish replaces it by the difference form), nA entries.
// Train side (optional): cidx/ck, nB entries.
void launch_f32_knn(const F32Problem& pr, int32_t* j1, int32_t* j2, float* k1, float* k2,
                    int32_t* cidx, float* ck, void* scratch, hipStream_t s);

// ---- bf16 coarse pass + exact re-rank (bf16_kernels.hip) -------------------------------
// bf16 shadow of a float set: 32-row tiles of dimp/16 k-steps x 64 lanes x 8 bf16 (round to
// nearest even), lane h*32 + r holding k = 16 s + 8 h .. + 7 of row r: the operand order of
// v_mfma_f32_32x32x16_bf16.  maxnorm_sq = the largest squared row norm of the set.
constexpr int kCoarseCap = 128;  // survivors one (split, query, lane half) list can hold

// One allocation per set: [bf16 rows][256 B holding maxnorm_sq][init_l2: n_pad + 64 floats]
// [init_dot: n_pad + 64 floats][plain row-major f32 rows: n_pad x dimp].  The init arrays are the
// accumulator start values of the coarse kernel: -|b|^2 / 2 resp. 0 for real rows, -inf for padding
// rows.  The row-major copy serves the re-rank, which reads single rows (in the 32-row operand tiles
// a row is 64 pieces of 16 bytes in 64 different cache lines).
struct Bf16Shadow {
    const unsigned short* rows;
    const float* maxnorm_sq;
    const float* init_l2;
    const float* init_dot;
    const float* rows_f32;
};
inline size_t bf16_rows_bytes(int n_pad, int dim) { return (size_t)n_pad * f32_dimp(dim) * 2; }
inline size_t bf16_shadow_bytes(int n_pad, int dim)
{
    return bf16_rows_bytes(n_pad, dim) + 256 + 2 * ((size_t)n_pad + 64) * sizeof(float) +
           (size_t)n_pad * f32_dimp(dim) * sizeof(float);
}
inline Bf16Shadow bf16_shadow_view(const void* base, int n_pad, int dim)
{
    const char* p = static_cast<const char*>(base);
    Bf16Shadow v;
    v.rows = reinterpret_cast<const unsigned short*>(p);
    v.maxnorm_sq = reinterpret_cast<const float*>(p + bf16_rows_bytes(n_pad, dim));
    v.init_l2 = v.maxnorm_sq + 64;
    v.init_dot = v.init_l2 + n_pad + 64;
    v.rows_f32 = v.init_dot + n_pad + 64;
    return v;
}
void launch_bf16_transform(const float* src, int n, int n_pad, int dim, bool dim_major, void* shadow_base,
                           const float* nrm, hipStream_t s);

// One directed search of a submission: every row of set A against set B.  A submission is an array
// of these in device memory; the coarse and re-rank kernels find their problem from the block index.
struct CoarseProblem {
    F32Problem pr;
    const unsigned short* bfA;
    const unsigned short* bfB;
    const float* initB;         // accumulator start values of B for pr.metric
    const float* maxnormB_sq;
    const float* rowsA;         // plain row-major f32 rows (dimp floats each)
    const float* rowsB;
    int32_t nstrips, nsplits, tiles_per_split;
    int32_t blk0;               // first workgroup of the coarse grid that belongs to this problem
    int32_t top1;               // 1: only the best row of a query matters (j2 / k2 are then meaningless)
    float key_floor;            // dot ranking: rows whose exact score is not above it cannot matter (-inf: none)
    int64_t part_off;           // first survivor sublist (sublists: [split][query][lane half])
    int64_t out_off;            // first output row in j1 / j2 / k1 / k2
    // Optional query list: query slot q (< *qcount <= pr.nA) is row qlist[q] of A; parts, lists and outputs
    // are indexed by slot, and pr.nA / pr.nA_pad are then the capacity of the list, not A's row count.
    const int32_t* qlist;
    const int32_t* qcount;
    // Optional database list: position d (< *dbcount) of the search is row dblist[d] of B, ascending; tiles, list
    // entries and ties go by position, results name rows.  pr.nB / pr.nB_pad stay B's own (the capacity).
    const int32_t* dblist;
    const int32_t* dbcount;
};

struct CoarseTotals {
    int coarse_blocks = 0;
    int rerank_blocks = 0;   // query blocks of the largest problem (grid x of the re-rank)
    int max_nsplits = 0;     // most database splits any problem has (picks the re-rank kernel)
    int64_t sublists = 0, out_rows = 0;
    int64_t out_rows0 = 0;   // first output row of the group (out_rows - out_rows0 rows belong to it)
};

// Fills the geometry and offsets of n problems (pr, bf*, initB, maxnormB_sq already set) so that the
// whole submission has about `target_wgs` coarse workgroups when the databases allow it.
// out_rows0: first output row (a second group of problems of the same submission continues after the first).
CoarseTotals plan_coarse(CoarseProblem* probs, int n, int target_wgs, int64_t out_rows0 = 0);
size_t coarse_scratch_bytes(const CoarseTotals& t);
// All problems must share f32_dimp(dim), top1 and whether they carry a database list.  Outputs as launch_f32_knn's query side at out_off, bit for
// bit, PROVIDED *overflow (device int, cleared by the caller) reads 0 afterwards; a set flag means a
// survivor list was too short and the results are void.
// Returns true when L2 results are already ordered and reported in difference form (launch_f32_l2_finish is
// then not needed), false when they are still keys.
// thr_cleared: the caller has cleared the words the splits of a search share (coarse_shared_thr) itself.
// by_args / args_block (one-pair calls, no database lists): the coarse kernel takes the searches from its own
// arguments and writes the pair's records to args_block (the submission's block, d_probs = args_block + 256) for the
// kernels behind it, instead of finding them there.
struct F32PlanArgs;
bool launch_bf16_batch(const CoarseProblem* d_probs, int nprobs, int dim, const CoarseTotals& t,
                       int32_t* j1, int32_t* j2, float* k1, float* k2, void* scratch, int* overflow,
                       hipStream_t s, bool top1 = false, bool db_lists = false, bool thr_cleared = false,
                       const F32PlanArgs* by_args = nullptr, char* args_block = nullptr);
// Where those words lie in the scratch of a group of searches, and how many there are (0: no search is split).
unsigned* coarse_shared_thr(void* scratch, const CoarseTotals& t, int* words);

// L2 searches only (problems of the other metric are skipped): orders each query's two best rows by the
// difference-form squared distance and stores those distances in k1 / k2.  Problems come from device memory
// (d_probs[nprobs]) or, with `single`, as one problem by value; max_rows = most query rows of a problem.
void launch_f32_l2_finish(const CoarseProblem* d_probs, int nprobs, int max_rows, const CoarseProblem* single,
                          int32_t* j1, int32_t* j2, float* k1, float* k2, hipStream_t s);

}  // namespace vsm
