// float32 descriptor path (SuperPoint-256 style): declarations shared with the C-ABI layer.
#pragma once

#include <hip/hip_runtime.h>
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

inline int f32_dimp(int dim) { return dim <= 64 ? 64 : dim <= 128 ? 128 : 256; }

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
// Query side: j1/j2 (indices), k1/k2 (DOT: dot products; L2: squared distances), nA entries.
// Train side (optional): cidx/ck, nB entries.
void launch_f32_knn(const F32Problem& pr, int32_t* j1, int32_t* j2, float* k1, float* k2,
                    int32_t* cidx, float* ck, void* scratch, hipStream_t s);

}  // namespace vsm
