// float32 descriptors: brute-force 2-NN (dot-product or squared-L2 ranking) on the f32-input
// MFMA of gfx950, which is bit-for-bit a k-ascending single-precision fma chain, so every
// score equals the scalar definition s = fmaf(a[k], b[k], s) exactly.
//
// Semantics follow the reference's dense score + mutual argmax block
// (Others/SuperPoint_data/models/superglue.py:267-286, first index on ties) and its
// brute-force L2 nearest-code search (Graph_Idea/model/model_2_0.py:43-78).
#include "f32_kernels.h"

#include <math.h>

namespace vsm {

typedef float v16f __attribute__((ext_vector_type(16)));

#define VSM_LDS __attribute__((address_space(3)))
#define VSM_GLB __attribute__((address_space(1)))

// ---------------------------------------------------------------------------------
// rows -> fragment-major f32 + norms
// ---------------------------------------------------------------------------------
__global__ void f32_transform_kernel(const float* __restrict__ src, int n, int n_pad, int dim, int dimp,
                                     int dim_major, float* __restrict__ frag)
{
    // one thread per (row, group, half)
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int groups = dimp / 8;
    const long long total = (long long)n_pad * groups * 2;
    if (gid >= total) return;
    const int h = (int)(gid & 1);
    const int g = (int)((gid >> 1) % groups);
    const int row = (int)((gid >> 1) / groups);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 8 * g + 2 * e + h;
        float x = 0.0f;
        if (row < n && k < dim) x = dim_major ? src[(size_t)k * n + row] : src[(size_t)row * dim + k];
        v[e] = x;
    }
    const int tile = row >> 5, r = row & 31;
    float4* dst = reinterpret_cast<float4*>(frag + (size_t)tile * 32 * dimp + (size_t)g * 256 + (h * 32 + r) * 4);
    *dst = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ void f32_norm_kernel(const float* __restrict__ src, int n, int n_pad, int dim, int dim_major,
                                float* __restrict__ nrm)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_pad + 64) return;
    float s = 0.0f;
    if (row < n) {
        for (int k = 0; k < dim; ++k) {
            const float x = dim_major ? src[(size_t)k * n + row] : src[(size_t)row * dim + k];
            s = fmaf(x, x, s);
        }
    }
    nrm[row] = s;
}

// The same norms for a row-major source with every byte read once: one wave takes 64 rows, brings them
// through LDS 64 columns at a time with coalesced loads, and lane r then walks row r's columns in
// ascending order (the same fmaf chain as above).
__global__ __launch_bounds__(64) void f32_norm_rows_kernel(const float* __restrict__ src, int n, int n_pad, int dim,
                                                           float* __restrict__ nrm)
{
    __shared__ float tile[64][65];
    const int row0 = blockIdx.x * 64, lane = threadIdx.x;
    float s = 0.0f;
    for (int k0 = 0; k0 < dim; k0 += 64) {
        const int kw = min(64, dim - k0);
#pragma unroll 8
        for (int r = 0; r < 64; ++r) {
            const int row = row0 + r;
            tile[r][lane] = (row < n && lane < kw) ? src[(size_t)row * dim + k0 + lane] : 0.0f;
        }
        __syncthreads();
        for (int k = 0; k < kw; ++k) {
            const float x = tile[lane][k];
            s = fmaf(x, x, s);
        }
        __syncthreads();
    }
    const int row = row0 + lane;
    if (row < n_pad + 64) nrm[row] = row < n ? s : 0.0f;
}

void launch_f32_transform(const float* src, int n, int n_pad, int dim, bool dim_major,
                          float* frag, float* nrm, hipStream_t s)
{
    const int dimp = f32_dimp(dim);
    const long long total = (long long)n_pad * (dimp / 8) * 2;
    hipLaunchKernelGGL(f32_transform_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       src, n, n_pad, dim, dimp, dim_major ? 1 : 0, frag);
    if (dim_major)
        hipLaunchKernelGGL(f32_norm_kernel, dim3((n_pad + 64 + 255) / 256), dim3(256), 0, s, src, n, n_pad, dim, 1, nrm);
    else
        hipLaunchKernelGGL(f32_norm_rows_kernel, dim3((n_pad + 64 + 63) / 64), dim3(64), 0, s, src, n, n_pad, dim, nrm);
}

// ---------------------------------------------------------------------------------
// main kernel
// ---------------------------------------------------------------------------------
// Workgroup = 8 waves; wave w keeps the MFMA operand of 32 queries in registers (the
// accumulator's lane axis), the database streams through LDS one 32-row tile at a time
// (LDS-DMA, double buffered).  A lane therefore sees the database rows of its query in
// ascending order and keeps the running two best in four registers.
struct Top2 {
    float s1, s2;
    int j1, j2;
};

__device__ __forceinline__ bool better(float x, int jx, float y, int jy)
{
    return x > y || (x == y && (unsigned)jx < (unsigned)jy);
}

__device__ __forceinline__ Top2 merge_top2(const Top2& a, const Top2& b)
{
    Top2 r;
    if (better(a.s1, a.j1, b.s1, b.j1)) {
        r.s1 = a.s1; r.j1 = a.j1;
        if (better(a.s2, a.j2, b.s1, b.j1)) { r.s2 = a.s2; r.j2 = a.j2; }
        else { r.s2 = b.s1; r.j2 = b.j1; }
    } else {
        r.s1 = b.s1; r.j1 = b.j1;
        if (better(a.s1, a.j1, b.s2, b.j2)) { r.s2 = a.s1; r.j2 = a.j1; }
        else { r.s2 = b.s2; r.j2 = b.j2; }
    }
    return r;
}

struct QPart {  // 16 bytes
    float s1, s2;
    int j1, j2;
};
struct CPart {  // 8 bytes
    float s;
    int i;
};

template <int DIMP>
__global__ __launch_bounds__(512, 2)
void f32_knn_kernel(F32Problem pr, int nstrips, int tiles_per_split,
                    QPart* __restrict__ qpart, CPart* __restrict__ colpart)
{
    constexpr int G = DIMP / 8;
    constexpr int TILE_FLOATS = 32 * DIMP;
    constexpr int TILE_BYTES = TILE_FLOATS * 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [0, 2*TILE_BYTES) database tiles, then 2 * 256 bytes of norms

    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r8 = nb & 7, xcd = bid & 7;
    const int L = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int split = L / nstrips, strip = L % nstrips;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane5 = lane & 31, half = lane >> 5;
    const bool l2 = pr.metric == 1;

    const int qtile = strip * 8 + wave;
    const int qidx = qtile * 32 + lane5;
    const bool qvalid = qidx < pr.nA;
    const float na = pr.nrmA[qidx];
    float4 qf[G];
    {
        const float4* src = reinterpret_cast<const float4*>(pr.fragA + (size_t)qtile * TILE_FLOATS) + lane;
#pragma unroll
        for (int g = 0; g < G; ++g) qf[g] = src[g * 64];
    }

    const int ntiles = pr.nB_pad >> 5;
    const int t0 = split * tiles_per_split;
    const int t1 = min(t0 + tiles_per_split, ntiles);

    Top2 st;
    st.s1 = -INFINITY; st.s2 = -INFINITY; st.j1 = -1; st.j2 = -1;

    auto stage = [&](int buf, int t) {
        const char* src = reinterpret_cast<const char*>(pr.fragB + (size_t)t * TILE_FLOATS) + lane * 16;
        char* dst = smem + buf * TILE_BYTES;
#pragma unroll
        for (int p = 0; p < (G + 7) / 8; ++p) {
            const int piece = wave + p * 8;
            if (piece < G)
                __builtin_amdgcn_global_load_lds((const VSM_GLB void*)(src + piece * 1024),
                                                 (VSM_LDS void*)(dst + piece * 1024), 16, 0, 0);
        }
        if (wave == 0)
            __builtin_amdgcn_global_load_lds((const VSM_GLB void*)(pr.nrmB + t * 32 + lane),
                                             (VSM_LDS void*)(smem + 2 * TILE_BYTES + buf * 256), 4, 0, 0);
    };

    if (t0 < t1) stage(0, t0);
    __syncthreads();

    for (int t = t0; t < t1; ++t) {
        const int cur = (t - t0) & 1;
        if (t + 1 < t1) stage(cur ^ 1, t + 1);
        const float4* At = reinterpret_cast<const float4*>(smem + cur * TILE_BYTES) + lane;
        const float* nbl = reinterpret_cast<const float*>(smem + 2 * TILE_BYTES + cur * 256);

        v16f acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 af = At[g * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, qf[g].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, qf[g].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, qf[g].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, qf[g].w, acc, 0, 0, 0);
        }

#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int rin = (reg & 3) + 8 * (reg >> 2) + 4 * half;
            const int row = t * 32 + rin;
            const float d = acc[reg];
            float s = l2 ? fmaf(2.0f, d, -nbl[rin]) : d;
            if (row >= pr.nB) s = -INFINITY;
            if (s > st.s1) {
                st.s2 = st.s1; st.j2 = st.j1;
                st.s1 = s; st.j1 = row;
            } else if (s > st.s2) {
                st.s2 = s; st.j2 = row;
            }
            if (colpart) {
                // best query of this database row among the wave's 32 queries
                float sp = l2 ? fmaf(2.0f, d, -na) : d;
                if (!qvalid) sp = -INFINITY;
                int qi = qidx;
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) {
                    const float os = __shfl_xor(sp, m);
                    const int oi = __shfl_xor(qi, m);
                    if (better(os, oi, sp, qi)) { sp = os; qi = oi; }
                }
                if (lane5 == 0) {
                    CPart cp;
                    cp.s = sp; cp.i = qi;
                    colpart[(size_t)qtile * pr.nB_pad + row] = cp;
                }
            }
        }
        __syncthreads();
    }

    // the two halves of the wave hold disjoint database rows of the same queries
    Top2 o;
    o.s1 = __shfl_xor(st.s1, 32); o.s2 = __shfl_xor(st.s2, 32);
    o.j1 = __shfl_xor(st.j1, 32); o.j2 = __shfl_xor(st.j2, 32);
    const Top2 m = merge_top2(st, o);
    if (half == 0) {
        QPart qp;
        qp.s1 = m.s1; qp.s2 = m.s2; qp.j1 = m.j1; qp.j2 = m.j2;
        qpart[(size_t)split * pr.nA_pad + qidx] = qp;
    }
}

__global__ void f32_query_merge_kernel(F32Problem pr, int nsplits, const QPart* __restrict__ qpart,
                                       int32_t* __restrict__ j1, int32_t* __restrict__ j2,
                                       float* __restrict__ k1, float* __restrict__ k2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pr.nA) return;
    Top2 m;
    m.s1 = -INFINITY; m.s2 = -INFINITY; m.j1 = -1; m.j2 = -1;
    for (int s = 0; s < nsplits; ++s) {
        const QPart qp = qpart[(size_t)s * pr.nA_pad + i];
        Top2 o;
        o.s1 = qp.s1; o.s2 = qp.s2; o.j1 = qp.j1; o.j2 = qp.j2;
        m = merge_top2(m, o);
    }
    const bool l2 = pr.metric == 1;
    const float na = pr.nrmA[i];
    j1[i] = m.j1;
    j2[i] = m.j2;
    k1[i] = m.j1 < 0 ? (l2 ? INFINITY : -INFINITY) : (l2 ? fmaxf(-m.s1 + na, 0.0f) : m.s1);
    k2[i] = m.j2 < 0 ? (l2 ? INFINITY : -INFINITY) : (l2 ? fmaxf(-m.s2 + na, 0.0f) : m.s2);
}

__global__ void f32_col_merge_kernel(F32Problem pr, int nqtiles, const CPart* __restrict__ colpart,
                                     int32_t* __restrict__ cidx, float* __restrict__ ck)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= pr.nB) return;
    float bs = -INFINITY;
    int bi = -1;
    for (int t = 0; t < nqtiles; ++t) {
        const CPart cp = colpart[(size_t)t * pr.nB_pad + j];
        if (better(cp.s, cp.i, bs, bi)) { bs = cp.s; bi = cp.i; }
    }
    const bool l2 = pr.metric == 1;
    cidx[j] = bi;
    ck[j] = bi < 0 ? (l2 ? INFINITY : -INFINITY) : (l2 ? fmaxf(-bs + pr.nrmB[j], 0.0f) : bs);
}

static void f32_geometry(const F32Problem& pr, int* nstrips, int* nsplits, int* tiles_per_split)
{
    const int ntiles = pr.nB_pad / 32;
    *nstrips = pr.nA_pad / kF32StripRows;
    int want = (512 + *nstrips - 1) / *nstrips;  // aim at >= 2 workgroups per CU's worth of work
    if (want > ntiles) want = ntiles;
    if (want < 1) want = 1;
    *tiles_per_split = (ntiles + want - 1) / want;
    *nsplits = (ntiles + *tiles_per_split - 1) / *tiles_per_split;
}

size_t f32_scratch_bytes(const F32Problem& pr, bool want_col)
{
    int nstrips, nsplits, tps;
    f32_geometry(pr, &nstrips, &nsplits, &tps);
    size_t b = (size_t)nsplits * pr.nA_pad * sizeof(QPart);
    if (want_col) b += (size_t)(pr.nA_pad / 32) * pr.nB_pad * sizeof(CPart);
    return b + 256;
}

template <int DIMP>
static void launch_knn_t(const F32Problem& pr, int nstrips, int nsplits, int tps, QPart* qpart, CPart* colpart,
                         hipStream_t s)
{
    const int lds = 2 * 32 * DIMP * 4 + 512;
    static DeviceFlags attr_set;   // the attribute is per device
    ensure_dynamic_lds(attr_set, reinterpret_cast<const void*>(f32_knn_kernel<DIMP>), lds);
    hipLaunchKernelGGL(f32_knn_kernel<DIMP>, dim3(nstrips * nsplits), dim3(512), lds, s,
                       pr, nstrips, tps, qpart, colpart);
}

void launch_f32_knn(const F32Problem& pr, int32_t* j1, int32_t* j2, float* k1, float* k2,
                    int32_t* cidx, float* ck, void* scratch, hipStream_t s)
{
    int nstrips, nsplits, tps;
    f32_geometry(pr, &nstrips, &nsplits, &tps);
    QPart* qpart = static_cast<QPart*>(scratch);
    CPart* colpart = nullptr;
    if (cidx) {
        size_t off = ((size_t)nsplits * pr.nA_pad * sizeof(QPart) + 255) & ~(size_t)255;
        colpart = reinterpret_cast<CPart*>(static_cast<char*>(scratch) + off);
    }
    const int dimp = f32_dimp(pr.dim);
    if (dimp == 64) launch_knn_t<64>(pr, nstrips, nsplits, tps, qpart, colpart, s);
    else if (dimp == 128) launch_knn_t<128>(pr, nstrips, nsplits, tps, qpart, colpart, s);
    else launch_knn_t<256>(pr, nstrips, nsplits, tps, qpart, colpart, s);
    hipLaunchKernelGGL(f32_query_merge_kernel, dim3((pr.nA + 255) / 256), dim3(256), 0, s,
                       pr, nsplits, qpart, j1, j2, k1, k2);
    if (cidx)
        hipLaunchKernelGGL(f32_col_merge_kernel, dim3((pr.nB + 255) / 256), dim3(256), 0, s,
                           pr, pr.nA_pad / 32, colpart, cidx, ck);
}

// ---------------------------------------------------------------------------------
// L2 searches: the two best rows ordered, and their distances reported, in difference form
// ---------------------------------------------------------------------------------
// The matrix cores rank by the key |b|^2 - 2 a.b, whose rounding error (about 1e-7 of |a|^2 + |b|^2)
// exceeds the squared distance of near-duplicates.  The reference ranks by the norm of the difference
// (Graph_Idea/model/model_2_0.py:68), which has no cancellation.  So the two best rows are FOUND by the key
// and then ordered by d = the k-ascending fmaf chain of (a[k] - b[k])^2 (lower row first on equal d); k1 and
// k2 become those distances.  One thread per query; rows come from the plain row-major copies.
__global__ void f32_l2_finish_kernel(const CoarseProblem* __restrict__ probs, CoarseProblem single, int use_single,
                                     int32_t* __restrict__ j1, int32_t* __restrict__ j2,
                                     float* __restrict__ k1, float* __restrict__ k2)
{
    const CoarseProblem& p = use_single ? single : probs[blockIdx.y];
    if (p.pr.metric != 1) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // query slot
    if (i >= (p.qlist ? *p.qcount : p.pr.nA)) return;
    const size_t o = (size_t)p.out_off + i;
    int a1 = j1[o], a2 = j2[o];
    if (a1 < 0) return;
    const int dimp = f32_dimp(p.pr.dim);
    const float4* qa = reinterpret_cast<const float4*>(p.rowsA + (size_t)(p.qlist ? p.qlist[i] : i) * dimp);
    const float4* r1 = reinterpret_cast<const float4*>(p.rowsB + (size_t)a1 * dimp);
    const float4* r2 = reinterpret_cast<const float4*>(p.rowsB + (size_t)(a2 < 0 ? a1 : a2) * dimp);
    float d1 = 0.0f, d2 = 0.0f;
    for (int g = 0; g < dimp / 4; ++g) {     // the padding columns are zero in every row: they add exact zeros
        const float4 a = qa[g], x = r1[g], y = r2[g];
        float t;
        t = a.x - x.x; d1 = fmaf(t, t, d1);
        t = a.y - x.y; d1 = fmaf(t, t, d1);
        t = a.z - x.z; d1 = fmaf(t, t, d1);
        t = a.w - x.w; d1 = fmaf(t, t, d1);
        t = a.x - y.x; d2 = fmaf(t, t, d2);
        t = a.y - y.y; d2 = fmaf(t, t, d2);
        t = a.z - y.z; d2 = fmaf(t, t, d2);
        t = a.w - y.w; d2 = fmaf(t, t, d2);
    }
    if (a2 < 0) {           // a database of one row has no runner-up
        k1[o] = d1;
        return;
    }
    if (d2 < d1 || (d2 == d1 && a2 < a1)) {
        const float t = d1; d1 = d2; d2 = t;
        const int u = a1; a1 = a2; a2 = u;
    }
    j1[o] = a1;
    j2[o] = a2;
    k1[o] = d1;
    k2[o] = d2;
}

void launch_f32_l2_finish(const CoarseProblem* d_probs, int nprobs, int max_rows, const CoarseProblem* single,
                          int32_t* j1, int32_t* j2, float* k1, float* k2, hipStream_t s)
{
    if (nprobs <= 0 || max_rows <= 0) return;
    const dim3 grid((max_rows + 255) / 256, nprobs);
    hipLaunchKernelGGL(f32_l2_finish_kernel, grid, dim3(256), 0, s, d_probs, single ? *single : CoarseProblem{},
                       single ? 1 : 0, j1, j2, k1, k2);
}

}  // namespace vsm
