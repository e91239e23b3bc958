// Fast path for float descriptors: a bf16-MFMA coarse pass that keeps a PROVABLE superset of the
// two best rows of every query, followed by an exact single-precision re-rank of the few survivors.
// The result equals the exact-f32 path bit for bit (the re-rank evaluates the same k-ascending fmaf
// chain, and both of the true best rows are guaranteed to be among the survivors).
//
// Error bound.  With a^ = bf16(a), b^ = bf16(b) (round to nearest: |x^ - x| <= 2^-9 |x|) the coarse
// dot product differs from the real one by at most (2^-8 + 2^-18) sum|a_k b_k| <= 2^-8 (1 + 2^-10)
// |a||b|; f32 accumulation of 256 bf16 products and the f32 rounding of the exact chain add at most
// 3 * 256 * 2^-24 |a||b|.  eps(i) = kCoarseEps * |a_i| * max_j |b_j| covers both with room to spare.
// Let T2 be the second largest coarse score of query i.  Every row whose exact score is at least the
// exact runner-up has coarse score >= T2 - 2 eps.  The coarse pass appends every row that beats the
// RUNNING T2 (of the rows one lane has seen) minus that margin, which only grows, so the lists hold
// a superset; the re-rank filters with the final T2.  Both T2s are taken over the per-(lane, tile) maxima
// (16 rows each) instead of over the rows: never more than the T2 over rows, so the thresholds are lower
// bounds of the ones the argument needs, and the epilogue is one max3 per two scores plus three instructions
// per tile (6.43 -> 6.22 ms on the 16384 x 1M query against the two-largest-over-rows form).  For the L2 ranking the coarse keys are
// dot - |b|^2 / 2 (half the exact key; the accumulators start from -|b|^2 / 2), which adds the f32
// rounding of partial sums of that size: 2 * 256 * 2^-24 * max|b|^2 / 2 on top of 2 eps.
// Inputs are assumed finite and far from the ends of the f32 exponent range.
//
// Searches cut into several database splits share their thresholds: the splits of a query run side by side, each
// with the running runner-up (top1: best) of its own rows only, which is young for a good part of a short split.
// After tiles 4, 8, 16 and every 32 of its split a lane reads a word per query and, when its own value is better, publishes it
// (atomic maximum over an order-preserving integer image of the float): any split's running runner-up is at most the
// final T2, so (that value - margin) is a floor no survivor lies under.  Which entries get appended then depends on
// timing; the survivors of the final threshold, and with them the results, do not.
//
// Two reductions of the work for callers that need less than the two best rows of every query:
//  * top1: only the best row matters.  The threshold is then T1 - 2 eps (T1 = the largest coarse score): the
//    exact best row scores at least what the coarse-best row scores exactly, which is >= T1 - eps, so its own
//    coarse score is >= T1 - 2 eps; rows that tie with it exactly satisfy the same bound.
//  * key_floor (dot ranking): a row whose exact score is not above the floor cannot be accepted by the caller's
//    score test, and a row above it has coarse score > floor - eps; the threshold never drops below that.
//    Queries whose best score is not above the floor come out with no row at all (-1).
// And a query list (qlist / qcount) restricts a search to some rows of A, chosen on the device by an earlier
// kernel: the mutual check of a pair only asks about the train rows that some query row has chosen.
// A database list (dblist / dbcount) restricts it to some rows of B the same way: under a pure threshold rule a
// row that could come before a candidate in its column is a candidate itself (its own best score is at least
// its score in that column, which is at least the candidate's), so the reverse searches of a mutual check only
// look at the candidates' rows.  Each lane of a staging wave fetches the 16-byte pieces of ITS row out of
// whichever tile holds it; positions past the end of the list take the start value of a padding row (-inf).
#include "f32_kernels.h"
#include "vsm_internal.h"

#include <math.h>

#include <algorithm>
#include <type_traits>

namespace vsm {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

#define VSM_LDS __attribute__((address_space(3)))
#define VSM_GLB __attribute__((address_space(1)))

constexpr float kCoarseEps = 0.0041f;

__device__ __forceinline__ float coarse_margin(bool l2, float na_sq, float maxnb_sq)
{
    const float eps = kCoarseEps * sqrtf(na_sq) * sqrtf(maxnb_sq);
    return l2 ? 2.0f * eps + 3.1e-5f * maxnb_sq : 2.0f * eps;
}

// ---------------------------------------------------------------------------------
// f32 rows -> bf16 operand-order shadow (+ largest row norm of the set)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f32_to_bf16_rne(float x)
{
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

__global__ void bf16_transform_kernel(const float* __restrict__ src, int n, int n_pad, int dim, int dimp,
                                      int dim_major, unsigned short* __restrict__ shadow,
                                      float* __restrict__ rows_f32)
{
    // one thread per (row, k-step, half): 8 consecutive k
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int steps = dimp / 16;
    const long long total = (long long)n_pad * steps * 2;
    if (gid >= total) return;
    const int h = (int)(gid & 1);
    const int s = (int)((gid >> 1) % steps);
    const int row = (int)((gid >> 1) / steps);
    unsigned v[8];
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * s + 8 * h + j;
        float x = 0.0f;
        if (row < n && k < dim) x = dim_major ? src[(size_t)k * n + row] : src[(size_t)row * dim + k];
        f[j] = x;
        v[j] = f32_to_bf16_rne(x);
    }
    float4* plain = reinterpret_cast<float4*>(rows_f32 + (size_t)row * dimp + 16 * s + 8 * h);
    plain[0] = make_float4(f[0], f[1], f[2], f[3]);
    plain[1] = make_float4(f[4], f[5], f[6], f[7]);
    const int tile = row >> 5, r = row & 31;
    uint4 out;
    out.x = v[0] | (v[1] << 16);
    out.y = v[2] | (v[3] << 16);
    out.z = v[4] | (v[5] << 16);
    out.w = v[6] | (v[7] << 16);
    *reinterpret_cast<uint4*>(shadow + ((size_t)tile * steps + s) * 512 + (h * 32 + r) * 8) = out;
}

__global__ void max_norm_kernel(const float* __restrict__ nrm, int n, int n_slots, float* __restrict__ out_maxnorm,
                                float* __restrict__ init_l2, float* __restrict__ init_dot)
{
    float m = 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += gridDim.x * blockDim.x) {
        const bool real = i < n;
        const float v = real ? nrm[i] : 0.0f;
        m = fmaxf(m, v);
        init_l2[i] = real ? -0.5f * v : -INFINITY;
        init_dot[i] = real ? 0.0f : -INFINITY;
    }
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    // squared norms are >= 0, so their bit patterns order like unsigned integers
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(out_maxnorm), __float_as_uint(m));
}

void launch_bf16_transform(const float* src, int n, int n_pad, int dim, bool dim_major, void* shadow_base,
                           const float* nrm, hipStream_t s)
{
    const Bf16Shadow v = bf16_shadow_view(shadow_base, n_pad, dim);
    unsigned short* shadow = const_cast<unsigned short*>(v.rows);
    float* maxnorm_sq = const_cast<float*>(v.maxnorm_sq);
    const int dimp = f32_dimp(dim);
    const long long total = (long long)n_pad * (dimp / 16) * 2;
    hipLaunchKernelGGL(bf16_transform_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       src, n, n_pad, dim, dimp, dim_major ? 1 : 0, shadow, const_cast<float*>(v.rows_f32));
    (void)hipMemsetAsync(maxnorm_sq, 0, sizeof(float), s);
    const int slots = n_pad + 64;
    hipLaunchKernelGGL(max_norm_kernel, dim3(std::min(64, (slots + 255) / 256)), dim3(256), 0, s, nrm, n, slots,
                       maxnorm_sq, const_cast<float*>(v.init_l2), const_cast<float*>(v.init_dot));
}

// ---------------------------------------------------------------------------------
// coarse pass
// ---------------------------------------------------------------------------------
struct CoarsePart {   // per (split, query, lane half): coarse top-2 values and list length
    float t1, t2;
    int count;
    int pad;
};

// Workgroup = 4 waves; wave w keeps the bf16 operands of kCoarseRB x 32 queries in registers (the
// accumulator's lane axis), the database streams through LDS in 16 KiB steps (LDS-DMA, double
// buffered) and every tile read from LDS feeds kCoarseRB MFMA chains.
// Survivor lists are stored 64 at a time, entry-major: the e-th entries of 64 consecutive lists (the
// 32 queries x 2 lane halves one wave of the coarse kernel serves) are adjacent, so the part of the
// list memory that is written and read back is dense.
__device__ __forceinline__ size_t list_slot(size_t sublist, int e)
{
    return (sublist >> 6) * (size_t)(64 * kCoarseCap) + (size_t)e * 64 + (sublist & 63);
}

constexpr int kCoarseRB = 2;
constexpr int kCoarseWaves = 4;
constexpr int kCoarseStripRows = kCoarseWaves * kCoarseRB * 32;  // queries per workgroup
constexpr int kCoarseStepBytes = 16384;
static_assert(kF32RowPad % kCoarseStripRows == 0, "query strips must tile the padded row count");

// The accumulators start from a per-row constant instead of zero: -|b|^2 / 2 for the L2 ranking (the
// coarse key is then dot - |b|^2 / 2, half the exact key), 0 for the dot ranking, and -inf for the
// padding rows past the end of the set, which therefore never rank.  The kernel itself is the same
// for both rankings.
// index of the problem whose block range holds block L (first-block offsets ascending)
template <class GetFirst>
__device__ __forceinline__ int find_problem(int nprobs, int L, GetFirst first)
{
    int lo = 0, hi = nprobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first(mid) <= L) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// order-preserving image of a float in an unsigned integer (0 is below every float, including -inf)
__device__ __forceinline__ unsigned float_order_bits(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_of_order_bits(unsigned e)
{
    return e == 0 ? -INFINITY : __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

// position of a workgroup in the list of (problem, split, strip) items: an XCD gets a contiguous run of them
__device__ __forceinline__ int coarse_item_of_block()
{
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}

template <int DIMP, bool TOP1, bool DBL>
__device__ __forceinline__ void coarse_body(const CoarseProblem& P, const int L,
                                            CoarsePart* __restrict__ part_all, uint2* __restrict__ lists_all,
                                            int* __restrict__ overflow, unsigned* __restrict__ shared_thr, int64_t out_rows0)
{
    constexpr int S = DIMP / 16;             // k-steps (MFMAs) per tile
    constexpr int RB = kCoarseRB;
    constexpr int TILE_BYTES = 32 * DIMP * 2;
    constexpr int STEP_TILES = kCoarseStepBytes / TILE_BYTES;   // 1, 2 or 4
    constexpr int STEP_ROWS = STEP_TILES * 32;
    // database steps, double buffered; the two buffers and the two constant arrays are addressed with
    // compile-time indices so that reads of one are known not to touch the DMA target of the other
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) float s_init0[256];
    __shared__ __attribute__((aligned(16))) float s_init1[256];

    const F32Problem pr = P.pr;
    const unsigned short* __restrict__ bfA = P.bfA;
    const unsigned short* __restrict__ bfB = P.bfB;
    const float* __restrict__ initB = P.initB;
    const int nstrips = P.nstrips, tiles_per_split = P.tiles_per_split;
    CoarsePart* __restrict__ part = part_all + P.part_off;
    uint2* __restrict__ lists = lists_all + P.part_off * kCoarseCap;
    const int Lp = L - P.blk0;
    const int split = Lp / nstrips, strip = Lp % nstrips;
    const int32_t* __restrict__ qlist = P.qlist;
    const int nq = qlist ? *P.qcount : pr.nA;
    if (strip * kCoarseStripRows >= nq) return;   // the whole strip is past the end of the query list

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane5 = lane & 31, half = lane >> 5;
    const bool l2 = pr.metric == 1;
    const float maxnb = *P.maxnormB_sq;

    // persistent operands: this wave's RB x 32 queries
    v8bf qf[RB][S];
    float margin[RB], floorc[RB], thr[RB], b1[RB], b2[RB];
    int cnt[RB];
    bool live[RB];
    const int qidx0 = ((strip * kCoarseWaves + wave) * RB) * 32 + lane5;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int qi = qidx0 + rb * 32;               // query slot
        const bool real = qi < nq;
        int qrow = qi;
        if (qlist) {
            // the operands of this lane's own row, out of whichever tile holds it
            qrow = real ? qlist[qi] : 0;
            const v8bf* src = reinterpret_cast<const v8bf*>(bfA + (size_t)(qrow >> 5) * S * 512) + half * 32 + (qrow & 31);
#pragma unroll
            for (int s = 0; s < S; ++s) qf[rb][s] = src[s * 64];
        } else {
            const int qtile = (strip * kCoarseWaves + wave) * RB + rb;
            const v8bf* src = reinterpret_cast<const v8bf*>(bfA + (size_t)qtile * S * 512) + lane;
#pragma unroll
            for (int s = 0; s < S; ++s) qf[rb][s] = src[s * 64];
        }
        margin[rb] = coarse_margin(l2, pr.nrmA[qrow], maxnb);
        floorc[rb] = l2 ? -INFINITY : P.key_floor - 0.5f * margin[rb];
        // padded queries never append
        thr[rb] = real ? -INFINITY : INFINITY;
        if (!real) {
            margin[rb] = -INFINITY;
            floorc[rb] = INFINITY;
        }
        b1[rb] = -INFINITY;
        b2[rb] = -INFINITY;
        cnt[rb] = 0;
        live[rb] = real;
    }
    // one word per query slot, shared by the splits of the search (cleared by the launcher); unused with one split
    unsigned* __restrict__ shared_q = shared_thr && P.nsplits > 1 ? shared_thr + (P.out_off - out_rows0) + qidx0 : nullptr;

    const int32_t* __restrict__ dblist = P.dblist;
    const int ndb = DBL ? *P.dbcount : 0;
    // whole steps either way: multiples of 8 tiles without a list, the listed rows rounded up to a step with one
    const int ntiles = DBL ? (ndb + STEP_ROWS - 1) / STEP_ROWS * STEP_TILES : pr.nB_pad >> 5;
    const int t0 = split * tiles_per_split;
    const int t1e = min(t0 + tiles_per_split, ntiles);
    const int nsteps = max(t1e - t0, 0) / STEP_TILES;

    // database list: the rows at this lane's positions (lane & 31 of every tile) of the step staged next,
    // fetched a step ahead of the staging that uses them; -1 past the end of the list
    int rid[STEP_TILES];
    auto fetch_ids = [&](int st) {
        if (!DBL) return;
#pragma unroll
        for (int tt = 0; tt < STEP_TILES; ++tt) {
            const int pos = (t0 + st * STEP_TILES + tt) * 32 + lane5;
            rid[tt] = pos < ndb ? dblist[pos] : -1;
        }
    };
    auto rid_of = [&](int tt) {
        int r = rid[0];
#pragma unroll
        for (int k = 1; k < STEP_TILES; ++k) r = tt == k ? rid[k] : r;
        return r;
    };

    auto stage = [&](auto BUF, int st) {
        constexpr int buf = decltype(BUF)::value;
        const int tile = t0 + st * STEP_TILES;
        char* dst = smem + buf * kCoarseStepBytes;
        if (DBL) {
#pragma unroll
            for (int p = 0; p < kCoarseStepBytes / 1024 / kCoarseWaves; ++p) {
                const int piece = wave + p * kCoarseWaves;   // 1 KiB = (tile of the step, k-step): 64 lanes = (half, row)
                const int row = max(rid_of(piece / S), 0);   // padding positions stage some real row; their start value is -inf
                const char* src = reinterpret_cast<const char*>(bfB) + ((size_t)(row >> 5) * S + (piece % S)) * 1024 +
                                  (half * 32 + (row & 31)) * 16;
                __builtin_amdgcn_global_load_lds((const VSM_GLB void*)src, (VSM_LDS void*)(dst + piece * 1024), 16, 0, 0);
            }
            if (wave * 64 < STEP_ROWS) {
                const int r = rid_of(min(wave * 2 + half, STEP_TILES - 1));
                __builtin_amdgcn_global_load_lds((const VSM_GLB void*)(initB + (r >= 0 ? r : pr.nB_pad)),
                                                 (VSM_LDS void*)((buf ? s_init1 : s_init0) + wave * 64), 4, 0, 0);
            }
            return;
        }
        const char* src = reinterpret_cast<const char*>(bfB) + (size_t)tile * TILE_BYTES + lane * 16;
#pragma unroll
        for (int p = 0; p < kCoarseStepBytes / 1024 / kCoarseWaves; ++p) {
            const int piece = wave + p * kCoarseWaves;
            __builtin_amdgcn_global_load_lds((const VSM_GLB void*)(src + piece * 1024),
                                             (VSM_LDS void*)(dst + piece * 1024), 16, 0, 0);
        }
        if (wave * 64 < STEP_ROWS)
            __builtin_amdgcn_global_load_lds((const VSM_GLB void*)(initB + tile * 32 + wave * 64 + lane),
                                             (VSM_LDS void*)((buf ? s_init1 : s_init0) + wave * 64), 4, 0, 0);
    };

    auto compute = [&](auto BUF, int st) {
        constexpr int buf = decltype(BUF)::value;
        const float* s_init = buf ? s_init1 : s_init0;
#pragma unroll 2
        for (int tt = 0; tt < STEP_TILES; ++tt) {
            const int tile = t0 + st * STEP_TILES + tt;
            const v8bf* At = reinterpret_cast<const v8bf*>(smem + buf * kCoarseStepBytes + tt * TILE_BYTES) + lane;
            v16f acc[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = *reinterpret_cast<const float4*>(s_init + tt * 32 + 8 * g + 4 * half);
                    acc[rb][4 * g] = v.x; acc[rb][4 * g + 1] = v.y; acc[rb][4 * g + 2] = v.z; acc[rb][4 * g + 3] = v.w;
                }
            }
            // Operand reads run one group of G k-steps ahead of the MFMAs that use them, pinned in program order.
            // Left to itself the compiler reads a pair of k-steps, waits for everything outstanding and then issues
            // their MFMAs, so every wave idles for one LDS round trip per four MFMAs (6.27 -> 5.83 ms on the
            // 16384 x 1M query with the reads ahead; groups of four k-steps ran 1.5 % slower than pairs).
            constexpr int G = 2;
            v8bf cur[G], nxt[G];
#pragma unroll
            for (int g = 0; g < G; ++g) cur[g] = At[g * 64];
#pragma unroll
            for (int s = 0; s < S; s += G) {
                // the wait for this group's operands lands on these empty statements, i.e. before the reads of the
                // next group are issued (placed at the MFMAs it would also wait for the reads just issued)
#pragma unroll
                for (int g = 0; g < G; ++g) asm volatile("" : "+v"(cur[g]));
#pragma unroll
                for (int g = 0; g < G; ++g) nxt[g] = s + G < S ? At[(s + G + g) * 64] : cur[g];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
                        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[g], qf[rb][s + g], acc[rb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < G; ++g) cur[g] = nxt[g];
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float tmax = -INFINITY;
                if (TOP1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) tmax = fmaxf(fmaxf(tmax, acc[rb][2 * e]), acc[rb][2 * e + 1]);
                    b1[rb] = fmaxf(b1[rb], tmax);
                    thr[rb] = fmaxf(b1[rb] - margin[rb], floorc[rb]);   // padded queries: NaN or +inf against +inf
                } else {
                    // per lane and tile only the maximum of the 16 scores; the lane's runner-up is then the second
                    // largest of its tile maxima: a lower bound of its true runner-up (they differ only when a lane's
                    // two best rows sit in the same 16 rows of one tile), and a lower threshold only adds survivors
#pragma unroll
                    for (int e = 0; e < 8; ++e) tmax = fmaxf(fmaxf(tmax, acc[rb][2 * e]), acc[rb][2 * e + 1]);
                    b2[rb] = __builtin_amdgcn_fmed3f(b1[rb], b2[rb], tmax);
                    b1[rb] = fmaxf(b1[rb], tmax);
                    thr[rb] = fmaxf(b2[rb] - margin[rb], floorc[rb]);   // padded queries: NaN or +inf against +inf
                }
                if (tmax >= thr[rb]) {
                    // some of this lane's 16 rows reach the threshold as it stands after this tile (no row
                    // below it can be among the final survivors): one list entry for the tile, with a bit
                    // per such row and the largest score as its key.  Padding rows (-inf) get in only while
                    // thr is -inf; the re-rank drops them.
                    unsigned rowbits = 0;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) rowbits |= (acc[rb][reg] >= thr[rb] ? 1u : 0u) << reg;
                    const size_t mysub = ((size_t)split * pr.nA_pad + qidx0 + rb * 32) * 2 + half;
                    if (cnt[rb] == kCoarseCap) {
                        // full: the threshold has risen since the early entries went in, and an entry whose
                        // key is below the present threshold can never hold a survivor; drop those
                        int kept = 0;
                        for (int e = 0; e < kCoarseCap; ++e) {
                            const uint2 en = lists[list_slot(mysub, e)];
                            if (__uint_as_float(en.y) >= thr[rb]) lists[list_slot(mysub, kept++)] = en;
                        }
                        cnt[rb] = kept;
                    }
                    if (cnt[rb] < kCoarseCap)
                        lists[list_slot(mysub, cnt[rb])] =
                            make_uint2(((unsigned)(tile - t0) << 16) | rowbits, __float_as_uint(tmax));
                    ++cnt[rb];
                }
            }
            const int rel = tile - t0;
            // (not after the split's last tile: the other splits of a search run beside this one and end with it)
            if (shared_q && ((rel & 31) == 31 || rel == 3 || rel == 7 || rel == 15) && tile + 1 < t1e) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    if (!live[rb]) continue;
                    // read what the other splits have reached (past this CU's cache), publish only an improvement:
                    // once the thresholds have settled the exchange is a load
                    const float mine = TOP1 ? b1[rb] : b2[rb];
                    const unsigned mybits = float_order_bits(mine);
                    const unsigned seen = __hip_atomic_load(shared_q + rb * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (mybits > seen) atomicMax(shared_q + rb * 32, mybits);
                    floorc[rb] = fmaxf(floorc[rb], fmaxf(float_of_order_bits(seen), mine) - margin[rb]);
                }
            }
        }
    };

    using Buf0 = std::integral_constant<int, 0>;
    using Buf1 = std::integral_constant<int, 1>;
    fetch_ids(0);
    if (nsteps > 0) stage(Buf0{}, 0);
    fetch_ids(1);
    __syncthreads();
    for (int st = 0; st < nsteps; st += 2) {
        if (st + 1 < nsteps) stage(Buf1{}, st + 1);
        fetch_ids(st + 2);
        compute(Buf0{}, st);
        __syncthreads();
        if (st + 1 < nsteps) {
            if (st + 2 < nsteps) stage(Buf0{}, st + 2);
            fetch_ids(st + 3);
            compute(Buf1{}, st + 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        if (cnt[rb] > kCoarseCap) atomicOr(overflow, 1);
        CoarsePart cp;
        cp.t1 = b1[rb]; cp.t2 = b2[rb]; cp.count = min(cnt[rb], kCoarseCap); cp.pad = 0;
        part[((size_t)split * pr.nA_pad + qidx0 + rb * 32) * 2 + half] = cp;
    }
}

template <int DIMP, bool TOP1, bool DBL>
__global__ __launch_bounds__(kCoarseWaves * 64, 2)
void bf16_coarse_kernel(const CoarseProblem* __restrict__ probs, int nprobs,
                        CoarsePart* __restrict__ part_all, uint2* __restrict__ lists_all, int* __restrict__ overflow,
                        unsigned* __restrict__ shared_thr, int64_t out_rows0)
{
    const int L = coarse_item_of_block();
    const int pi = find_problem(nprobs, L, [&](int i) { return probs[i].blk0; });
    coarse_body<DIMP, TOP1, DBL>(probs[pi], L, part_all, lists_all, overflow, shared_thr, out_rows0);
}

// The same for a one-pair call: the searches (and the pair's other device records) arrive as kernel arguments, so
// nothing has to be copied or written before the first workgroup can start.  The first item also writes the records
// into the submission's block in device memory, where the kernels behind this one read them.
template <int DIMP, bool TOP1>
__global__ __launch_bounds__(kCoarseWaves * 64, 2)
void bf16_coarse_args_kernel(const F32PlanArgs a, char* __restrict__ block,
                             CoarsePart* __restrict__ part_all, uint2* __restrict__ lists_all, int* __restrict__ overflow,
                             unsigned* __restrict__ shared_thr, int64_t out_rows0)
{
    const int L = coarse_item_of_block();
    if (L == 0) {
        constexpr int kProbWords = sizeof(CoarseProblem) / 4, kPairWords = sizeof(SiftPair) / 4, kRefWords = sizeof(F32PairRef) / 4;
        const int tid = threadIdx.x;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&a.probs[0]);
        uint32_t* dst = reinterpret_cast<uint32_t*>(block + 256);
        for (int i = tid; i < a.nprobs * kProbWords; i += kCoarseWaves * 64) dst[i] = src[i];
        const uint32_t* pair_src = reinterpret_cast<const uint32_t*>(&a.pair);
        uint32_t* pair = reinterpret_cast<uint32_t*>(block + 256 + a.pbytes);
        if (tid < kPairWords) pair[tid] = pair_src[tid];
        const uint32_t* ref_src = reinterpret_cast<const uint32_t*>(&a.ref);
        uint32_t* ref = reinterpret_cast<uint32_t*>(block + 256 + a.pbytes + a.pair_bytes);
        if (tid < kRefWords) ref[tid] = ref_src[tid];
    }
    const CoarseProblem& P = a.nprobs > 1 && L >= a.probs[1].blk0 ? a.probs[1] : a.probs[0];
    coarse_body<DIMP, TOP1, false>(P, L, part_all, lists_all, overflow, shared_thr, out_rows0);
}

// ---------------------------------------------------------------------------------
// exact re-rank of the survivors
// ---------------------------------------------------------------------------------
// Half a wave per query.  Lane l of the half first folds the coarse top-2 of sublist l into the query's final coarse
// runner-up T2, then the half walks the concatenated survivor lists of the sublists that can hold a survivor
// 32 entries at a time, keeps the entries with coarse score >= T2 - margin, and scores those exactly with the
// k-ascending fmaf chain on the plain row-major f32 copies of the two rows.
constexpr int kRerankWaves = 4;
// workgroups per CU the per-query re-rank is compiled for: 20 waves per CU (96 registers); with 24 and 32 the
// spills cost more than the residency gives (16.3, 16.9, 21.5, 22.9 us for 5, 6, 7, 8 on a 4096 x 4096 pair)
constexpr int kRerankMinBlocks = 5;
constexpr int kRerankPerWave = 2;   // queries per wave: one per half

// A list entry names up to 16 rows of one database tile: x = (tile - first tile of the split) << 16 |
// bit per accumulator register, y = the largest coarse score among them.  Register `reg` of lane half
// `half` holds row (reg & 3) + 8 * (reg >> 2) + 4 * half of the tile.
__device__ __forceinline__ int entry_row(unsigned x, int reg, int half, int first_tile)
{
    return (first_tile + (int)(x >> 16)) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

struct Best2 {
    float s1, s2;
    int r1, r2;
};

__device__ __forceinline__ bool ranks_before(float x, int jx, float y, int jy)
{
    return x > y || (x == y && (unsigned)jx < (unsigned)jy);
}

__device__ __forceinline__ void best2_insert(Best2& b, float sc, int row)
{
    if (ranks_before(sc, row, b.s1, b.r1)) {
        b.s2 = b.s1; b.r2 = b.r1;
        b.s1 = sc; b.r1 = row;
    } else if (ranks_before(sc, row, b.s2, b.r2)) {
        b.s2 = sc; b.r2 = row;
    }
}

// Exact scores of up to four surviving rows at a time: sixteen lanes per row, lane `sub` of a group
// holds columns [sub * DIMP/16, (sub + 1) * DIMP/16) of the row (one round trip to memory for the whole
// row), and the k-ascending fmaf chain runs through the group's lanes in order.  The query row sits in LDS.
// Few registers on purpose: a wave lives 7 us, most of it waiting for one of four trips to memory, so what a
// 4096 x 4096 pair (8192 waves over both directions) costs is set by how many of them are resident at once.
template <int DIMP>
__device__ __forceinline__ void rerank_body(const CoarseProblem& P,
                                            const CoarsePart* __restrict__ part_all, const uint2* __restrict__ lists_all,
                                            int32_t* __restrict__ j1_all, int32_t* __restrict__ j2_all,
                                            float* __restrict__ k1_all, float* __restrict__ k2_all)
{
    constexpr int Q = DIMP / 64;                        // float4 pieces of a row per lane of a sixteen-lane group
    constexpr int AP = DIMP > 128 ? DIMP / 128 : 1;     // float4 pieces of the query row a lane fetches
    __shared__ int s_rows[kRerankWaves][2][32];
    __shared__ __attribute__((aligned(16))) float s_a[kRerankWaves][2][DIMP];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5, l5 = lane & 31;         // a query per half wave
    const F32Problem pr = P.pr;
    const int nsplits = P.nsplits;
    const CoarsePart* __restrict__ part = part_all + P.part_off;
    const uint2* __restrict__ lists = lists_all + P.part_off * kCoarseCap;
    const int i0 = ((int)blockIdx.x * kRerankWaves + wave) * kRerankPerWave;
    const int i = i0 + half;                            // query slot
    const int32_t* __restrict__ qlist = P.qlist;
    const int nq = qlist ? *P.qcount : pr.nA;
    if (i0 >= nq) return;   // per wave: nothing below waits for another wave
    const bool live = i < nq;
    const int qrow = live ? (qlist ? qlist[i] : i) : 0;
    // the query row goes to LDS once the coarse records have been asked for: one trip to memory for both
    float4 apiece[AP];
#pragma unroll
    for (int k = 0; k < AP; ++k) {
        const int piece = l5 + 32 * k;
        apiece[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (live && piece * 4 < DIMP) apiece[k] = *reinterpret_cast<const float4*>(P.rowsA + (size_t)qrow * DIMP + piece * 4);
    }

    volatile int* rows = s_rows[wave][half];
    const bool l2 = pr.metric == 1;
    const float na = pr.nrmA[qrow];
    const float margin = coarse_margin(l2, na, *P.maxnormB_sq);
    const float floorc = l2 ? -INFINITY : P.key_floor - 0.5f * margin;
    const int nsub = nsplits * 2;
    auto sub_index = [&](int sid) { return ((size_t)(sid >> 1) * pr.nA_pad + i) * 2 + (sid & 1); };
    // with a database list the entries name positions in it
    const int32_t* __restrict__ dblist = P.dblist;
    const int nrows_b = dblist ? *P.dbcount : pr.nB;

    // final coarse runner-up over all splits and both lane halves of the coarse kernel
    float T1 = -INFINITY, T2 = -INFINITY;
    int cnt_first = 0;             // list length of sublist `l5` (first chunk of 32 sublists)
    float top_first = -INFINITY;   // and the largest key in it
    for (int sid = l5; sid < nsub; sid += 32) {
        if (!live) continue;
        const CoarsePart cp = part[sub_index(sid)];
        T2 = fmaxf(fmaxf(T2, cp.t2), fminf(T1, cp.t1));
        T1 = fmaxf(T1, cp.t1);
        if (sid == l5) {
            cnt_first = cp.count;
            top_first = cp.t1;
        }
    }
#pragma unroll
    for (int k = 0; k < AP; ++k) {
        const int piece = l5 + 32 * k;
        if (piece * 4 < DIMP) *reinterpret_cast<float4*>(&s_a[wave][half][piece * 4]) = apiece[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) {
        const float o1 = __shfl_xor(T1, m), o2 = __shfl_xor(T2, m);
        T2 = fmaxf(fmaxf(T2, o2), fminf(T1, o1));
        T1 = fmaxf(T1, o1);
    }
    const float thr = fmaxf((P.top1 ? T1 : T2) - margin, floorc);

    Best2 best;
    best.s1 = -INFINITY; best.s2 = -INFINITY; best.r1 = -1; best.r2 = -1;
    int npending = 0;   // rows of this half's query waiting in `rows` (the same in every lane of the half)
    // Sixteen lanes per surviving row, two rows of either query at a time: lane `sub` of a group holds columns
    // [sub * DIMP/16, (sub + 1) * DIMP/16) of the row and of the query, and the k-ascending chain runs through the
    // group's lanes in order, handed on by a move within the row of sixteen lanes.
    const int sub = lane & 15, gsel = (lane >> 4) & 1;
    const float4* a4 = reinterpret_cast<const float4*>(&s_a[wave][half][0]) + sub * Q;
    auto flush = [&]() {
        for (int base = 0; __any(base < npending); base += 2) {
            const bool act = base + gsel < npending;
            const int row = act ? rows[base + gsel] : 0;
            const float4* pb = reinterpret_cast<const float4*>(P.rowsB + (size_t)row * DIMP) + sub * Q;
            float4 b[Q];
#pragma unroll
            for (int u = 0; u < Q; ++u) b[u] = pb[u];
            const float nbr = pr.nrmB[row];
            float d = 0.0f;
#pragma unroll
            for (int stage = 0; stage < 16; ++stage) {
                if (sub == stage) {
#pragma unroll
                    for (int u = 0; u < Q; ++u) {
                        const float4 a = a4[u];
                        d = fmaf(b[u].x, a.x, d);
                        d = fmaf(b[u].y, a.y, d);
                        d = fmaf(b[u].z, a.z, d);
                        d = fmaf(b[u].w, a.w, d);
                    }
                }
                if (stage < 15) {
                    // the value of the lane below (row_shr:1)
                    const float handed = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(d), __float_as_int(d), 0x111, 0xf, 0xf, false));
                    if (sub == stage + 1) d = handed;
                }
            }
            const float sc = l2 ? fmaf(2.0f, d, -nbr) : d;
            if (act && sub == 15 && sc > -INFINITY) best2_insert(best, sc, row);   // the finished chain
        }
        npending = 0;
    };

    for (int chunk = 0; chunk < nsub; chunk += 32) {
        const int sid = chunk + l5;
        // a sublist's largest key is its t1: below the final threshold none of its entries survives
        int cnt = 0;
        if (chunk == 0) {
            cnt = top_first >= thr ? cnt_first : 0;
        } else if (live && sid < nsub) {
            const CoarsePart cp = part[sub_index(sid)];
            cnt = cp.t1 >= thr ? cp.count : 0;
        }
        int incl = cnt;   // inclusive prefix sum of the chunk's list lengths, per half
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const int o = __shfl_up(incl, d, 32);
            if (l5 >= d) incl += o;
        }
        const int total = __shfl(incl, 31, 32);
        for (int e0 = 0; __any(e0 < total); e0 += 32) {
            const int e = e0 + l5;
            // sublist of flat entry e: the first lane of the half whose inclusive prefix exceeds e
            int lo = 0;
#pragma unroll
            for (int step = 16; step >= 1; step >>= 1) {
                const int probe = __shfl(incl, lo + step - 1, 32);
                if (probe <= e) lo += step;
            }
            lo = min(lo, 31);
            const int within = e - (__shfl(incl, lo, 32) - __shfl(cnt, lo, 32));
            unsigned ex = 0;   // the entry's rows, if it survives the final threshold
            if (e < total) {
                const uint2 en = lists[list_slot(sub_index(chunk + lo), within)];
                if (__uint_as_float(en.y) >= thr) ex = en.x;
            }
            if (!__ballot((ex & 0xffffu) != 0)) continue;
            const int esid = chunk + lo;
            const int first_tile = (esid >> 1) * P.tiles_per_split;
            // round j queues the j-th named row of every surviving entry (one round, as a rule)
            unsigned bits = ex & 0xffffu;
            while (__ballot(bits != 0)) {
                const int reg = bits ? __ffs(bits) - 1 : 0;
                int row = entry_row(ex, reg, esid & 1, first_tile);
                const bool keep = bits != 0 && row < nrows_b;
                if (dblist && keep) row = dblist[row];
                bits &= bits - 1;
                const unsigned long long mask = __ballot(keep);
                if (!mask) continue;
                const unsigned mine = (unsigned)(mask >> (32 * half));   // this half's lanes
                const int n = __popc(mine);
                if (__any(npending + n > 32)) flush();
                if (keep) rows[npending + __popc(mine & ((1u << l5) - 1u))] = row;
                npending += n;
            }
        }
    }
    flush();

    // the two groups of a half hold disjoint rows: merge by (score descending, row ascending)
    {
        Best2 o;
        o.s1 = __shfl_xor(best.s1, 16); o.s2 = __shfl_xor(best.s2, 16);
        o.r1 = __shfl_xor(best.r1, 16); o.r2 = __shfl_xor(best.r2, 16);
        if (o.r1 >= 0) best2_insert(best, o.s1, o.r1);
        if (o.r2 >= 0) best2_insert(best, o.s2, o.r2);
    }
    if (live && l5 == 15) {
        const int64_t o = P.out_off + i;
        j1_all[o] = best.r1;
        j2_all[o] = best.r2;
        k1_all[o] = best.r1 < 0 ? (l2 ? INFINITY : -INFINITY) : (l2 ? fmaxf(-best.s1 + na, 0.0f) : best.s1);
        k2_all[o] = best.r2 < 0 ? (l2 ? INFINITY : -INFINITY) : (l2 ? fmaxf(-best.s2 + na, 0.0f) : best.s2);
    }
}

template <int DIMP>
__global__ __launch_bounds__(kRerankWaves * 64, kRerankMinBlocks)
void bf16_rerank_kernel(const CoarseProblem* __restrict__ probs, int nprobs,
                        const CoarsePart* __restrict__ part_all, const uint2* __restrict__ lists_all,
                        int32_t* __restrict__ j1_all, int32_t* __restrict__ j2_all,
                        float* __restrict__ k1_all, float* __restrict__ k2_all)
{
    // grid: x = block of kRerankWaves queries, y = problem (no search: a re-rank wave is short)
    rerank_body<DIMP>(probs[blockIdx.y], part_all, lists_all, j1_all, j2_all, k1_all, k2_all);
}

// the searches as kernel arguments (one-pair calls, like bf16_coarse_args_kernel)
template <int DIMP>
__global__ __launch_bounds__(kRerankWaves * 64, kRerankMinBlocks)
void bf16_rerank_args_kernel(const F32PlanArgs a,
                             const CoarsePart* __restrict__ part_all, const uint2* __restrict__ lists_all,
                             int32_t* __restrict__ j1_all, int32_t* __restrict__ j2_all,
                             float* __restrict__ k1_all, float* __restrict__ k2_all)
{
    const CoarseProblem& P = a.probs[blockIdx.y];
    rerank_body<DIMP>(P, part_all, lists_all, j1_all, j2_all, k1_all, k2_all);
}

// Results of a group's query, written by its first lane; `best` is the same in every lane of the group and lane
// `sub` holds columns [sub * DIMP / 8, (sub + 1) * DIMP / 8) of the query row in areg.
template <int Q>
__device__ __forceinline__ void group_write_results(const CoarseProblem& P, const F32Problem& pr, int i, bool live,
                                                    bool l2, int sub, int gbase, const float4 (&areg)[Q], const Best2& best,
                                                    int32_t* __restrict__ j1_all, int32_t* __restrict__ j2_all,
                                                    float* __restrict__ k1_all, float* __restrict__ k2_all)
{
    constexpr int DIMP = Q * 32;
    if (l2) {
        // L2 results in difference form, as f32_l2_finish_kernel states them: the two rows found by the key are
        // ordered by the k-ascending fmaf chain of (a[k] - b[k])^2 (lower row first when equal) and those
        // distances are reported.  The query's columns are in registers and both rows were read a moment ago.
        const bool has1 = best.r1 >= 0, has2 = best.r2 >= 0;
        const float4* p1 = reinterpret_cast<const float4*>(P.rowsB + (size_t)(has1 ? best.r1 : 0) * DIMP) + sub * Q;
        const float4* p2 = reinterpret_cast<const float4*>(P.rowsB + (size_t)(has2 ? best.r2 : (has1 ? best.r1 : 0)) * DIMP) + sub * Q;
        float4 x[Q], y[Q];
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            x[u] = p1[u];
            y[u] = p2[u];
        }
        float d1 = 0.0f, d2 = 0.0f;
#pragma unroll
        for (int stage = 0; stage < 8; ++stage) {
            if (sub == stage) {
#pragma unroll
                for (int u = 0; u < Q; ++u) {
                    float t;
                    t = areg[u].x - x[u].x; d1 = fmaf(t, t, d1);
                    t = areg[u].y - x[u].y; d1 = fmaf(t, t, d1);
                    t = areg[u].z - x[u].z; d1 = fmaf(t, t, d1);
                    t = areg[u].w - x[u].w; d1 = fmaf(t, t, d1);
                    t = areg[u].x - y[u].x; d2 = fmaf(t, t, d2);
                    t = areg[u].y - y[u].y; d2 = fmaf(t, t, d2);
                    t = areg[u].z - y[u].z; d2 = fmaf(t, t, d2);
                    t = areg[u].w - y[u].w; d2 = fmaf(t, t, d2);
                }
            }
            const float h1 = __shfl(d1, gbase + stage), h2 = __shfl(d2, gbase + stage);
            if (sub == stage + 1 || stage == 7) {
                d1 = h1;
                d2 = h2;
            }
        }
        if (live && sub == 0) {
            const int64_t o = P.out_off + i;
            int a1 = best.r1, a2 = best.r2;
            if (has2 && (d2 < d1 || (d2 == d1 && a2 < a1))) {
                const float t = d1; d1 = d2; d2 = t;
                const int u = a1; a1 = a2; a2 = u;
            }
            j1_all[o] = a1;
            j2_all[o] = a2;
            k1_all[o] = has1 ? d1 : INFINITY;
            k2_all[o] = has2 ? d2 : INFINITY;
        }
        return;
    }
    if (live && sub == 0) {
        const int64_t o = P.out_off + i;
        j1_all[o] = best.r1;
        j2_all[o] = best.r2;
        k1_all[o] = best.r1 < 0 ? -INFINITY : best.s1;
        k2_all[o] = best.r2 < 0 ? -INFINITY : best.s2;
    }
}

// The same re-rank for submissions of many small problems (few sublists per query, millions of
// queries): eight queries per wave.  A group of eight lanes reads the coarse results and the list
// entries of one query and queues the rows that survive the final threshold; every group then scores the
// rows of its own queue, one row per round: lane `sub` holds the same DIMP / 8 columns of the row as of the
// query (registers; the whole row in one round trip to memory) and the k-ascending chain runs through the
// group's lanes in order, handed on by a shuffle.  A query's row is only fetched when one of its lists holds
// an entry (thresholded searches: most queries have none).  Results are bit-identical to bf16_rerank_kernel.
constexpr int kGroupWaves = 4;
constexpr int kGroupQueries = kGroupWaves * 8;   // queries per workgroup
constexpr int kGroupQueue = 32;                  // surviving rows a group collects before they are scored
constexpr int kGroupMaxSub = 64;                 // sublists per query it can serve (a bit each in a 64-bit mask)
static_assert(kCoarseCap < 65536, "list lengths travel as 16-bit words");

template <int DIMP>
__global__ __launch_bounds__(kGroupWaves * 64)
void bf16_rerank_group_kernel(const CoarseProblem* __restrict__ probs, int nprobs,
                              const CoarsePart* __restrict__ part_all, const uint2* __restrict__ lists_all,
                              int32_t* __restrict__ j1_all, int32_t* __restrict__ j2_all,
                              float* __restrict__ k1_all, float* __restrict__ k2_all)
{
    constexpr int Q = DIMP / 32;      // float4 pieces of a row per lane of a group
    __shared__ int s_queue[kGroupWaves][8][kGroupQueue];
    __shared__ unsigned short s_count[kGroupWaves][8][kGroupMaxSub];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane & 7, grp = lane >> 3, gbase = lane & ~7;
    const CoarseProblem& P = probs[blockIdx.y];
    const F32Problem pr = P.pr;
    const int i = ((int)blockIdx.x * kGroupWaves + wave) * 8 + grp;   // query slot
    const int32_t* __restrict__ qlist = P.qlist;
    const int nslots = qlist ? *P.qcount : pr.nA;
    if ((int)blockIdx.x * kGroupQueries >= nslots) return;   // the whole block is past the end of the query list
    const bool live = i < nslots;
    const int qrow = live ? (qlist ? qlist[i] : i) : 0;
    volatile int* queue = s_queue[wave][grp];
    const int nsub = P.nsplits * 2;
    const CoarsePart* __restrict__ part = part_all + P.part_off;
    const uint2* __restrict__ lists = lists_all + P.part_off * kCoarseCap;
    auto sub_index = [&](int sid) { return ((size_t)(sid >> 1) * pr.nA_pad + i) * 2 + (sid & 1); };
    // with a database list the entries name positions in it
    const int32_t* __restrict__ dblist = P.dblist;
    const int nrows_b = dblist ? *P.dbcount : pr.nB;

    const bool l2 = pr.metric == 1;
    const float na = live ? pr.nrmA[qrow] : 0.0f;
    const float margin = coarse_margin(l2, na, *P.maxnormB_sq);
    const float floorc = l2 ? -INFINITY : P.key_floor - 0.5f * margin;

    // final coarse runner-up over all splits and both lane halves.  Lane `sub` reads the records of sublists
    // sub, sub + 8, ... (all of them in flight at once) and keeps each one's largest key and list length.
    float T1 = -INFINITY, T2 = -INFINITY;
    float top[kGroupMaxSub / 8];
    int len[kGroupMaxSub / 8];
#pragma unroll
    for (int k = 0; k < kGroupMaxSub / 8; ++k) {
        const int sid = sub + 8 * k;
        top[k] = -INFINITY;
        len[k] = 0;
        if (live && sid < nsub) {
            const CoarsePart cp = part[sub_index(sid)];
            top[k] = cp.t1;
            len[k] = cp.count;
            T2 = fmaxf(fmaxf(T2, cp.t2), fminf(T1, cp.t1));
            T1 = fmaxf(T1, cp.t1);
        }
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
        const float o1 = __shfl_xor(T1, m), o2 = __shfl_xor(T2, m);
        T2 = fmaxf(fmaxf(T2, o2), fminf(T1, o1));
        T1 = fmaxf(T1, o1);
    }
    const float thr = fmaxf((P.top1 ? T1 : T2) - margin, floorc);
    // Only sublists whose largest key reaches the final threshold can hold a survivor (a search cut into many
    // splits: one or two of them): a bit per such sublist, the same in every lane of the group.
    unsigned long long todo = 0;
#pragma unroll
    for (int k = 0; k < kGroupMaxSub / 8; ++k) {
        if (len[k] > 0 && top[k] >= thr) {
            todo |= 1ull << (sub + 8 * k);
            ((volatile unsigned short*)s_count[wave][grp])[sub + 8 * k] = (unsigned short)len[k];
        }
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) todo |= __shfl_xor(todo, m);
    // the query row is only ever read while a surviving row is scored: a query whose lists are all empty
    // (a best score under the floor of a thresholded search) leaves it in memory
    float4 areg[Q];   // columns [sub * DIMP / 8, (sub + 1) * DIMP / 8) of the group's query row
#pragma unroll
    for (int u = 0; u < Q; ++u) areg[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (todo != 0) {
        const float4* pa = reinterpret_cast<const float4*>(P.rowsA + (size_t)qrow * DIMP) + sub * Q;
#pragma unroll
        for (int u = 0; u < Q; ++u) areg[u] = pa[u];
    }

    Best2 best;   // the same in every lane of the group
    best.s1 = -INFINITY; best.s2 = -INFINITY; best.r1 = -1; best.r2 = -1;
    int nq = 0;   // rows in this group's queue (the same in every lane of the group)

    // whole wave: every group scores the rows of its own queue, as many rounds as the longest queue has rows
    auto score_all = [&]() {
        int rounds = nq;
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) rounds = max(rounds, __shfl_xor(rounds, m));
        for (int q = 0; q < rounds; ++q) {
            const int row = q < nq ? queue[q] : -1;   // -1: a padding row of the last tile, or no row this round
            const bool act = row >= 0;
            const float4* pb = reinterpret_cast<const float4*>(P.rowsB + (size_t)(act ? row : 0) * DIMP) + sub * Q;
            float4 b[Q];
#pragma unroll
            for (int u = 0; u < Q; ++u) b[u] = pb[u];
            const float nbr = l2 ? pr.nrmB[act ? row : 0] : 0.0f;
            float d = 0.0f;
#pragma unroll
            for (int stage = 0; stage < 8; ++stage) {
                if (sub == stage) {
#pragma unroll
                    for (int u = 0; u < Q; ++u) {
                        d = fmaf(b[u].x, areg[u].x, d);
                        d = fmaf(b[u].y, areg[u].y, d);
                        d = fmaf(b[u].z, areg[u].z, d);
                        d = fmaf(b[u].w, areg[u].w, d);
                    }
                }
                const float handed = __shfl(d, gbase + stage);
                if (sub == stage + 1) d = handed;
                if (stage == 7) d = handed;   // the finished chain, in every lane of the group
            }
            const float sc = l2 ? fmaf(2.0f, d, -nbr) : d;
            if (act && sc > -INFINITY) best2_insert(best, sc, row);
        }
        nq = 0;
    };

    // every group walks its own sublists: the groups of a wave are at different ones in the same iteration
    while (__any(todo != 0)) {
        const bool has = todo != 0;
        const int sid = has ? __ffsll((long long)todo) - 1 : 0;
        todo &= todo - 1;
        const int cnt = has ? ((volatile unsigned short*)s_count[wave][grp])[sid] : 0;
        const size_t mysub = sub_index(sid);
        const int first_tile = (sid >> 1) * P.tiles_per_split;
        for (int e0 = 0; __any(e0 < cnt); e0 += 32) {
            unsigned ex[4];   // four entries per lane in flight
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + 8 * k + sub;
                ex[k] = 0;
                if (e < cnt) {
                    const uint2 en = lists[list_slot(mysub, e)];
                    if (__uint_as_float(en.y) >= thr) ex[k] = en.x;
                }
            }
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                unsigned m = (unsigned)(__ballot((ex[0] & 0xffffu) != 0) >> gbase) & 0xffu;   // this group's survivors
                while (__any(m != 0)) {
                    if (__any(nq + 16 > kGroupQueue)) score_all();
                    if (m) {
                        const int src = __ffs(m) - 1;
                        m &= m - 1;
                        const unsigned x = __shfl(ex[0], gbase + src);
                        const unsigned bits = x & 0xffffu;
                        // lane `sub` places the rows of registers sub and sub + 8
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int reg = sub + 8 * h;
                            if ((bits >> reg) & 1u) {
                                const int r = entry_row(x, reg, sid & 1, first_tile);
                                queue[nq + __popc(bits & ((1u << reg) - 1u))] = r < nrows_b ? (dblist ? dblist[r] : r) : -1;
                            }
                        }
                        nq += __popc(bits);
                    }
                }
                ex[0] = ex[1]; ex[1] = ex[2]; ex[2] = ex[3]; ex[3] = 0;
            }
        }
    }
    score_all();
    group_write_results<Q>(P, pr, i, live, l2, sub, gbase, areg, best, j1_all, j2_all, k1_all, k2_all);
}

// ---------------------------------------------------------------------------------
// planning and launch
// ---------------------------------------------------------------------------------
CoarseTotals plan_coarse(CoarseProblem* probs, int n, int target_wgs, int64_t out_rows0)
{
    long long unsplit = 0;
    for (int i = 0; i < n; ++i) unsplit += probs[i].pr.nA_pad / kCoarseStripRows;
    // database splits per problem: only as many as it takes to fill the chip
    const int want = unsplit >= target_wgs || unsplit == 0 ? 1 : (int)((target_wgs + unsplit - 1) / unsplit);
    CoarseTotals t;
    t.out_rows = out_rows0;
    t.out_rows0 = out_rows0;
    for (int i = 0; i < n; ++i) {
        CoarseProblem& p = probs[i];
        const int ntiles = p.pr.nB_pad / 32;
        p.nstrips = p.pr.nA_pad / kCoarseStripRows;
        int tps = (ntiles + want - 1) / want;
        tps = std::max(8, (tps + 7) & ~7);            // whole LDS steps at every dimension
        tps = std::min(tps, 65536);                   // a list entry has 16 bits for the tile within its split
        p.tiles_per_split = tps;
        p.nsplits = std::max(1, (ntiles + tps - 1) / tps);
        p.blk0 = t.coarse_blocks;
        p.part_off = t.sublists;
        p.out_off = t.out_rows;
        t.coarse_blocks += p.nstrips * p.nsplits;
        t.rerank_blocks = std::max(t.rerank_blocks, (p.pr.nA + kRerankWaves - 1) / kRerankWaves);
        t.max_nsplits = std::max(t.max_nsplits, p.nsplits);
        t.sublists += (int64_t)p.nsplits * p.pr.nA_pad * 2;
        t.out_rows += p.pr.nA_pad;
    }
    return t;
}

static size_t part_bytes(int64_t sublists) { return ((size_t)sublists * sizeof(CoarsePart) + 255) & ~(size_t)255; }

static size_t list_bytes(int64_t sublists) { return ((size_t)sublists * kCoarseCap * sizeof(uint2) + 255) & ~(size_t)255; }

// [parts][lists][one shared-threshold word per output row of the group]
size_t coarse_scratch_bytes(const CoarseTotals& t)
{
    return part_bytes(t.sublists) + list_bytes(t.sublists) + (size_t)(t.out_rows - t.out_rows0) * sizeof(unsigned) + 256;
}

template <int DIMP, bool TOP1, bool DBL>
static void launch_coarse_t(const CoarseProblem* d_probs, int nprobs, int blocks, CoarsePart* part, uint2* lists,
                            int* overflow, unsigned* shared_thr, int64_t out_rows0, hipStream_t s)
{
    const int lds = 2 * kCoarseStepBytes;
    static DeviceFlags attr_set;   // the attribute is per device (and per instantiation: one flag set each)
    ensure_dynamic_lds(attr_set, reinterpret_cast<const void*>(bf16_coarse_kernel<DIMP, TOP1, DBL>), lds);
    hipLaunchKernelGGL((bf16_coarse_kernel<DIMP, TOP1, DBL>), dim3(blocks), dim3(kCoarseWaves * 64), lds, s,
                       d_probs, nprobs, part, lists, overflow, shared_thr, out_rows0);
}

template <int DIMP>
static void launch_coarse_d(const CoarseProblem* d_probs, int nprobs, int blocks, CoarsePart* part, uint2* lists,
                            int* overflow, unsigned* shared_thr, int64_t out_rows0, hipStream_t s, bool top1, bool db_lists)
{
    if (db_lists) {
        if (top1) launch_coarse_t<DIMP, true, true>(d_probs, nprobs, blocks, part, lists, overflow, shared_thr, out_rows0, s);
        else launch_coarse_t<DIMP, false, true>(d_probs, nprobs, blocks, part, lists, overflow, shared_thr, out_rows0, s);
    } else {
        if (top1) launch_coarse_t<DIMP, true, false>(d_probs, nprobs, blocks, part, lists, overflow, shared_thr, out_rows0, s);
        else launch_coarse_t<DIMP, false, false>(d_probs, nprobs, blocks, part, lists, overflow, shared_thr, out_rows0, s);
    }
}

unsigned* coarse_shared_thr(void* scratch, const CoarseTotals& t, int* words)
{
    *words = t.max_nsplits > 1 ? (int)(t.out_rows - t.out_rows0) : 0;
    return reinterpret_cast<unsigned*>(static_cast<char*>(scratch) + part_bytes(t.sublists) + list_bytes(t.sublists));
}

template <int DIMP, bool TOP1>
static void launch_coarse_args_t(const F32PlanArgs& a, char* block, int blocks, CoarsePart* part, uint2* lists,
                                 int* overflow, unsigned* shared_thr, int64_t out_rows0, hipStream_t s)
{
    const int lds = 2 * kCoarseStepBytes;
    static DeviceFlags attr_set;
    ensure_dynamic_lds(attr_set, reinterpret_cast<const void*>(bf16_coarse_args_kernel<DIMP, TOP1>), lds);
    hipLaunchKernelGGL((bf16_coarse_args_kernel<DIMP, TOP1>), dim3(blocks), dim3(kCoarseWaves * 64), lds, s,
                       a, block, part, lists, overflow, shared_thr, out_rows0);
}

bool launch_bf16_batch(const CoarseProblem* d_probs, int nprobs, int dim, const CoarseTotals& t,
                       int32_t* j1, int32_t* j2, float* k1, float* k2, void* scratch, int* overflow,
                       hipStream_t s, bool top1, bool db_lists, bool thr_cleared, const F32PlanArgs* by_args,
                       char* args_block)
{
    if (nprobs <= 0 || t.coarse_blocks <= 0) return false;
    CoarsePart* part = static_cast<CoarsePart*>(scratch);
    uint2* lists = reinterpret_cast<uint2*>(static_cast<char*>(scratch) + part_bytes(t.sublists));
    unsigned* shared_thr = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lists) + list_bytes(t.sublists));
    // the words the splits of a search share: 0 = nothing published yet
    if (shared_thr && t.max_nsplits > 1 && !thr_cleared) (void)hipMemsetAsync(shared_thr, 0, (size_t)(t.out_rows - t.out_rows0) * sizeof(unsigned), s);
    const int dimp = f32_dimp(dim);
    if (by_args && !db_lists) {
        if (dimp == 64) top1 ? launch_coarse_args_t<64, true>(*by_args, args_block, t.coarse_blocks, part, lists, overflow, shared_thr, t.out_rows0, s)
                             : launch_coarse_args_t<64, false>(*by_args, args_block, t.coarse_blocks, part, lists, overflow, shared_thr, t.out_rows0, s);
        else if (dimp == 128) top1 ? launch_coarse_args_t<128, true>(*by_args, args_block, t.coarse_blocks, part, lists, overflow, shared_thr, t.out_rows0, s)
                                   : launch_coarse_args_t<128, false>(*by_args, args_block, t.coarse_blocks, part, lists, overflow, shared_thr, t.out_rows0, s);
        else top1 ? launch_coarse_args_t<256, true>(*by_args, args_block, t.coarse_blocks, part, lists, overflow, shared_thr, t.out_rows0, s)
                  : launch_coarse_args_t<256, false>(*by_args, args_block, t.coarse_blocks, part, lists, overflow, shared_thr, t.out_rows0, s);
    } else
    if (dimp == 64) launch_coarse_d<64>(d_probs, nprobs, t.coarse_blocks, part, lists, overflow, shared_thr, t.out_rows0, s, top1, db_lists);
    else if (dimp == 128) launch_coarse_d<128>(d_probs, nprobs, t.coarse_blocks, part, lists, overflow, shared_thr, t.out_rows0, s, top1, db_lists);
    else launch_coarse_d<256>(d_probs, nprobs, t.coarse_blocks, part, lists, overflow, shared_thr, t.out_rows0, s, top1, db_lists);
    if (t.rerank_blocks <= 0) return false;
    // many queries with few sublists each: eight queries per wave; few queries with many sublists
    // (a sliced database): one wave per query
    const bool grouped = t.max_nsplits * 2 <= 16 && (int64_t)t.rerank_blocks * nprobs >= 4096;
    if (grouped) {
        const int per = kGroupQueries / kRerankWaves;
        const dim3 grid((t.rerank_blocks + per - 1) / per, nprobs), block(kGroupWaves * 64);
        if (dimp == 64) hipLaunchKernelGGL(bf16_rerank_group_kernel<64>, grid, block, 0, s, d_probs, nprobs, part, lists, j1, j2, k1, k2);
        else if (dimp == 128) hipLaunchKernelGGL(bf16_rerank_group_kernel<128>, grid, block, 0, s, d_probs, nprobs, part, lists, j1, j2, k1, k2);
        else hipLaunchKernelGGL(bf16_rerank_group_kernel<256>, grid, block, 0, s, d_probs, nprobs, part, lists, j1, j2, k1, k2);
    } else if (by_args) {
        const dim3 grid((t.rerank_blocks + kRerankPerWave - 1) / kRerankPerWave, nprobs), block(kRerankWaves * 64);
        if (dimp == 64) hipLaunchKernelGGL(bf16_rerank_args_kernel<64>, grid, block, 0, s, *by_args, part, lists, j1, j2, k1, k2);
        else if (dimp == 128) hipLaunchKernelGGL(bf16_rerank_args_kernel<128>, grid, block, 0, s, *by_args, part, lists, j1, j2, k1, k2);
        else hipLaunchKernelGGL(bf16_rerank_args_kernel<256>, grid, block, 0, s, *by_args, part, lists, j1, j2, k1, k2);
    } else {
        const dim3 grid((t.rerank_blocks + kRerankPerWave - 1) / kRerankPerWave, nprobs), block(kRerankWaves * 64);
        if (dimp == 64) hipLaunchKernelGGL(bf16_rerank_kernel<64>, grid, block, 0, s, d_probs, nprobs, part, lists, j1, j2, k1, k2);
        else if (dimp == 128) hipLaunchKernelGGL(bf16_rerank_kernel<128>, grid, block, 0, s, d_probs, nprobs, part, lists, j1, j2, k1, k2);
        else hipLaunchKernelGGL(bf16_rerank_kernel<256>, grid, block, 0, s, d_probs, nprobs, part, lists, j1, j2, k1, k2);
    }
    return grouped;   // the grouped kernel leaves L2 results in difference form already
}

}  // namespace vsm
