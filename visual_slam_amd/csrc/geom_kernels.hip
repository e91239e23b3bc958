// Two-view geometric verification of putative matches: a fundamental matrix per image pair by
// RANSAC over the normalised eight-point solve, Sampson-distance inliers.  This is the step that
// follows matching in the workflow the reference feeds (TwoViewGeometry + inlier lists of a match
// record, MatchFile.h:68-161 and MatchFile.cpp:596-748; cv2.findEssentialMat + RANSAC in
// models/utils.py:288-312).  The reference holds no estimator of its own (it calls OpenCV), so the
// algorithm below is this library's, stated operation by operation so that the CPU checker can follow
// it bit for bit:
//
//   sample h of pair p : eight distinct matches drawn from a counter-based generator (splitmix64 of
//                        seed, the pair's key = key of the call's first pair + position, h, draw)
//   solve              : Hartley normalisation of the eight points (per image), the 8 x 9 system of
//                        x1^T F x0 = 0, Gaussian elimination with complete pivoting, free variable 1,
//                        back substitution, unit norm, de-normalisation, unit Frobenius norm (f64)
//   score              : F rounded to f32; a match is an inlier when e^2 < t^2 * (a0^2 + a1^2 + b0^2 + b1^2)
//                        with a = F x0, b = F^T x1, e = x1 . a (f32, every multiply-add one fused operation
//                        in the order written in sampson_inlier)
//   winner             : most inliers, lowest sample index among equals
//   finish             : rank 2 enforced on the winner (f64 Jacobi eigenvectors of F^T F, the direction
//                        of the smallest eigenvalue projected out), unit Frobenius norm, rounded to f32
//   refit (four times) : least-squares eight-point fit to ALL matches within 3, 2, 1 and 1 thresholds of
//                        the present matrix (a wide net first: the best sample of a noisy pair often
//                        fits half of the true matches only): Hartley normalisation over them, the
//                        9 x 9 normal matrix, its smallest
//                        eigenvector by five inverse iterations from (1, ..., 1) on an unpivoted LU,
//                        de-normalised, rank 2; sums over the inliers run over 64 interleaved partial
//                        sums (match m in partial sum m mod 64) joined by a fixed binary tree
//   result             : of the winner and the four refits the matrix with most inliers (the earliest
//                        of equals), its inliers listed in match order
//
// Every operation is an IEEE add, multiply, fused multiply-add (only where written as one), divide,
// square root or compare in a fixed order: results do not depend on the grid, and the file is compiled
// without floating-point contraction.
//
// Kernels: gather (match coordinates), solve (one sample per thread, its 8 x 9 system in registers), score (one sample per thread, all
// matches of the pair, workgroup-wide choice of the winner), emit (one wave per pair: rank 2, refits,
// final inlier list).
#include <hip/hip_runtime.h>

#include "vsm_internal.h"

namespace vsm {

namespace {

constexpr int kGeomThreads = 128;     // hypothesis kernel: threads (samples in flight) per image pair
constexpr int kGeomRefits = 4;

__host__ __device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// coordinates of the putative matches of all pairs, gathered once: (x0, y0, x1, y1) per match
__global__ void geom_gather_kernel(const GeomPair* __restrict__ pairs, int npairs,
                                   const int32_t* __restrict__ idx0, const int32_t* __restrict__ idx1,
                                   float4* __restrict__ coords)
{
    const GeomPair gp = pairs[blockIdx.y];
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < gp.n; m += gridDim.x * blockDim.x) {
        const int i = idx0[gp.off + m], j = idx1[gp.off + m];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i >= 0 && i < gp.n0 && j >= 0 && j < gp.n1) {
            const float2 a = reinterpret_cast<const float2*>(gp.xy0)[i];
            const float2 b = reinterpret_cast<const float2*>(gp.xy1)[j];
            v = make_float4(a.x, a.y, b.x, b.y);
        } else {
            v.x = v.y = v.z = v.w = __builtin_nanf("");   // never an inlier, never a usable sample
        }
        coords[gp.off + m] = v;
    }
}

__device__ __forceinline__ bool sampson_inlier(const float (&F)[9], const float4 p, float t2)
{
    const float a0 = __builtin_fmaf(F[0], p.x, __builtin_fmaf(F[1], p.y, F[2]));
    const float a1 = __builtin_fmaf(F[3], p.x, __builtin_fmaf(F[4], p.y, F[5]));
    const float a2 = __builtin_fmaf(F[6], p.x, __builtin_fmaf(F[7], p.y, F[8]));
    const float b0 = __builtin_fmaf(F[0], p.z, __builtin_fmaf(F[3], p.w, F[6]));
    const float b1 = __builtin_fmaf(F[1], p.z, __builtin_fmaf(F[4], p.w, F[7]));
    const float e = __builtin_fmaf(p.z, a0, __builtin_fmaf(p.w, a1, a2));
    const float den = __builtin_fmaf(b1, b1, __builtin_fmaf(b0, b0, __builtin_fmaf(a1, a1, a0 * a0)));
    return e * e < t2 * den;
}

// The eight-point solve of one sample.  The 8 x 9 system stays in registers (144 of them): every loop below has
// compile-time bounds and is unrolled, and the rows and columns the pivot search picks at run time are brought into
// place by conditional swaps (row r changes places with row k when r is the pivot's row, for every r > k), so no
// element is ever addressed by a run-time index.  With the system in LDS, addressed by run-time indices, the kernel
// was limited to one wave per SIMD by the 72 KB a workgroup needed and spent its time on LDS round trips.
__device__ bool solve_sample(const float4 (&pt)[8], float (&Fout)[9])
{
    double a[8][9];
    // Hartley normalisation per image
    double cx0 = 0, cy0 = 0, cx1 = 0, cy1 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        cx0 += (double)pt[k].x; cy0 += (double)pt[k].y;
        cx1 += (double)pt[k].z; cy1 += (double)pt[k].w;
    }
    cx0 /= 8.0; cy0 /= 8.0; cx1 /= 8.0; cy1 /= 8.0;
    double md0 = 0, md1 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double dx0 = (double)pt[k].x - cx0, dy0 = (double)pt[k].y - cy0;
        const double dx1 = (double)pt[k].z - cx1, dy1 = (double)pt[k].w - cy1;
        md0 += sqrt(dx0 * dx0 + dy0 * dy0);
        md1 += sqrt(dx1 * dx1 + dy1 * dy1);
    }
    md0 /= 8.0; md1 /= 8.0;
    if (!(md0 > 0.0) || !(md1 > 0.0)) return false;   // also false for NaN coordinates
    const double s0 = 1.4142135623730951 / md0, s1 = 1.4142135623730951 / md1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double x0 = ((double)pt[k].x - cx0) * s0, y0 = ((double)pt[k].y - cy0) * s0;
        const double x1 = ((double)pt[k].z - cx1) * s1, y1 = ((double)pt[k].w - cy1) * s1;
        a[k][0] = x1 * x0; a[k][1] = x1 * y0; a[k][2] = x1;
        a[k][3] = y1 * x0; a[k][4] = y1 * y0; a[k][5] = y1;
        a[k][6] = x0;      a[k][7] = y0;      a[k][8] = 1.0;
    }
    // elimination with complete pivoting; perm[j]: the unknown that column j of the working matrix stands for
    int perm[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) perm[j] = j;
    bool singular = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double best = 0.0;
        int pr = k, pc = k;
#pragma unroll
        for (int r = k; r < 8; ++r)
#pragma unroll
            for (int c = k; c < 9; ++c) {
                const double v = fabs(a[r][c]);
                const bool up = v > best;      // the first of equal magnitudes in this scan order stays
                best = up ? v : best;
                pr = up ? r : pr;
                pc = up ? c : pc;
            }
        singular = singular || !(best > 0.0);
#pragma unroll
        for (int r = k + 1; r < 8; ++r) {      // rows k and pr change places
            const bool sw = r == pr;
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const double x = a[k][c], y = a[r][c];
                a[k][c] = sw ? y : x;
                a[r][c] = sw ? x : y;
            }
        }
#pragma unroll
        for (int c = k + 1; c < 9; ++c) {      // columns k and pc change places, and the unknowns they stand for
            const bool sw = c == pc;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const double x = a[r][k], y = a[r][c];
                a[r][k] = sw ? y : x;
                a[r][c] = sw ? x : y;
            }
            const int pk = perm[k], pcv = perm[c];
            perm[k] = sw ? pcv : pk;
            perm[c] = sw ? pk : pcv;
        }
        const double piv = a[k][k];
#pragma unroll
        for (int r = k + 1; r < 8; ++r) {
            const double f = a[r][k] / piv;
#pragma unroll
            for (int c = k + 1; c < 9; ++c) a[r][c] -= f * a[k][c];
        }
    }
    if (singular) return false;   // (a zero pivot makes the values above NaN or infinite; they are not used)
    // back substitution with the free unknown (working column 8) set to 1
    double x[9];
    x[8] = 1.0;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        double s = 0.0;
#pragma unroll
        for (int c = k + 1; c < 9; ++c) s += a[k][c] * x[c];
        x[k] = -s / a[k][k];
    }
    // unknowns back in their places, unit norm
    double f[9], nn = 0.0;
#pragma unroll
    for (int u = 0; u < 9; ++u) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 9; ++j) v = perm[j] == u ? x[j] : v;
        f[u] = v;
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) nn += f[j] * f[j];
    nn = sqrt(nn);
    if (!(nn > 0.0)) return false;
#pragma unroll
    for (int j = 0; j < 9; ++j) f[j] /= nn;
    // F = T1^T Fn T0 with T = [s 0 -s cx; 0 s -s cy; 0 0 1]
    double M[9], G[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        M[3 * i + 0] = f[3 * i + 0] * s0;
        M[3 * i + 1] = f[3 * i + 1] * s0;
        M[3 * i + 2] = f[3 * i + 2] - (M[3 * i + 0] * cx0 + M[3 * i + 1] * cy0);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        G[j] = s1 * M[j];
        G[3 + j] = s1 * M[3 + j];
        G[6 + j] = M[6 + j] - (G[j] * cx1 + G[3 + j] * cy1);
    }
    double fn = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) fn += G[j] * G[j];
    fn = sqrt(fn);
    if (!(fn > 0.0) || !(fn < 1e300)) return false;
#pragma unroll
    for (int j = 0; j < 9; ++j) Fout[j] = (float)(G[j] / fn);
    return true;
}

// One sample per thread: grid (ceil(samples / 128), pairs).  An unusable sample (fewer than eight
// matches, repeated draws, degenerate points) leaves a NaN in the first element of its matrix.
__global__ __launch_bounds__(kGeomThreads)
void geom_solve_kernel(const GeomPair* __restrict__ pairs, const float4* __restrict__ coords,
                       GeomRule rule, float* __restrict__ Fh)
{
    const GeomPair gp = pairs[blockIdx.y];
    const int h = blockIdx.x * kGeomThreads + threadIdx.x;
    if (h >= rule.hypotheses) return;
    float* out = Fh + ((size_t)blockIdx.y * rule.hypotheses + h) * 9;
    const float4* __restrict__ pts = coords + gp.off;
    const int n = gp.n;
    float F[9];
    bool ok = n >= 8;
    if (ok) {
        const unsigned long long base = splitmix64(((unsigned long long)rule.seed << 32) ^ (rule.first_pair + (unsigned long long)blockIdx.y));
        int pick[8];
        int got = 0;
        for (int ctr = 0; ctr < 64 && got < 8; ++ctr) {
            const unsigned long long r = splitmix64(base + ((unsigned long long)h << 8) + (unsigned long long)ctr);
            const int idx = (int)(((r >> 32) * (unsigned long long)n) >> 32);
            bool dup = false;
#pragma unroll
            for (int k = 0; k < 8; ++k) dup = dup || (k < got && pick[k] == idx);
            if (!dup) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k == got) pick[k] = idx;
                ++got;
            }
        }
        ok = got == 8;
        if (ok) {
            float4 pt[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) pt[k] = pts[pick[k]];
            ok = solve_sample(pt, F);
        }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) out[j] = ok ? F[j] : __builtin_nanf("");
}

// one workgroup per image pair; thread t scores samples t, t + 256, ... against all matches
constexpr int kGeomScoreThreads = 256;
__global__ __launch_bounds__(kGeomScoreThreads)
void geom_score_kernel(const GeomPair* __restrict__ pairs, const float4* __restrict__ coords,
                       GeomRule rule, const float* __restrict__ Fh, float* __restrict__ Fbest,
                       int32_t* __restrict__ best_count)
{
    __shared__ unsigned long long s_key[kGeomScoreThreads];
    const GeomPair gp = pairs[blockIdx.x];
    const int t = threadIdx.x;
    const float4* __restrict__ pts = coords + gp.off;
    const int n = gp.n;
    const float t2 = rule.max_sampson * rule.max_sampson;
    int mycount = -1, myh = 0;
    for (int h = t; h < rule.hypotheses; h += kGeomScoreThreads) {
        const float* src = Fh + ((size_t)blockIdx.x * rule.hypotheses + h) * 9;
        float F[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) F[j] = src[j];
        if (F[0] != F[0]) continue;   // unusable sample
        int cnt = 0;
        for (int m = 0; m < n; ++m) cnt += sampson_inlier(F, pts[m], t2) ? 1 : 0;
        if (cnt > mycount) {   // samples come in ascending order: the first of equals stays
            mycount = cnt;
            myh = h;
        }
    }
    // most inliers, then lowest sample index
    const unsigned long long mine = mycount < 0 ? 0ull : (((unsigned long long)(unsigned)(mycount + 1)) << 32) | (unsigned)(0x7fffffff - myh);
    s_key[t] = mine;
    __syncthreads();
    for (int d = kGeomScoreThreads / 2; d > 0; d >>= 1) {
        if (t < d && s_key[t + d] > s_key[t]) s_key[t] = s_key[t + d];
        __syncthreads();
    }
    const unsigned long long win = s_key[0];
    if (win == 0ull) {
        if (t == 0) {
            best_count[blockIdx.x] = -1;
            for (int j = 0; j < 9; ++j) Fbest[(size_t)blockIdx.x * 9 + j] = 0.0f;
        }
        return;
    }
    if (mine == win) {   // exactly one thread: sample indices are distinct
        best_count[blockIdx.x] = mycount;
        const float* src = Fh + ((size_t)blockIdx.x * rule.hypotheses + myh) * 9;
        for (int j = 0; j < 9; ++j) Fbest[(size_t)blockIdx.x * 9 + j] = src[j];
    }
}

// rank 2: remove the right singular direction of the smallest singular value (f64 cyclic Jacobi on F^T F)
__device__ void enforce_rank2(const float* __restrict__ Fin, float (&Fout)[9])
{
    double F[9], S[9], V[9];
    for (int j = 0; j < 9; ++j) F[j] = (double)Fin[j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            S[3 * i + j] = (F[i] * F[j] + F[3 + i] * F[3 + j]) + F[6 + i] * F[6 + j];   // (F^T F)[i][j]
            V[3 * i + j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 6; ++sweep) {
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = S[3 * p + q];
            if (apq == 0.0) continue;
            const double theta = (S[3 * q + q] - S[3 * p + p]) / (2.0 * apq);
            const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
            // S <- J^T S J, V <- V J with J the rotation in the (p, q) plane
            for (int k = 0; k < 3; ++k) {
                const double skp = S[3 * k + p], skq = S[3 * k + q];
                S[3 * k + p] = c * skp - s * skq;
                S[3 * k + q] = s * skp + c * skq;
            }
            for (int k = 0; k < 3; ++k) {
                const double spk = S[3 * p + k], sqk = S[3 * q + k];
                S[3 * p + k] = c * spk - s * sqk;
                S[3 * q + k] = s * spk + c * sqk;
            }
            for (int k = 0; k < 3; ++k) {
                const double vkp = V[3 * k + p], vkq = V[3 * k + q];
                V[3 * k + p] = c * vkp - s * vkq;
                V[3 * k + q] = s * vkp + c * vkq;
            }
        }
    }
    int im = 0;
    if (S[4] < S[0]) im = 1;
    if (S[8] < S[3 * im + im]) im = 2;
    const double v0 = V[im], v1 = V[3 + im], v2 = V[6 + im];
    double G[9], fn = 0.0;
    for (int i = 0; i < 3; ++i) {
        const double fv = (F[3 * i] * v0 + F[3 * i + 1] * v1) + F[3 * i + 2] * v2;
        G[3 * i + 0] = F[3 * i + 0] - fv * v0;
        G[3 * i + 1] = F[3 * i + 1] - fv * v1;
        G[3 * i + 2] = F[3 * i + 2] - fv * v2;
    }
    for (int j = 0; j < 9; ++j) fn += G[j] * G[j];
    fn = sqrt(fn);
    for (int j = 0; j < 9; ++j) Fout[j] = fn > 0.0 ? (float)(G[j] / fn) : 0.0f;
}

// sum over the 64 lanes in a fixed order: lane t takes lane t + d for d = 32, 16, ..., 1; the total
// (in lane 0) goes to every lane
__device__ __forceinline__ double wave_tree_sum(double v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    return __shfl(v, 0);
}

__device__ __forceinline__ int wave_count_inliers(const float4* __restrict__ pts, int n, const float (&F)[9], float t2, int lane)
{
    int c = 0;
    for (int m = lane; m < n; m += 64) c += sampson_inlier(F, pts[m], t2) ? 1 : 0;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
    return c;
}

// Least-squares eight-point fit to the inliers of Fcur, by the whole wave (every lane ends with the same
// values).  False when fewer than eight inliers or a degenerate system.
__device__ bool refit_on_inliers(const float4* __restrict__ pts, int n, const float (&Fcur)[9], float t2, int lane,
                                 float (&Fnew)[9])
{
    double cnt = 0, sx0 = 0, sy0 = 0, sx1 = 0, sy1 = 0;
    for (int m = lane; m < n; m += 64) {
        const float4 p = pts[m];
        if (sampson_inlier(Fcur, p, t2)) {
            cnt += 1.0;
            sx0 += (double)p.x; sy0 += (double)p.y; sx1 += (double)p.z; sy1 += (double)p.w;
        }
    }
    cnt = wave_tree_sum(cnt);
    if (!(cnt >= 8.0)) return false;
    const double cx0 = wave_tree_sum(sx0) / cnt, cy0 = wave_tree_sum(sy0) / cnt;
    const double cx1 = wave_tree_sum(sx1) / cnt, cy1 = wave_tree_sum(sy1) / cnt;
    double md0 = 0, md1 = 0;
    for (int m = lane; m < n; m += 64) {
        const float4 p = pts[m];
        if (sampson_inlier(Fcur, p, t2)) {
            const double dx0 = (double)p.x - cx0, dy0 = (double)p.y - cy0;
            const double dx1 = (double)p.z - cx1, dy1 = (double)p.w - cy1;
            md0 += sqrt(dx0 * dx0 + dy0 * dy0);
            md1 += sqrt(dx1 * dx1 + dy1 * dy1);
        }
    }
    md0 = wave_tree_sum(md0) / cnt;
    md1 = wave_tree_sum(md1) / cnt;
    if (!(md0 > 0.0) || !(md1 > 0.0)) return false;
    const double s0 = 1.4142135623730951 / md0, s1 = 1.4142135623730951 / md1;
    // upper triangle of the normal matrix, row by row
    double N[45];
#pragma unroll
    for (int q = 0; q < 45; ++q) N[q] = 0.0;
    for (int m = lane; m < n; m += 64) {
        const float4 p = pts[m];
        if (sampson_inlier(Fcur, p, t2)) {
            const double x0 = ((double)p.x - cx0) * s0, y0 = ((double)p.y - cy0) * s0;
            const double x1 = ((double)p.z - cx1) * s1, y1 = ((double)p.w - cy1) * s1;
            const double r[9] = {x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, 1.0};
            int q = 0;
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int j = i; j < 9; ++j) N[q++] += r[i] * r[j];
        }
    }
    double A[9][9];
    {
        int q = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int j = i; j < 9; ++j) {
                const double v = wave_tree_sum(N[q++]);
                A[i][j] = v;
                A[j][i] = v;
            }
    }
    // LU without pivoting (the matrix is symmetric positive definite unless the points are degenerate)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (!(A[k][k] > 0.0)) return false;
#pragma unroll
        for (int r = k + 1; r < 9; ++r) {
            const double f = A[r][k] / A[k][k];
            A[r][k] = f;
#pragma unroll
            for (int c = k + 1; c < 9; ++c) A[r][c] -= f * A[k][c];
        }
    }
    double x[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) x[j] = 1.0;
    for (int it = 0; it < 5; ++it) {
        double y[9], z[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            double sacc = x[r];
#pragma unroll
            for (int c = 0; c < r; ++c) sacc -= A[r][c] * y[c];
            y[r] = sacc;
        }
#pragma unroll
        for (int r = 8; r >= 0; --r) {
            double sacc = y[r];
#pragma unroll
            for (int c = r + 1; c < 9; ++c) sacc -= A[r][c] * z[c];
            z[r] = sacc / A[r][r];
        }
        double nn = 0.0;
#pragma unroll
        for (int j = 0; j < 9; ++j) nn += z[j] * z[j];
        nn = sqrt(nn);
        if (!(nn > 0.0) || !(nn < 1e300)) return false;
#pragma unroll
        for (int j = 0; j < 9; ++j) x[j] = z[j] / nn;
    }
    // F = T1^T Fn T0, unit Frobenius norm, f32, rank 2
    double M[9], G[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        M[3 * i + 0] = x[3 * i + 0] * s0;
        M[3 * i + 1] = x[3 * i + 1] * s0;
        M[3 * i + 2] = x[3 * i + 2] - (M[3 * i + 0] * cx0 + M[3 * i + 1] * cy0);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        G[j] = s1 * M[j];
        G[3 + j] = s1 * M[3 + j];
        G[6 + j] = M[6 + j] - (G[j] * cx1 + G[3 + j] * cy1);
    }
    double fn = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) fn += G[j] * G[j];
    fn = sqrt(fn);
    if (!(fn > 0.0) || !(fn < 1e300)) return false;
    float Fl[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) Fl[j] = (float)(G[j] / fn);
    enforce_rank2(Fl, Fnew);
    return true;
}

// one wave per pair: final matrix, inliers in match order
__global__ __launch_bounds__(64)
void geom_emit_kernel(const GeomPair* __restrict__ pairs, const float4* __restrict__ coords,
                      const int32_t* __restrict__ idx0, const int32_t* __restrict__ idx1, GeomRule rule,
                      const float* __restrict__ Fbest, const int32_t* __restrict__ best_count,
                      float* __restrict__ Fout, int32_t* __restrict__ counts,
                      int32_t* __restrict__ slot_i0, int32_t* __restrict__ slot_i1)
{
    const GeomPair gp = pairs[blockIdx.x];
    const int lane = threadIdx.x;
    const bool have = best_count[blockIdx.x] >= 0;
    const float t2 = rule.max_sampson * rule.max_sampson;
    const float4* __restrict__ pts = coords + gp.off;
    float F[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) F[j] = 0.0f;
    int nin = 0;
    if (have) {
        float Fc[9];
        enforce_rank2(Fbest + (size_t)blockIdx.x * 9, Fc);
        nin = wave_count_inliers(pts, gp.n, Fc, t2, lane);
#pragma unroll
        for (int j = 0; j < 9; ++j) F[j] = Fc[j];
        for (int it = 0; it < kGeomRefits; ++it) {
            float Fn[9];
            const float widen = it == 0 ? 9.0f : it == 1 ? 4.0f : 1.0f;   // threshold^2 of the points this refit uses
            if (!refit_on_inliers(pts, gp.n, Fc, t2 * widen, lane, Fn)) break;
            const int cn = wave_count_inliers(pts, gp.n, Fn, t2, lane);
            if (cn > nin) {
                nin = cn;
#pragma unroll
                for (int j = 0; j < 9; ++j) F[j] = Fn[j];
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) Fc[j] = Fn[j];
        }
    }
    const bool keep = have && nin >= rule.min_inliers && nin > 0;
    if (keep) {
        int base = 0;
        for (int m0 = 0; m0 < gp.n; m0 += 64) {
            const int m = m0 + lane;
            const bool ok = m < gp.n && sampson_inlier(F, pts[m], t2);
            const unsigned long long bal = __ballot(ok);
            if (ok) {
                const size_t o = gp.off + base + __popcll(bal & ((1ull << lane) - 1ull));
                slot_i0[o] = idx0[gp.off + m];
                slot_i1[o] = idx1[gp.off + m];
            }
            base += __popcll(bal);
        }
    }
    if (lane == 0) {
        counts[blockIdx.x] = keep ? nin : 0;
        for (int j = 0; j < 9; ++j) Fout[(size_t)blockIdx.x * 9 + j] = keep ? F[j] : 0.0f;
    }
}

// ---------------------------------------------------------------------------------
// relative pose from a fundamental matrix and the two cameras' intrinsics
// ---------------------------------------------------------------------------------
// What cv2.findEssentialMat + cv2.recoverPose give the reference's estimate_pose
// (Others/SuperPoint_data/models/utils.py:288-312) and what a match record's TwoViewGeometry holds as
// NE, R, T (MatchFile.h:68-83): E = K1^T F K0; its singular vectors (Jacobi on E^T E, the two
// leading left vectors by Gram-Schmidt on E v1, E v2, third vectors as cross products so that both bases
// are right-handed); the four decompositions (U W V^T, +-u3), (U W^T V^T, +-u3); the one with most
// inlier matches in front of both cameras (earliest of equals in that order).  All in double, every
// operation written out in a fixed order so that the scalar checker reproduces the bits.
// Convention: X1 = R X0 + T for a point X0 in the first camera's frame, |T| = 1.
struct PoseCandidates {
    double R[2][9];
    double t[3];
    bool ok;
};

__device__ void pose_candidates(const float* __restrict__ Fin, const float* __restrict__ cam, PoseCandidates& pc)
{
    const double fx0 = cam[0], fy0 = cam[1], cx0 = cam[2], cy0 = cam[3];
    const double fx1 = cam[4], fy1 = cam[5], cx1 = cam[6], cy1 = cam[7];
    double M[9], E[9], S[9], V[9];
    for (int i = 0; i < 3; ++i) {
        const double f0 = (double)Fin[3 * i], f1 = (double)Fin[3 * i + 1], f2 = (double)Fin[3 * i + 2];
        M[3 * i + 0] = f0 * fx0;
        M[3 * i + 1] = f1 * fy0;
        M[3 * i + 2] = (f0 * cx0 + f1 * cy0) + f2;
    }
    for (int j = 0; j < 3; ++j) {
        E[j] = fx1 * M[j];
        E[3 + j] = fy1 * M[3 + j];
        E[6 + j] = (cx1 * M[j] + cy1 * M[3 + j]) + M[6 + j];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            S[3 * i + j] = (E[i] * E[j] + E[3 + i] * E[3 + j]) + E[6 + i] * E[6 + j];
            V[3 * i + j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 6; ++sweep) {
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = S[3 * p + q];
            if (apq == 0.0) continue;
            const double theta = (S[3 * q + q] - S[3 * p + p]) / (2.0 * apq);
            const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
            for (int k = 0; k < 3; ++k) {
                const double skp = S[3 * k + p], skq = S[3 * k + q];
                S[3 * k + p] = c * skp - s * skq;
                S[3 * k + q] = s * skp + c * skq;
            }
            for (int k = 0; k < 3; ++k) {
                const double spk = S[3 * p + k], sqk = S[3 * q + k];
                S[3 * p + k] = c * spk - s * sqk;
                S[3 * q + k] = s * spk + c * sqk;
            }
            for (int k = 0; k < 3; ++k) {
                const double vkp = V[3 * k + p], vkq = V[3 * k + q];
                V[3 * k + p] = c * vkp - s * vkq;
                V[3 * k + q] = s * vkp + c * vkq;
            }
        }
    }
    int im = 0;
    if (S[4] < S[0]) im = 1;
    if (S[8] < S[3 * im + im]) im = 2;
    int ia = im == 0 ? 1 : 0, ib = im == 2 ? 1 : 2;
    if (S[3 * ib + ib] > S[3 * ia + ia]) { const int t = ia; ia = ib; ib = t; }
    const double v1[3] = {V[ia], V[3 + ia], V[6 + ia]}, v2[3] = {V[ib], V[3 + ib], V[6 + ib]};
    const double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
    double a[3], b[3];
    for (int r = 0; r < 3; ++r) {
        a[r] = (E[3 * r] * v1[0] + E[3 * r + 1] * v1[1]) + E[3 * r + 2] * v1[2];
        b[r] = (E[3 * r] * v2[0] + E[3 * r + 1] * v2[1]) + E[3 * r + 2] * v2[2];
    }
    const double na = sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
    pc.ok = false;
    if (!(na > 0.0) || !(na < 1e300)) return;
    const double u1[3] = {a[0] / na, a[1] / na, a[2] / na};
    const double d = (b[0] * u1[0] + b[1] * u1[1]) + b[2] * u1[2];
    b[0] -= d * u1[0]; b[1] -= d * u1[1]; b[2] -= d * u1[2];
    const double nb = sqrt((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]);
    if (!(nb > 0.0)) return;
    const double u2[3] = {b[0] / nb, b[1] / nb, b[2] / nb};
    const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            pc.R[0][3 * r + c] = (u2[r] * v1[c] - u1[r] * v2[c]) + u3[r] * v3[c];   // U W V^T
            pc.R[1][3 * r + c] = (u1[r] * v2[c] - u2[r] * v1[c]) + u3[r] * v3[c];   // U W^T V^T
        }
    pc.t[0] = u3[0]; pc.t[1] = u3[1]; pc.t[2] = u3[2];
    pc.ok = true;
}

// which of the four decompositions put the match (x0, y0) <-> (x1, y1) in front of both cameras: bit
// 2 * rot + (translation negated)
__device__ __forceinline__ unsigned pose_front_bits(const PoseCandidates& pc, const float* __restrict__ cam, const float4 p)
{
    const double x0 = ((double)p.x - (double)cam[2]) / (double)cam[0], y0 = ((double)p.y - (double)cam[3]) / (double)cam[1];
    const double x1 = ((double)p.z - (double)cam[6]) / (double)cam[4], y1 = ((double)p.w - (double)cam[7]) / (double)cam[5];
    // e = x1^ x t (x1^ = (x1, y1, 1))
    const double e0 = y1 * pc.t[2] - pc.t[1], e1 = pc.t[0] - x1 * pc.t[2], e2 = x1 * pc.t[1] - y1 * pc.t[0];
    unsigned bits = 0;
    for (int k = 0; k < 2; ++k) {
        const double* R = pc.R[k];
        const double r0 = (R[0] * x0 + R[1] * y0) + R[2], r1 = (R[3] * x0 + R[4] * y0) + R[5], r2 = (R[6] * x0 + R[7] * y0) + R[8];
        const double c0 = y1 * r2 - r1, c1 = r0 - x1 * r2, c2 = x1 * r1 - y1 * r0;   // x1^ x R x0^
        const double cc = (c0 * c0 + c1 * c1) + c2 * c2;
        if (!(cc > 0.0)) continue;
        const double g = (e0 * c0 + e1 * c1) + e2 * c2;
        const double lp = -g / cc, lm = g / cc;            // depth in the first camera for +t and -t
        if (lp > 0.0 && lp * r2 + pc.t[2] > 0.0) bits |= 1u << (2 * k);
        if (lm > 0.0 && lm * r2 - pc.t[2] > 0.0) bits |= 1u << (2 * k + 1);
    }
    return bits;
}

// one wave per pair
__global__ __launch_bounds__(64)
void geom_pose_kernel(const GeomPair* __restrict__ pairs, const float* __restrict__ cams, const float* __restrict__ Fs,
                      const int32_t* __restrict__ idx0, const int32_t* __restrict__ idx1,
                      float* __restrict__ Rout, float* __restrict__ Tout, int32_t* __restrict__ ne)
{
    const GeomPair gp = pairs[blockIdx.x];
    const int lane = threadIdx.x;
    const float* cam = cams + 8 * (size_t)blockIdx.x;
    const float* F = Fs + 9 * (size_t)blockIdx.x;
    PoseCandidates pc;
    pose_candidates(F, cam, pc);
    int cnt[4] = {0, 0, 0, 0};
    if (pc.ok) {
        for (int m = lane; m < gp.n; m += 64) {
            const int i = idx0[gp.off + m], j = idx1[gp.off + m];
            if (i < 0 || i >= gp.n0 || j < 0 || j >= gp.n1) continue;
            const float4 p = make_float4(gp.xy0[2 * i], gp.xy0[2 * i + 1], gp.xy1[2 * j], gp.xy1[2 * j + 1]);
            const unsigned bits = pose_front_bits(pc, cam, p);
#pragma unroll
            for (int k = 0; k < 4; ++k) cnt[k] += (bits >> k) & 1u;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int d = 32; d > 0; d >>= 1) cnt[k] += __shfl_xor(cnt[k], d);
    int best = 0;
    for (int k = 1; k < 4; ++k)
        if (cnt[k] > cnt[best]) best = k;
    if (lane == 0) {
        const bool have = pc.ok && cnt[best] > 0;
        ne[blockIdx.x] = have ? cnt[best] : 0;
        for (int j = 0; j < 9; ++j) Rout[9 * (size_t)blockIdx.x + j] = have ? (float)pc.R[best >> 1][j] : 0.0f;
        for (int j = 0; j < 3; ++j) Tout[3 * (size_t)blockIdx.x + j] = have ? (float)((best & 1) ? -pc.t[j] : pc.t[j]) : 0.0f;
    }
}

__global__ void geom_pack_pairs_kernel(const GeomPair* __restrict__ gp, int npairs, SiftPair* __restrict__ sp)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    SiftPair s{};
    s.nA = gp[p].n;
    s.out0_off = gp[p].off;
    sp[p] = s;
}

}  // namespace

void launch_geom_pose(const GeomPair* pairs, int npairs, const float* cams, const float* Fs, const int32_t* idx0,
                      const int32_t* idx1, float* R, float* T, int32_t* ne, hipStream_t s)
{
    if (npairs <= 0) return;
    hipLaunchKernelGGL(geom_pose_kernel, dim3(npairs), dim3(64), 0, s, pairs, cams, Fs, idx0, idx1, R, T, ne);
}


void launch_geom_verify(const GeomPair* pairs, int npairs, int max_matches, const int32_t* idx0, const int32_t* idx1,
                        const GeomRule& rule, float4* coords, float* Fh, float* Fbest, int32_t* best_count,
                        float* Fout, int32_t* counts, int32_t* slot_i0, int32_t* slot_i1, SiftPair* pack_pairs,
                        hipStream_t s)
{
    if (npairs <= 0) return;
    const int gx = std::max(1, std::min(64, (max_matches + 255) / 256));
    hipLaunchKernelGGL(geom_gather_kernel, dim3(gx, npairs), dim3(256), 0, s, pairs, npairs, idx0, idx1, coords);
    hipLaunchKernelGGL(geom_solve_kernel, dim3((rule.hypotheses + kGeomThreads - 1) / kGeomThreads, npairs),
                       dim3(kGeomThreads), 0, s, pairs, coords, rule, Fh);
    hipLaunchKernelGGL(geom_score_kernel, dim3(npairs), dim3(kGeomScoreThreads), 0, s,
                       pairs, coords, rule, Fh, Fbest, best_count);
    hipLaunchKernelGGL(geom_emit_kernel, dim3(npairs), dim3(64), 0, s,
                       pairs, coords, idx0, idx1, rule, Fbest, best_count, Fout, counts, slot_i0, slot_i1);
    hipLaunchKernelGGL(geom_pack_pairs_kernel, dim3((npairs + 255) / 256), dim3(256), 0, s, pairs, npairs, pack_pairs);
}

}  // namespace vsm
