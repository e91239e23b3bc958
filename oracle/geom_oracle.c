/* CPU checker for the geometric verification of putative matches.  TEST INFRASTRUCTURE ONLY: nothing
 * in the product library links, loads or calls this file.
 *
 * The reference has no estimator of its own for this step: its C++ side stores what an estimator
 * returns (TwoViewGeometry and the inlier lists of a match record, MatchFile.h:68-161,
 * MatchFile.cpp:596-748) and its Python side calls OpenCV, which is not installed here
 * (cv2.findEssentialMat with RANSAC, models/utils.py:288-312).  PARITY UNPINNED: no fixture or output
 * of the reference covers the estimate.  What this file pins is the library's own stated algorithm,
 * restated in scalar C with plain arrays, operation for operation:
 *   sample h of the pair at position p of a call: eight distinct matches from splitmix64 counters;
 *   Hartley normalisation, 8 x 9 system of x1^T F x0 = 0, Gaussian elimination with complete
 *   pivoting, free unknown 1, unit norm, de-normalisation, unit Frobenius norm (double);
 *   inliers by e^2 < t^2 (a0^2 + a1^2 + b0^2 + b1^2) in float with explicit fused multiply-adds;
 *   most inliers wins, lowest h on ties;
 *   the winner projected to rank 2 (Jacobi on F^T F); four refits (least-squares eight-point fit to the
 *   matches within 3, 2, 1, 1 thresholds of the present matrix: 64 interleaved partial sums joined by a
 *   binary tree, unpivoted LU of the normal matrix, five inverse iterations from (1, ..., 1)); the matrix
 *   with most inliers among the five (earliest of equals), its inliers listed in match order. */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static int inlier(const float* F, const float* p, float t2)
{
    const float a0 = fmaf(F[0], p[0], fmaf(F[1], p[1], F[2]));
    const float a1 = fmaf(F[3], p[0], fmaf(F[4], p[1], F[5]));
    const float a2 = fmaf(F[6], p[0], fmaf(F[7], p[1], F[8]));
    const float b0 = fmaf(F[0], p[2], fmaf(F[3], p[3], F[6]));
    const float b1 = fmaf(F[1], p[2], fmaf(F[4], p[3], F[7]));
    const float e = fmaf(p[2], a0, fmaf(p[3], a1, a2));
    const float den = fmaf(b1, b1, fmaf(b0, b0, fmaf(a1, a1, a0 * a0)));
    return e * e < t2 * den;
}

/* eight matches (x0, y0, x1, y1) -> F (float, unit Frobenius norm); 0 if the sample is degenerate */
static int eight_point(const float pt[8][4], float* Fout)
{
    double c[4] = {0, 0, 0, 0};
    for (int k = 0; k < 8; ++k)
        for (int d = 0; d < 4; ++d) c[d] += (double)pt[k][d];
    for (int d = 0; d < 4; ++d) c[d] /= 8.0;
    double md0 = 0, md1 = 0;
    for (int k = 0; k < 8; ++k) {
        const double dx0 = (double)pt[k][0] - c[0], dy0 = (double)pt[k][1] - c[1];
        const double dx1 = (double)pt[k][2] - c[2], dy1 = (double)pt[k][3] - c[3];
        md0 += sqrt(dx0 * dx0 + dy0 * dy0);
        md1 += sqrt(dx1 * dx1 + dy1 * dy1);
    }
    md0 /= 8.0;
    md1 /= 8.0;
    if (!(md0 > 0.0) || !(md1 > 0.0)) return 0;
    const double s0 = 1.4142135623730951 / md0, s1 = 1.4142135623730951 / md1;
    double A[8][9];
    for (int k = 0; k < 8; ++k) {
        const double x0 = ((double)pt[k][0] - c[0]) * s0, y0 = ((double)pt[k][1] - c[1]) * s0;
        const double x1 = ((double)pt[k][2] - c[2]) * s1, y1 = ((double)pt[k][3] - c[3]) * s1;
        A[k][0] = x1 * x0; A[k][1] = x1 * y0; A[k][2] = x1;
        A[k][3] = y1 * x0; A[k][4] = y1 * y0; A[k][5] = y1;
        A[k][6] = x0;      A[k][7] = y0;      A[k][8] = 1.0;
    }
    int perm[9];
    for (int j = 0; j < 9; ++j) perm[j] = j;
    for (int k = 0; k < 8; ++k) {
        double best = 0.0;
        int pr = k, pc = k;
        for (int r = k; r < 8; ++r)
            for (int cc = k; cc < 9; ++cc) {
                const double v = fabs(A[r][cc]);
                if (v > best) { best = v; pr = r; pc = cc; }
            }
        if (!(best > 0.0)) return 0;
        if (pr != k)
            for (int cc = 0; cc < 9; ++cc) { const double t = A[k][cc]; A[k][cc] = A[pr][cc]; A[pr][cc] = t; }
        if (pc != k) {
            for (int r = 0; r < 8; ++r) { const double t = A[r][k]; A[r][k] = A[r][pc]; A[r][pc] = t; }
            const int t = perm[k]; perm[k] = perm[pc]; perm[pc] = t;
        }
        const double piv = A[k][k];
        for (int r = k + 1; r < 8; ++r) {
            const double f = A[r][k] / piv;
            for (int cc = k + 1; cc < 9; ++cc) A[r][cc] -= f * A[k][cc];
        }
    }
    double x[9];
    x[8] = 1.0;
    for (int k = 7; k >= 0; --k) {
        double s = 0.0;
        for (int cc = k + 1; cc < 9; ++cc) s += A[k][cc] * x[cc];
        x[k] = -s / A[k][k];
    }
    double f[9], nn = 0.0;
    for (int j = 0; j < 9; ++j) f[perm[j]] = x[j];
    for (int j = 0; j < 9; ++j) nn += f[j] * f[j];
    nn = sqrt(nn);
    if (!(nn > 0.0)) return 0;
    for (int j = 0; j < 9; ++j) f[j] /= nn;
    double M[9], G[9];
    for (int i = 0; i < 3; ++i) {
        M[3 * i + 0] = f[3 * i + 0] * s0;
        M[3 * i + 1] = f[3 * i + 1] * s0;
        M[3 * i + 2] = f[3 * i + 2] - (M[3 * i + 0] * c[0] + M[3 * i + 1] * c[1]);
    }
    for (int j = 0; j < 3; ++j) {
        G[j] = s1 * M[j];
        G[3 + j] = s1 * M[3 + j];
        G[6 + j] = M[6 + j] - (G[j] * c[2] + G[3 + j] * c[3]);
    }
    double fn = 0.0;
    for (int j = 0; j < 9; ++j) fn += G[j] * G[j];
    fn = sqrt(fn);
    if (!(fn > 0.0) || !(fn < 1e300)) return 0;
    for (int j = 0; j < 9; ++j) Fout[j] = (float)(G[j] / fn);
    return 1;
}

static void rank2(const float* Fin, float* Fout)
{
    double F[9], S[9], V[9];
    for (int j = 0; j < 9; ++j) F[j] = (double)Fin[j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            S[3 * i + j] = (F[i] * F[j] + F[3 + i] * F[3 + j]) + F[6 + i] * F[6 + j];
            V[3 * i + j] = i == j ? 1.0 : 0.0;
        }
    static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int sweep = 0; sweep < 6; ++sweep)
        for (int pq = 0; pq < 3; ++pq) {
            const int p = P[pq], q = Q[pq];
            const double apq = S[3 * p + q];
            if (apq == 0.0) continue;
            const double theta = (S[3 * q + q] - S[3 * p + p]) / (2.0 * apq);
            const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
            for (int k = 0; k < 3; ++k) {
                const double a = S[3 * k + p], b = S[3 * k + q];
                S[3 * k + p] = c * a - s * b;
                S[3 * k + q] = s * a + c * b;
            }
            for (int k = 0; k < 3; ++k) {
                const double a = S[3 * p + k], b = S[3 * q + k];
                S[3 * p + k] = c * a - s * b;
                S[3 * q + k] = s * a + c * b;
            }
            for (int k = 0; k < 3; ++k) {
                const double a = V[3 * k + p], b = V[3 * k + q];
                V[3 * k + p] = c * a - s * b;
                V[3 * k + q] = s * a + c * b;
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

/* acc[t] += acc[t + d] for d = 32, 16, ..., 1 */
static double tree64(double* acc)
{
    for (int d = 32; d > 0; d >>= 1)
        for (int t = 0; t < d; ++t) acc[t] += acc[t + d];
    return acc[0];
}

static int count_inliers(const float* pts, int n, const float* F, float t2)
{
    int c = 0;
    for (int m = 0; m < n; ++m) c += inlier(F, pts + 4 * m, t2);
    return c;
}

static int refit(const float* pts, int n, const float* Fcur, float t2, float* Fnew)
{
    double acc[5][64];
    memset(acc, 0, sizeof acc);
    for (int m = 0; m < n; ++m)
        if (inlier(Fcur, pts + 4 * m, t2)) {
            const int t = m & 63;
            acc[0][t] += 1.0;
            for (int d = 0; d < 4; ++d) acc[1 + d][t] += (double)pts[4 * m + d];
        }
    const double cnt = tree64(acc[0]);
    if (!(cnt >= 8.0)) return 0;
    double c[4];
    for (int d = 0; d < 4; ++d) c[d] = tree64(acc[1 + d]) / cnt;
    double a0[64], a1[64];
    memset(a0, 0, sizeof a0);
    memset(a1, 0, sizeof a1);
    for (int m = 0; m < n; ++m)
        if (inlier(Fcur, pts + 4 * m, t2)) {
            const float* p = pts + 4 * m;
            const double dx0 = (double)p[0] - c[0], dy0 = (double)p[1] - c[1];
            const double dx1 = (double)p[2] - c[2], dy1 = (double)p[3] - c[3];
            a0[m & 63] += sqrt(dx0 * dx0 + dy0 * dy0);
            a1[m & 63] += sqrt(dx1 * dx1 + dy1 * dy1);
        }
    const double md0 = tree64(a0) / cnt, md1 = tree64(a1) / cnt;
    if (!(md0 > 0.0) || !(md1 > 0.0)) return 0;
    const double s0 = 1.4142135623730951 / md0, s1 = 1.4142135623730951 / md1;
    double N[45][64];
    memset(N, 0, sizeof N);
    for (int m = 0; m < n; ++m)
        if (inlier(Fcur, pts + 4 * m, t2)) {
            const float* p = pts + 4 * m;
            const double x0 = ((double)p[0] - c[0]) * s0, y0 = ((double)p[1] - c[1]) * s0;
            const double x1 = ((double)p[2] - c[2]) * s1, y1 = ((double)p[3] - c[3]) * s1;
            const double r[9] = {x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, 1.0};
            int q = 0;
            for (int i = 0; i < 9; ++i)
                for (int j = i; j < 9; ++j) N[q++][m & 63] += r[i] * r[j];
        }
    double A[9][9];
    {
        int q = 0;
        for (int i = 0; i < 9; ++i)
            for (int j = i; j < 9; ++j) {
                const double v = tree64(N[q++]);
                A[i][j] = v;
                A[j][i] = v;
            }
    }
    for (int k = 0; k < 9; ++k) {
        if (!(A[k][k] > 0.0)) return 0;
        for (int r = k + 1; r < 9; ++r) {
            const double f = A[r][k] / A[k][k];
            A[r][k] = f;
            for (int cc = k + 1; cc < 9; ++cc) A[r][cc] -= f * A[k][cc];
        }
    }
    double x[9];
    for (int j = 0; j < 9; ++j) x[j] = 1.0;
    for (int it = 0; it < 5; ++it) {
        double y[9], z[9];
        for (int r = 0; r < 9; ++r) {
            double s = x[r];
            for (int cc = 0; cc < r; ++cc) s -= A[r][cc] * y[cc];
            y[r] = s;
        }
        for (int r = 8; r >= 0; --r) {
            double s = y[r];
            for (int cc = r + 1; cc < 9; ++cc) s -= A[r][cc] * z[cc];
            z[r] = s / A[r][r];
        }
        double nn = 0.0;
        for (int j = 0; j < 9; ++j) nn += z[j] * z[j];
        nn = sqrt(nn);
        if (!(nn > 0.0) || !(nn < 1e300)) return 0;
        for (int j = 0; j < 9; ++j) x[j] = z[j] / nn;
    }
    double M[9], G[9];
    for (int i = 0; i < 3; ++i) {
        M[3 * i + 0] = x[3 * i + 0] * s0;
        M[3 * i + 1] = x[3 * i + 1] * s0;
        M[3 * i + 2] = x[3 * i + 2] - (M[3 * i + 0] * c[0] + M[3 * i + 1] * c[1]);
    }
    for (int j = 0; j < 3; ++j) {
        G[j] = s1 * M[j];
        G[3 + j] = s1 * M[3 + j];
        G[6 + j] = M[6 + j] - (G[j] * c[2] + G[3 + j] * c[3]);
    }
    double fn = 0.0;
    for (int j = 0; j < 9; ++j) fn += G[j] * G[j];
    fn = sqrt(fn);
    if (!(fn > 0.0) || !(fn < 1e300)) return 0;
    float Fl[9];
    for (int j = 0; j < 9; ++j) Fl[j] = (float)(G[j] / fn);
    rank2(Fl, Fnew);
    return 1;
}

/* One pair.  xy0 / xy1: keypoints (n0 / n1 rows of x, y); idx0 / idx1: n putative matches.
 * pair_key: the pair's sampling key (key of the call's first pair + position in the call).
 * Out: F[9] (zeros if the pair reports nothing), in0 / in1 (capacity n).  Returns NF. */
int oracle_geom_pair(const float* xy0, int n0, const float* xy1, int n1, const int32_t* idx0, const int32_t* idx1, int n,
                     float max_sampson, int hypotheses, unsigned seed, unsigned long long pair_key, int min_inliers,
                     float* F, int32_t* in0, int32_t* in1)
{
    for (int j = 0; j < 9; ++j) F[j] = 0.0f;
    if (n < 8) return 0;
    if (hypotheses <= 0) hypotheses = 256;
    float* pts = (float*)malloc((size_t)n * 4 * sizeof(float));
    for (int m = 0; m < n; ++m) {
        const int i = idx0[m], j = idx1[m];
        if (i >= 0 && i < n0 && j >= 0 && j < n1) {
            pts[4 * m + 0] = xy0[2 * i]; pts[4 * m + 1] = xy0[2 * i + 1];
            pts[4 * m + 2] = xy1[2 * j]; pts[4 * m + 3] = xy1[2 * j + 1];
        } else {
            for (int d = 0; d < 4; ++d) pts[4 * m + d] = NAN;
        }
    }
    const float t2 = max_sampson * max_sampson;
    const uint64_t base = splitmix64(((uint64_t)seed << 32) ^ (uint64_t)pair_key);
    int* counts = (int*)malloc((size_t)hypotheses * sizeof(int));
    float* Fs = (float*)malloc((size_t)hypotheses * 9 * sizeof(float));
#pragma omp parallel for schedule(dynamic, 4)
    for (int h = 0; h < hypotheses; ++h) {
        counts[h] = -1;
        int pick[8], got = 0;
        for (int ctr = 0; ctr < 64 && got < 8; ++ctr) {
            const uint64_t r = splitmix64(base + ((uint64_t)h << 8) + (uint64_t)ctr);
            const int idx = (int)(((r >> 32) * (uint64_t)n) >> 32);
            int dup = 0;
            for (int k = 0; k < got; ++k) dup |= pick[k] == idx;
            if (!dup) pick[got++] = idx;
        }
        if (got < 8) continue;
        float pt[8][4];
        for (int k = 0; k < 8; ++k) memcpy(pt[k], pts + 4 * pick[k], 4 * sizeof(float));
        float Fh[9];
        if (!eight_point(pt, Fh)) continue;
        int cnt = 0;
        for (int m = 0; m < n; ++m) cnt += inlier(Fh, pts + 4 * m, t2);
        counts[h] = cnt;
        memcpy(Fs + 9 * h, Fh, sizeof Fh);
    }
    int best = -1, bh = -1;
    for (int h = 0; h < hypotheses; ++h)
        if (counts[h] > best) { best = counts[h]; bh = h; }
    int nf = 0;
    if (bh >= 0) {
        float Ff[9], Fc[9], Fn[9];
        rank2(Fs + 9 * bh, Fc);
        int nin = count_inliers(pts, n, Fc, t2);
        memcpy(Ff, Fc, sizeof Ff);
        static const float widen[4] = {9.0f, 4.0f, 1.0f, 1.0f};   /* threshold^2 of the points a refit uses */
        for (int it = 0; it < 4; ++it) {
            if (!refit(pts, n, Fc, t2 * widen[it], Fn)) break;
            const int cn = count_inliers(pts, n, Fn, t2);
            if (cn > nin) {
                nin = cn;
                memcpy(Ff, Fn, sizeof Ff);
            }
            memcpy(Fc, Fn, sizeof Fc);
        }
        for (int m = 0; m < n; ++m)
            if (inlier(Ff, pts + 4 * m, t2)) {
                in0[nf] = idx0[m];
                in1[nf] = idx1[m];
                ++nf;
            }
        if (nf >= min_inliers && nf > 0) memcpy(F, Ff, sizeof Ff);
        else nf = 0;
    }
    free(pts);
    free(counts);
    free(Fs);
    return nf;
}

/* ---- relative pose from F and the intrinsics ------------------------------------------------
 * The library's stated procedure, restated (the reference gets R, t from OpenCV's findEssentialMat +
 * recoverPose, models/utils.py:288-312: PARITY UNPINNED, as for F): E = K1^T F K0; Jacobi on E^T E;
 * v1, v2 = eigenvectors of the two largest eigenvalues, v3 = v1 x v2; u1 = E v1 / |E v1|, u2 = E v2
 * orthogonalised against u1 and normalised, u3 = u1 x u2; candidates (U W V^T, +-u3), (U W^T V^T, +-u3);
 * the one with most matches in front of both cameras, earliest of equals.  All double, fixed order. */
typedef struct {
    double R[2][9];
    double t[3];
    int ok;
} pose_cand;

static void pose_candidates(const float* Fin, const float* cam, pose_cand* pc)
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
    static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int sweep = 0; sweep < 6; ++sweep)
        for (int pq = 0; pq < 3; ++pq) {
            const int p = P[pq], q = Q[pq];
            const double apq = S[3 * p + q];
            if (apq == 0.0) continue;
            const double theta = (S[3 * q + q] - S[3 * p + p]) / (2.0 * apq);
            const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
            for (int k = 0; k < 3; ++k) {
                const double a = S[3 * k + p], b = S[3 * k + q];
                S[3 * k + p] = c * a - s * b;
                S[3 * k + q] = s * a + c * b;
            }
            for (int k = 0; k < 3; ++k) {
                const double a = S[3 * p + k], b = S[3 * q + k];
                S[3 * p + k] = c * a - s * b;
                S[3 * q + k] = s * a + c * b;
            }
            for (int k = 0; k < 3; ++k) {
                const double a = V[3 * k + p], b = V[3 * k + q];
                V[3 * k + p] = c * a - s * b;
                V[3 * k + q] = s * a + c * b;
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
    pc->ok = 0;
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
            pc->R[0][3 * r + c] = (u2[r] * v1[c] - u1[r] * v2[c]) + u3[r] * v3[c];
            pc->R[1][3 * r + c] = (u1[r] * v2[c] - u2[r] * v1[c]) + u3[r] * v3[c];
        }
    pc->t[0] = u3[0]; pc->t[1] = u3[1]; pc->t[2] = u3[2];
    pc->ok = 1;
}

static unsigned pose_front_bits(const pose_cand* pc, const float* cam, const float* p)
{
    const double x0 = ((double)p[0] - (double)cam[2]) / (double)cam[0], y0 = ((double)p[1] - (double)cam[3]) / (double)cam[1];
    const double x1 = ((double)p[2] - (double)cam[6]) / (double)cam[4], y1 = ((double)p[3] - (double)cam[7]) / (double)cam[5];
    const double e0 = y1 * pc->t[2] - pc->t[1], e1 = pc->t[0] - x1 * pc->t[2], e2 = x1 * pc->t[1] - y1 * pc->t[0];
    unsigned bits = 0;
    for (int k = 0; k < 2; ++k) {
        const double* R = pc->R[k];
        const double r0 = (R[0] * x0 + R[1] * y0) + R[2], r1 = (R[3] * x0 + R[4] * y0) + R[5], r2 = (R[6] * x0 + R[7] * y0) + R[8];
        const double c0 = y1 * r2 - r1, c1 = r0 - x1 * r2, c2 = x1 * r1 - y1 * r0;
        const double cc = (c0 * c0 + c1 * c1) + c2 * c2;
        if (!(cc > 0.0)) continue;
        const double g = (e0 * c0 + e1 * c1) + e2 * c2;
        const double lp = -g / cc, lm = g / cc;
        if (lp > 0.0 && lp * r2 + pc->t[2] > 0.0) bits |= 1u << (2 * k);
        if (lm > 0.0 && lm * r2 - pc->t[2] > 0.0) bits |= 1u << (2 * k + 1);
    }
    return bits;
}

/* cam: fx, fy, cx, cy of image 0 then image 1.  idx0 / idx1: the n inlier matches.  Returns NE. */
int oracle_pose_pair(const float* xy0, int n0, const float* xy1, int n1, const int32_t* idx0, const int32_t* idx1, int n,
                     const float* F, const float* cam, float* R, float* T)
{
    for (int j = 0; j < 9; ++j) R[j] = 0.0f;
    for (int j = 0; j < 3; ++j) T[j] = 0.0f;
    pose_cand pc;
    pose_candidates(F, cam, &pc);
    if (!pc.ok) return 0;
    int cnt[4] = {0, 0, 0, 0};
    for (int m = 0; m < n; ++m) {
        const int i = idx0[m], j = idx1[m];
        if (i < 0 || i >= n0 || j < 0 || j >= n1) continue;
        const float p[4] = {xy0[2 * i], xy0[2 * i + 1], xy1[2 * j], xy1[2 * j + 1]};
        const unsigned bits = pose_front_bits(&pc, cam, p);
        for (int k = 0; k < 4; ++k) cnt[k] += (bits >> k) & 1u;
    }
    int best = 0;
    for (int k = 1; k < 4; ++k)
        if (cnt[k] > cnt[best]) best = k;
    if (cnt[best] <= 0) return 0;
    for (int j = 0; j < 9; ++j) R[j] = (float)pc.R[best >> 1][j];
    for (int j = 0; j < 3; ++j) T[j] = (float)((best & 1) ? -pc.t[j] : pc.t[j]);
    return cnt[best];
}


/* Threads of the loop over samples in oracle_geom_pair (the other oracles take the count as an argument). */
void oracle_geom_set_threads(int threads) { omp_set_num_threads(threads > 0 ? threads : 1); }
