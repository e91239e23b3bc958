/*
 * CPU oracle for the descriptor-matching hot path.  TEST INFRASTRUCTURE ONLY:
 * see oracle.h for what may use it and for which parts the reference pins.
 */
#include "oracle.h"

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#define DESC_DIM 128

/* ------------------------------------------------------------------------- */
/* SIFT-128 u8                                                               */
/* ------------------------------------------------------------------------- */

static inline int32_t l2sq_u8(const uint8_t* a, const uint8_t* b)
{
    int32_t s = 0;
    for (int k = 0; k < DESC_DIM; ++k) {
        int32_t d = (int32_t)a[k] - (int32_t)b[k];
        s += d * d;
    }
    return s;
}

void orc_knn2_u8(const uint8_t* A, int n, const uint8_t* B, int m, int threads,
                 int32_t* idx1, int32_t* d1, int32_t* d2)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int i = 0; i < n; ++i) {
        int32_t b1 = ORC_NO_SECOND, b2 = ORC_NO_SECOND, bi = -1;
        const uint8_t* a = A + (size_t)i * DESC_DIM;
        for (int j = 0; j < m; ++j) {
            int32_t d = l2sq_u8(a, B + (size_t)j * DESC_DIM);
            if (bi < 0 || d < b1) { /* strict: the first (lowest) index keeps a tie */
                b2 = b1;
                b1 = d;
                bi = j;
            } else if (d < b2) {
                b2 = d;
            }
        }
        idx1[i] = bi;
        d1[i] = (bi < 0) ? ORC_NO_SECOND : b1;
        d2[i] = b2;
    }
}

void orc_colbest_u8(const uint8_t* A, int n, const uint8_t* B, int m, int threads,
                    int32_t* cidx, int32_t* cd)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int j = 0; j < m; ++j) {
        int32_t b1 = ORC_NO_SECOND, bi = -1;
        const uint8_t* b = B + (size_t)j * DESC_DIM;
        for (int i = 0; i < n; ++i) {
            int32_t d = l2sq_u8(A + (size_t)i * DESC_DIM, b);
            if (bi < 0 || d < b1) {
                b1 = d;
                bi = i;
            }
        }
        cidx[j] = bi;
        cd[j] = b1;
    }
}

static int accept_u8(int32_t d1, int32_t d2, const orc_params* p)
{
    if (p->max_dist_sq >= 0 && (int64_t)d1 > p->max_dist_sq) return 0;
    if (p->ratio_num > 0 && d2 != ORC_NO_SECOND) {
        int64_t lhs = (int64_t)d1 * p->ratio_den * p->ratio_den;
        int64_t rhs = (int64_t)d2 * p->ratio_num * p->ratio_num;
        if (!(lhs < rhs)) return 0;
    }
    return 1;
}

void orc_match_dense_u8(const uint8_t* A, int n, const uint8_t* B, int m, const orc_params* p,
                        int threads, int32_t* matches0, int32_t* matches1)
{
    for (int j = 0; j < m; ++j) matches1[j] = -1;
    for (int i = 0; i < n; ++i) matches0[i] = -1;
    if (n <= 0 || m <= 0) return;
    int32_t* idx1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)n * 3);
    int32_t* d1 = idx1 + n;
    int32_t* d2 = d1 + n;
    int32_t* cidx = (int32_t*)malloc(sizeof(int32_t) * (size_t)m * 2);
    int32_t* cd = cidx + m;
    orc_knn2_u8(A, n, B, m, threads, idx1, d1, d2);
    if (p->mutual) orc_colbest_u8(A, n, B, m, threads, cidx, cd);
    for (int i = 0; i < n; ++i) {
        int j = idx1[i];
        if (j < 0) continue;
        if (!accept_u8(d1[i], d2[i], p)) continue;
        if (p->mutual && cidx[j] != i) continue;
        matches0[i] = j;
        /* without the mutual check several queries may claim one train row;
         * the lowest query index is reported on the train side */
        if (matches1[j] < 0) matches1[j] = i;
    }
    free(idx1);
    free(cidx);
}

int orc_match_u8(const uint8_t* A, int n, const uint8_t* B, int m, const orc_params* p,
                 int threads, int32_t* idx0, int32_t* idx1, int32_t* dist, int cap)
{
    if (n <= 0 || m <= 0) return 0;
    int32_t* m0 = (int32_t*)malloc(sizeof(int32_t) * ((size_t)n + (size_t)m));
    int32_t* m1 = m0 + n;
    orc_match_dense_u8(A, n, B, m, p, threads, m0, m1);
    int nm = 0;
    for (int i = 0; i < n; ++i) {
        if (m0[i] < 0) continue;
        if (nm < cap) {
            if (idx0) idx0[nm] = i;
            if (idx1) idx1[nm] = m0[i];
            if (dist) dist[nm] = l2sq_u8(A + (size_t)i * DESC_DIM, B + (size_t)m0[i] * DESC_DIM);
        }
        ++nm;
    }
    free(m0);
    return nm;
}

/* ------------------------------------------------------------------------- */
/* float32                                                                   */
/* ------------------------------------------------------------------------- */

float orc_dot_f32(const float* a, const float* b, int dim)
{
    float s = 0.0f;
    for (int k = 0; k < dim; ++k) s = fmaf(a[k], b[k], s);
    return s;
}

static inline int better_f32(float cand, float best, int metric)
{
    return metric == ORC_METRIC_DOT ? (cand > best) : (cand < best);
}

static inline float key_f32(const float* a, const float* b, int dim, int metric, float nb)
{
    float d = orc_dot_f32(a, b, dim);
    return metric == ORC_METRIC_DOT ? d : fmaf(-2.0f, d, nb);
}

static float* sqnorms_f32(const float* X, int n, int dim)
{
    float* r = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) r[i] = orc_dot_f32(X + (size_t)i * dim, X + (size_t)i * dim, dim);
    return r;
}

static inline float worst_f32(int metric) { return metric == ORC_METRIC_DOT ? -INFINITY : INFINITY; }

/* Squared distance in difference form, k ascending: the form the reference ranks by
 * (model_2_0.py:68, torch.norm(x - emb, 2, 1)); no cancellation for near-duplicates. */
static inline float diffsq_f32(const float* a, const float* b, int dim)
{
    float d = 0.0f;
    for (int k = 0; k < dim; ++k) {
        const float t = a[k] - b[k];
        d = fmaf(t, t, d);
    }
    return d;
}

/* L2 searches: the two best rows are FOUND by the key nb - 2 a.b (what the matrix cores rank by; first
 * index on equal keys), then ORDERED, and their distances reported, in difference form: s1 <= s2, and on
 * equal distances the lower row first. */
static void l2_finish(const float* a, const float* B, int dim, int32_t* bi, int32_t bi2, float* s1, float* s2)
{
    if (*bi < 0) return;
    float d1 = diffsq_f32(a, B + (size_t)*bi * dim, dim);
    if (bi2 < 0) {
        *s1 = d1;
        return;
    }
    float d2 = diffsq_f32(a, B + (size_t)bi2 * dim, dim);
    if (d2 < d1 || (d2 == d1 && bi2 < *bi)) {
        float t = d1; d1 = d2; d2 = t;
        *bi = bi2;
    }
    *s1 = d1;
    *s2 = d2;
}

void orc_knn2_f32(const float* A, int n, const float* B, int m, int dim, int metric,
                  int threads, int32_t* idx1, float* s1, float* s2)
{
    if (threads < 1) threads = 1;
    float* nb = sqnorms_f32(B, m, dim);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int i = 0; i < n; ++i) {
        float b1 = worst_f32(metric), b2 = worst_f32(metric);
        int32_t bi = -1, bi2 = -1;
        const float* a = A + (size_t)i * dim;
        for (int j = 0; j < m; ++j) {
            float k = key_f32(a, B + (size_t)j * dim, dim, metric, nb[j]);
            if (bi < 0 || better_f32(k, b1, metric)) {
                b2 = b1; bi2 = bi;
                b1 = k;
                bi = j;
            } else if (bi2 < 0 || better_f32(k, b2, metric)) {
                b2 = k; bi2 = j;
            }
        }
        if (metric == ORC_METRIC_L2) l2_finish(a, B, dim, &bi, bi2, &b1, &b2);
        idx1[i] = bi;
        s1[i] = b1;
        s2[i] = b2;
    }
    free(nb);
}

void orc_colbest_f32(const float* A, int n, const float* B, int m, int dim, int metric,
                     int threads, int32_t* cidx, float* cs)
{
    /* the train-side best is the search with the roles exchanged (every row of B against A) */
    float* s2 = (float*)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1));
    orc_knn2_f32(B, m, A, n, dim, metric, threads, cidx, cs, s2);
    free(s2);
}

static int accept_f32(float s1, float s2, int have2, int metric, const orc_params* p)
{
    if (metric == ORC_METRIC_DOT) {
        if (!(s1 > p->min_score)) return 0;
        return 1; /* the ratio test is defined on distances only */
    }
    if (p->max_dist_sq_f >= 0.0f && s1 > p->max_dist_sq_f) return 0;
    if (p->ratio_num > 0 && have2) {
        double lhs = (double)s1 * (double)p->ratio_den * (double)p->ratio_den;
        double rhs = (double)s2 * (double)p->ratio_num * (double)p->ratio_num;
        if (!(lhs < rhs)) return 0;
    }
    return 1;
}

void orc_match_dense_f32(const float* A, int n, const float* B, int m, int dim, int metric,
                         const orc_params* p, int threads,
                         int32_t* matches0, int32_t* matches1, float* best0, float* best1)
{
    for (int j = 0; j < m; ++j) matches1[j] = -1;
    for (int i = 0; i < n; ++i) matches0[i] = -1;
    if (n <= 0 || m <= 0) return;
    int32_t* idx1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    float* s1 = (float*)malloc(sizeof(float) * (size_t)n * 2);
    float* s2 = s1 + n;
    int32_t* cidx = (int32_t*)malloc(sizeof(int32_t) * (size_t)m);
    float* cs = (float*)malloc(sizeof(float) * (size_t)m);
    orc_knn2_f32(A, n, B, m, dim, metric, threads, idx1, s1, s2);
    orc_colbest_f32(A, n, B, m, dim, metric, threads, cidx, cs);
    for (int i = 0; i < n; ++i) {
        if (best0) best0[i] = s1[i];
        int j = idx1[i];
        if (j < 0) continue;
        if (!accept_f32(s1[i], s2[i], m >= 2, metric, p)) continue;
        if (p->mutual && cidx[j] != i) continue;
        matches0[i] = j;
        if (matches1[j] < 0) matches1[j] = i;
    }
    if (best1)
        for (int j = 0; j < m; ++j) best1[j] = cs[j];
    free(idx1);
    free(s1);
    free(cidx);
    free(cs);
}

int orc_match_f32(const float* A, int n, const float* B, int m, int dim, int metric,
                  const orc_params* p, int threads,
                  int32_t* idx0, int32_t* idx1, float* score, int cap)
{
    if (n <= 0 || m <= 0) return 0;
    int32_t* m0 = (int32_t*)malloc(sizeof(int32_t) * ((size_t)n + (size_t)m));
    int32_t* m1 = m0 + n;
    float* b0 = (float*)malloc(sizeof(float) * (size_t)n);
    orc_match_dense_f32(A, n, B, m, dim, metric, p, threads, m0, m1, b0, NULL);
    int nm = 0;
    for (int i = 0; i < n; ++i) {
        if (m0[i] < 0) continue;
        if (nm < cap) {
            if (idx0) idx0[nm] = i;
            if (idx1) idx1[nm] = m0[i];
            if (score) score[nm] = b0[i];
        }
        ++nm;
    }
    free(m0);
    free(b0);
    return nm;
}

void orc_nearest_embed_f32(const float* x, int batch, int dim, int npts,
                           const float* emb, int k, int threads,
                           int64_t* argmin, float* result)
{
    /* gather to row form, then L2 1-NN (model_2_0.py:68-70: norm of the
     * difference, min over the code axis, first index on ties) */
    if (threads < 1) threads = 1;
    float* E = (float*)malloc(sizeof(float) * (size_t)k * dim);
    for (int c = 0; c < k; ++c)
        for (int d = 0; d < dim; ++d) E[(size_t)c * dim + d] = emb[(size_t)d * k + c];
    float* ne = sqnorms_f32(E, k, dim);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int bp = 0; bp < batch * npts; ++bp) {
        int b = bp / npts, pt = bp % npts;
        float* q = (float*)malloc(sizeof(float) * (size_t)dim);
        for (int d = 0; d < dim; ++d) q[d] = x[((size_t)b * dim + d) * npts + pt];
        float best = INFINITY, second = INFINITY;
        int32_t bi = -1, bi2 = -1;
        for (int c = 0; c < k; ++c) {
            float key = key_f32(q, E + (size_t)c * dim, dim, ORC_METRIC_L2, ne[c]);
            if (bi < 0 || key < best) {
                second = best; bi2 = bi;
                best = key;
                bi = c;
            } else if (bi2 < 0 || key < second) {
                second = key; bi2 = c;
            }
        }
        l2_finish(q, E, dim, &bi, bi2, &best, &second);
        argmin[bp] = bi;
        if (result)
            for (int d = 0; d < dim; ++d)
                result[((size_t)b * dim + d) * npts + pt] = E[(size_t)bi * dim + d];
        free(q);
    }
    free(E);
    free(ne);
}

void orc_merge_top2_f32(const float* k1, const float* k2, const int32_t* i1, const int32_t* i2,
                        int nslices, int nq, int metric,
                        float* ok1, float* ok2, int32_t* oi1, int32_t* oi2)
{
    for (int q = 0; q < nq; ++q) {
        float b1 = worst_f32(metric), b2 = worst_f32(metric);
        int32_t x1 = -1, x2 = -1;
        for (int s = 0; s < nslices; ++s) {
            for (int r = 0; r < 2; ++r) {
                float k = r ? k2[(size_t)s * nq + q] : k1[(size_t)s * nq + q];
                int32_t id = r ? i2[(size_t)s * nq + q] : i1[(size_t)s * nq + q];
                if (id < 0) continue;
                /* order by (key, global index) */
                int beats1 = x1 < 0 || better_f32(k, b1, metric) || (k == b1 && id < x1);
                if (beats1) {
                    b2 = b1; x2 = x1;
                    b1 = k;  x1 = id;
                } else {
                    int beats2 = x2 < 0 || better_f32(k, b2, metric) || (k == b2 && id < x2);
                    if (beats2) { b2 = k; x2 = id; }
                }
            }
        }
        ok1[q] = b1; ok2[q] = b2; oi1[q] = x1; oi2[q] = x2;
    }
}

/* ------------------------------------------------------------------------- */
/* formats                                                                   */
/* ------------------------------------------------------------------------- */

#define SIFT_NAME      ('S' + ('I' << 8) + ('F' << 16) + ('T' << 24))
#define MSER_NAME      ('M' + ('S' << 8) + ('E' << 16) + ('R' << 24))
#define SIFT_VERSION_4 ('V' + ('4' << 8) + ('.' << 16) + ('0' << 24))
#define SIFT_VERSION_5 ('V' + ('5' << 8) + ('.' << 16) + ('0' << 24))
#define SIFT_EOF       (0xff + ('E' << 8) + ('O' << 16) + ('F' << 24))

static int write_all(int fd, const void* p, size_t n)
{
    const char* c = (const char*)p;
    while (n) {
        ssize_t w = write(fd, c, n);
        if (w <= 0) return -1;
        c += w;
        n -= (size_t)w;
    }
    return 0;
}

static ssize_t read_all(int fd, void* p, size_t n)
{
    char* c = (char*)p;
    size_t got = 0;
    while (got < n) {
        ssize_t r = read(fd, c + got, n - got);
        if (r < 0) return -1;
        if (r == 0) break;
        got += (size_t)r;
    }
    return (ssize_t)got;
}

long orc_sift_write(const char* path, const float* loc, int nloc_dim,
                    const uint8_t* des, int npoint)
{
    /* FeaturePoints.cpp:228-280: header, loc block, descriptor block, EOF marker */
    int fd = open(path, O_CREAT | O_WRONLY | O_TRUNC, S_IRUSR | S_IWUSR);
    if (fd < 0) return -1;
    int32_t hdr[5] = {SIFT_NAME, SIFT_VERSION_4, npoint, nloc_dim, DESC_DIM};
    int32_t eof = (int32_t)SIFT_EOF;
    long total = 0;
    if (write_all(fd, hdr, sizeof hdr)) goto fail;
    total += sizeof hdr;
    if (write_all(fd, loc, sizeof(float) * (size_t)npoint * nloc_dim)) goto fail;
    total += (long)(sizeof(float) * (size_t)npoint * nloc_dim);
    if (write_all(fd, des, (size_t)npoint * DESC_DIM)) goto fail;
    total += (long)npoint * DESC_DIM;
    if (write_all(fd, &eof, sizeof eof)) goto fail;
    total += sizeof eof;
    close(fd);
    return total;
fail:
    close(fd);
    return -1;
}

int orc_sift_read(const char* path, float* loc, uint8_t* des, int nmax, int* nloc_dim)
{
    /* FeaturePoints.cpp:388-446 */
    int fd = open(path, O_RDONLY);
    if (fd < 0) return 0;
    int32_t hdr[5];
    if (read_all(fd, hdr, sizeof hdr) != (ssize_t)sizeof hdr) { close(fd); return 0; }
    int name_ok = hdr[0] == SIFT_NAME || hdr[0] == MSER_NAME;
    int ver_ok = hdr[1] == SIFT_VERSION_4 || hdr[1] == SIFT_VERSION_5;
    if (!name_ok || !ver_ok || !(hdr[2] > 0 && hdr[3] > 0 && hdr[4] == DESC_DIM)) { close(fd); return 0; }
    int np = hdr[2];
    if (nloc_dim) *nloc_dim = hdr[3];
    if (loc || des) {
        int take = np < nmax ? np : nmax;
        size_t locbytes = sizeof(float) * (size_t)np * hdr[3];
        if (loc) {
            float* tmp = (float*)malloc(locbytes);
            if (read_all(fd, tmp, locbytes) != (ssize_t)locbytes) { free(tmp); close(fd); return 0; }
            memcpy(loc, tmp, sizeof(float) * (size_t)take * hdr[3]);
            free(tmp);
        } else {
            lseek(fd, (off_t)locbytes, SEEK_CUR);
        }
        if (des && read_all(fd, des, (size_t)take * DESC_DIM) != (ssize_t)take * DESC_DIM) { close(fd); return 0; }
    }
    close(fd);
    return np;
}

int orc_sift_read_ascii(const char* path, float* loc, uint8_t* des, int nmax)
{
    /* FeaturePoints.cpp:54-125 (ReadSIFTA, ReadFeatureDataA with locDim 4): the second number of a feature
     * is x, the first y, both + 0.5; then z = 0, scale, the orientation negated */
    FILE* fp = fopen(path, "r");
    if (!fp) return 0;
    int npt = 0, nd = 0;
    if (fscanf(fp, "%d %d", &npt, &nd) != 2 || npt <= 0 || nd != DESC_DIM) { fclose(fp); return 0; }
    for (int i = 0; i < npt && i < nmax; ++i) {
        float* l = loc + (size_t)i * 5;
        if (fscanf(fp, "%f", &l[1]) != 1 || fscanf(fp, "%f", &l[0]) != 1) { fclose(fp); return 0; }
        l[0] += 0.5f;
        l[1] += 0.5f;
        l[2] = 0;
        if (fscanf(fp, "%f", &l[3]) != 1 || fscanf(fp, "%f", &l[4]) != 1) { fclose(fp); return 0; }
        l[4] = -l[4];
        for (int j = 0; j < nd; ++j) {
            unsigned int value;
            if (fscanf(fp, "%u", &value) != 1) { fclose(fp); return 0; }
            des[(size_t)i * DESC_DIM + j] = (uint8_t)value;
        }
    }
    fclose(fp);
    return npt;
}

long orc_sift_write_ascii(const char* path, const float* loc, const uint8_t* des, int n)
{
    /* FeaturePoints.cpp:205-226 (saveSIFTA); `ofstream << float` prints like %g */
    if (n <= 0) return 0;
    FILE* fp = fopen(path, "w");
    if (!fp) return -1;
    fprintf(fp, "%d %d\n", n, DESC_DIM);
    for (int i = 0; i < n; ++i) {
        const float* l = loc + (size_t)i * 5;
        fprintf(fp, "%g %g %g %g\n", l[1], l[0], l[3], l[4]);
        for (int k = 0; k < DESC_DIM; ++k) {
            fprintf(fp, "%u ", (unsigned)des[(size_t)i * DESC_DIM + k]);
            if ((k + 1) % 20 == 0) fprintf(fp, "\n");
        }
        fprintf(fp, "\n");
    }
    long total = ftell(fp);
    fclose(fp);
    return total;
}

void orc_sift_sort(float* loc, uint8_t* des, int nf, int32_t* order)
{
    /* FeaturePoints.cpp:726-765 (SortSIFT): bubble sort of the reversed sequence by scale, strict compare;
     * then Points::reorder (points.h:413-436): new row i = old row order[i] */
    if (nf == 0) return;
    float* size = (float*)malloc(sizeof(float) * (size_t)nf);
    for (int i = 0; i < nf; ++i) order[i] = nf - i - 1;
    for (int i = 0; i < nf; ++i) size[i] = loc[5 * (nf - i - 1) + 3];
    for (int i = 0; i < nf; ++i) {
        int swapped = 0;
        for (int j = nf - 1; j > i; --j) {
            if (size[j] > size[j - 1]) {
                float dt = size[j]; size[j] = size[j - 1]; size[j - 1] = dt;
                int it = order[j]; order[j] = order[j - 1]; order[j - 1] = it;
                swapped = 1;
            }
        }
        if (!swapped) break;
    }
    float* l2 = (float*)malloc(sizeof(float) * 5 * (size_t)nf);
    uint8_t* d2 = (uint8_t*)malloc((size_t)DESC_DIM * nf);
    for (int i = 0; i < nf; ++i) {
        memcpy(l2 + 5 * (size_t)i, loc + 5 * (size_t)order[i], sizeof(float) * 5);
        memcpy(d2 + (size_t)DESC_DIM * i, des + (size_t)DESC_DIM * order[i], DESC_DIM);
    }
    memcpy(loc, l2, sizeof(float) * 5 * (size_t)nf);
    memcpy(des, d2, (size_t)DESC_DIM * nf);
    free(size); free(l2); free(d2);
}

int orc_sift_convert_a2b(float* loc, uint8_t* des, int n)
{
    /* FeaturePoints.cpp:655-695 (ConvertA2B before it saves).  An all-zero row would divide by zero there;
     * it is left alone here. */
    double normal = 0;
    for (int i = 0; i < DESC_DIM; ++i) normal += des[i] * des[i];
    normal = sqrt(normal);
    if (!(normal < 400)) return 0;
    for (int i = 0; i < n; ++i) {
        uint8_t* d = des + (size_t)DESC_DIM * i;
        normal = 0;
        for (int j = 0; j < DESC_DIM; ++j) normal += d[j] * d[j];
        if (normal == 0) continue;
        normal = 512.0 / sqrt(normal);
        for (int k = 0; k < DESC_DIM; ++k) d[k] = (uint8_t)(unsigned)floor(normal * d[k] + 0.5);
    }
    for (int j = 0; j < n; ++j) {
        float* l = loc + 5 * (size_t)j;
        float t = l[0]; l[0] = l[1]; l[1] = t;
    }
    return 1;
}

long orc_matchtxt_record(const char* name0, const char* name1, const int32_t* matches0, int n,
                         char* buf, size_t bufsz)
{
    /* main.py:114-130: a ' ' follows every entry except the one for keypoint n-1 */
    size_t cap0 = 16 * (size_t)(n > 0 ? n : 1), l0 = 0, l1 = 0;
    char* s0 = (char*)malloc(cap0);
    char* s1 = (char*)malloc(cap0);
    int nm = 0;
    for (int i = 0; i < n; ++i) {
        if (matches0[i] == -1) continue;
        ++nm;
        const char* sep = (i == n - 1) ? "" : " ";
        l0 += (size_t)sprintf(s0 + l0, "%d%s", i, sep);
        l1 += (size_t)sprintf(s1 + l1, "%d%s", (int)matches0[i], sep);
    }
    s0[l0] = 0;
    s1[l1] = 0;
    /* main.py:143-155 */
    long len = snprintf(buf, bufsz, "%s %s %d\n%s\n%s\n", name0, name1, nm, s0, s1);
    free(s0);
    free(s1);
    return len;
}

int orc_image_pairs(int n, int consecutive, const int32_t* discrete, int ndiscrete,
                    int32_t* pairs, int cap)
{
    /* main.py:157-169 */
    int c = 0;
    for (int i = 0; i < n; ++i) {
        for (int ii = 1; ii <= consecutive; ++ii)
            if (i + ii < n) {
                if (c < cap) { pairs[2 * c] = i; pairs[2 * c + 1] = i + ii; }
                ++c;
            }
        for (int d = 0; d < ndiscrete; ++d)
            if (i + discrete[d] < n) {
                if (c < cap) { pairs[2 * c] = i; pairs[2 * c + 1] = i + discrete[d]; }
                ++c;
            }
    }
    return c;
}

/* ---- .mat ---------------------------------------------------------------- */

#define MT_VERSION_3 ('M' + ('T' << 8) + ('0' << 16) + ('3' << 24))
#define MR_VERSION_3 ('M' + ('R' << 8) + ('V' << 16) + ('3' << 24))
#define MT_MAX_PATH 260

typedef struct { int32_t version, file_count, definition_size, definition_buf, feature_count; } mt_header;
typedef struct { int32_t fcount, read_loc, block_size, trash_size, extra_size; char file_name[4]; } mt_recloc;
typedef struct {
    int32_t version;
    int32_t NF, NE, NH, NH2;
    float F[3][3], R[3][3], T[3], F1, F2, H[3][3], GE, AA;
    int32_t extra[7];
} mt_record;

/* MatchFile.cpp:372-406 */
static void mt_relative_path(const char* path1, const char* path2, char* rel)
{
    int i = 0;
    while (path2[i] == path1[i] && path2[i] && path1[i]) i++;
    while (i > 0 && path1[i] != '/') i--;
    if (i == 0) {
        strcpy(rel, path2);
        return;
    }
    rel[0] = 0;
    const char* p = path1 + i + 1;
    char* rp = rel;
    int rc1 = 0, rc2 = 0;
    while (*p) {
        if (p[0] == '/' && p[1] != '/') { strcpy(rp, "../"); rp += 3; rc1++; }
        p++;
    }
    p = path2 + i + 1;
    while (*p) {
        if (p[0] == '/' && p[1] != '/') rc2++;
        p++;
    }
    if (rc1 <= 1 || rc2 <= 1) strcpy(rp, path2 + i + 1);
    else strcpy(rel, path2);
}

typedef struct {
    int fd;
    mt_header h;
    char* def;          /* definition buffer, definition_buf bytes */
    int valid;
} mt_file;

static int mt_open(mt_file* f, const char* image, int write_mode)
{
    char path[MT_MAX_PATH + 8];
    snprintf(path, sizeof path, "%s.mat", image);
    memset(f, 0, sizeof *f);
    f->fd = write_mode ? open(path, O_CREAT | O_RDWR, S_IRUSR | S_IWUSR) : open(path, O_RDONLY);
    if (f->fd < 0) return 0;
    ssize_t r = read_all(f->fd, &f->h, sizeof f->h);
    if (r == (ssize_t)sizeof f->h && f->h.definition_size > 0 &&
        f->h.definition_buf >= f->h.definition_size && f->h.version == MT_VERSION_3) {
        f->def = (char*)calloc((size_t)f->h.definition_buf, 1);
        if (read_all(f->fd, f->def, (size_t)f->h.definition_size) != f->h.definition_size) return 0;
        f->valid = 1;
    } else {
        memset(&f->h, 0, sizeof f->h);
        f->h.version = MT_VERSION_3;
    }
    return 1;
}

static void mt_close(mt_file* f)
{
    if (f->fd >= 0) close(f->fd);
    free(f->def);
    f->def = NULL;
}

/* walks the record directory (MatchFile.cpp:275-299); returns the record with the
 * given relative name, and the last record through *last */
static mt_recloc* mt_find(mt_file* f, const char* rel, mt_recloc** last)
{
    mt_recloc* hit = NULL;
    char* p = f->def;
    if (last) *last = NULL;
    for (int i = 0; i < f->h.file_count; ++i) {
        mt_recloc* rp = (mt_recloc*)p;
        if (rp->read_loc <= 0 || rp->block_size == 0 || rp->extra_size > MT_MAX_PATH ||
            rp->fcount > 1000000 || rp->fcount < 0)
            break;
        if (!hit && strcmp(rp->file_name, rel) == 0) hit = rp;
        if (last) *last = rp;
        p += sizeof(mt_recloc) + rp->extra_size;
    }
    return hit;
}

/* offsets of the directory records that pass the reader's plausibility checks (MatchFile.cpp:275-299) */
static int mt_list(mt_file* f, int* off, int cap)
{
    int n = 0;
    char* p = f->def;
    for (int i = 0; i < f->h.file_count && n < cap; ++i) {
        mt_recloc* rp = (mt_recloc*)p;
        if (rp->read_loc <= 0 || rp->block_size == 0 || rp->extra_size > MT_MAX_PATH ||
            rp->fcount > 1000000 || rp->fcount < 0)
            break;
        off[n++] = (int)(p - f->def);
        p += sizeof(mt_recloc) + rp->extra_size;
    }
    return n;
}

#define MT_MAX_RECORDS 65536

/* MatchFile::MoveRecordToEnd (MatchFile.cpp:422-453).  Returns the moved record at its new place. */
static mt_recloc* mt_move_record_to_end(mt_file* f, int i)
{
    int* off = (int*)malloc(sizeof(int) * MT_MAX_RECORDS);
    int n = mt_list(f, off, MT_MAX_RECORDS);
    if (i >= n - 1 || i < 0 || n < 2) {
        free(off);
        return NULL;
    }
    int loc1 = off[i], loc2 = off[i + 1], len = loc2 - loc1;                            /* :428-430 */
    char* temp = (char*)malloc((size_t)len);
    memcpy(temp, f->def + loc1, (size_t)len);                                           /* :432 */
    memmove(f->def + loc1, f->def + loc2, (size_t)(f->h.definition_size - loc2));       /* :433 */
    memcpy(f->def + f->h.definition_size - len, temp, (size_t)len);                     /* :435 */
    free(temp);
    n = mt_list(f, off, MT_MAX_RECORDS);                                                /* :437 */
    if (i == 0) {
        f->h.definition_buf += len;                                                     /* :441 */
        f->def = (char*)realloc(f->def, (size_t)f->h.definition_buf);
        memset(f->def + f->h.definition_buf - len, 0, (size_t)len);
    } else {
        ((mt_recloc*)(f->def + off[i - 1]))->trash_size += len;                         /* :445 */
    }
    mt_recloc* last = (mt_recloc*)(f->def + off[n - 1]);
    mt_recloc* prev = (mt_recloc*)(f->def + off[n - 2]);
    last->read_loc = prev->read_loc + prev->block_size;                                 /* :449-450 */
    free(off);
    return last;
}

/* The full-directory branch of MatchFile::AddImageMatch (MatchFile.cpp:494-560): the data blocks that sit
 * where the doubled directory will be move to the end of the file, their directory records rotate to the
 * end of the directory.  Returns the new record (name, sizes not yet set beyond what :542-548 sets). */
static mt_recloc* mt_grow_and_add(mt_file* f, const char* rel, int elen, int rlen)
{
    int* off = (int*)malloc(sizeof(int) * MT_MAX_RECORDS);
    int n = mt_list(f, off, MT_MAX_RECORDS);
    int extra_needed = rlen + f->h.definition_buf;                                      /* :497 */
    int move = 0, extra_space = 0;
    while (extra_space < extra_needed && move < n) {                                    /* :502-507 */
        mt_recloc* loc = (mt_recloc*)(f->def + off[move]);
        extra_space += loc->block_size + loc->trash_size;
        move++;
    }
    if (extra_space < extra_needed || n == 0) {   /* the reference would run off its record list here */
        free(off);
        return NULL;
    }
    char* buf = (char*)malloc((size_t)extra_space);
    memset(buf, 0, (size_t)extra_space);
    mt_recloc* first = (mt_recloc*)(f->def + off[0]);
    lseek(f->fd, first->read_loc, SEEK_SET);                                            /* :513-514 */
    read_all(f->fd, buf, (size_t)extra_space);
    mt_recloc* back = (mt_recloc*)(f->def + off[n - 1]);
    int wloc = back->read_loc + back->block_size;                                       /* :517 */
    char* pbuf = buf;
    for (int i = 0; i < move; ++i) {                                                    /* :519-527 */
        mt_recloc* loc = (mt_recloc*)(f->def + off[i]);
        lseek(f->fd, wloc, SEEK_SET);
        write_all(f->fd, pbuf, (size_t)loc->block_size);
        pbuf += loc->block_size + loc->trash_size;
        loc->read_loc = wloc;
        wloc += loc->block_size;
        loc->trash_size = 0;
    }
    free(buf);
    if (move < n) {                                                                     /* :530-537 */
        int definition_move = off[move];
        char* temp = (char*)malloc((size_t)f->h.definition_size);
        memcpy(temp, f->def, (size_t)f->h.definition_size);
        memcpy(f->def, temp + definition_move, (size_t)(f->h.definition_size - definition_move));
        memcpy(f->def + f->h.definition_size - definition_move, temp, (size_t)definition_move);
        free(temp);
    }
    int old_buf = f->h.definition_buf;
    f->h.definition_buf = f->h.definition_buf * 2 + rlen;                               /* :538-539 */
    f->def = (char*)realloc(f->def, (size_t)f->h.definition_buf);
    memset(f->def + old_buf, 0, (size_t)(f->h.definition_buf - old_buf));
    mt_recloc* rec = (mt_recloc*)(f->def + f->h.definition_size);                       /* :542-548 */
    rec->read_loc = wloc;
    rec->trash_size = 0;
    rec->block_size = 0;
    rec->extra_size = elen;
    strcpy(rec->file_name, rel);
    f->h.definition_size += rlen;                                                       /* :556-557 */
    f->h.file_count++;
    free(off);
    return rec;
}

static int mt_write_one(const char* image, const char* other, int fc_other, int nm,
                        const int32_t* row_this, const int32_t* row_other, int reservation)
{
    mt_file f;
    if (!mt_open(&f, image, 1)) return -1;
    char rel[MT_MAX_PATH];
    mt_relative_path(image, other, rel);
    int slen = ((int)strlen(rel) / 4 + 1) * 4, elen = slen - 4, rlen = elen + (int)sizeof(mt_recloc);
    mt_recloc* last = NULL;
    mt_recloc* loc = f.valid ? mt_find(&f, rel, &last) : NULL;
    int bs = (int)sizeof(mt_record) + (nm > 8 ? nm : 8) * (int)sizeof(int32_t) * 6;
    if (!loc) {
        if (f.h.file_count == 0) { /* MatchFile.cpp:465-482 */
            f.h.version = MT_VERSION_3;
            f.h.file_count = 1;
            f.h.definition_buf = (reservation > 20 ? reservation : 20) * (MT_MAX_PATH + (int)sizeof(mt_recloc));
            f.def = (char*)calloc((size_t)f.h.definition_buf, 1);
            f.h.definition_size = rlen;
            loc = (mt_recloc*)f.def;
            loc->read_loc = (int32_t)sizeof(mt_header) + f.h.definition_buf;
            strcpy(loc->file_name, rel);
            loc->extra_size = elen;
            loc->trash_size = 0;
        } else if (f.h.definition_size + rlen < f.h.definition_buf) { /* :483-493 */
            loc = (mt_recloc*)(f.def + f.h.definition_size);
            loc->read_loc = last->read_loc + last->block_size;
            f.h.definition_size += rlen;
            f.h.file_count++;
            strcpy(loc->file_name, rel);
            loc->extra_size = elen;
            loc->trash_size = 0;
        } else { /* :494-560 */
            loc = mt_grow_and_add(&f, rel, elen, rlen);
            if (!loc) { mt_close(&f); return -2; }
        }
        loc->fcount = fc_other;
        loc->block_size = bs; /* a new record is the last one: :781-786 */
    } else {
        loc->fcount = fc_other; /* :777 */
        if (loc == last) { /* :781-786 */
            loc->block_size = bs;
        } else {
            int bsmin = (int)sizeof(mt_record) + (nm > 0 ? nm : 0) * (int)sizeof(int32_t) * 4;
            if (bsmin > loc->block_size) {
                if (bsmin > loc->block_size + loc->trash_size) { /* :792-798 */
                    int* off = (int*)malloc(sizeof(int) * MT_MAX_RECORDS);
                    int n = mt_list(&f, off, MT_MAX_RECORDS), idx = -1;
                    for (int i = 0; i < n; ++i)
                        if ((mt_recloc*)(f.def + off[i]) == loc) idx = i;
                    free(off);
                    loc = mt_move_record_to_end(&f, idx);
                    if (!loc) { mt_close(&f); return -2; }
                    loc->block_size = bs;
                } else { /* :801-802 */
                    loc->block_size += loc->trash_size;
                    loc->trash_size = 0;
                }
            }
        }
    }
    /* :834-852 header + directory */
    lseek(f.fd, 0, SEEK_SET);
    write_all(f.fd, &f.h, sizeof f.h);
    write_all(f.fd, f.def, (size_t)f.h.definition_size);
    /* :813-831 record */
    mt_record rec;
    memset(&rec, 0, sizeof rec);
    rec.version = MR_VERSION_3;
    rec.GE = 100.0f;
    lseek(f.fd, loc->read_loc, SEEK_SET);
    int32_t nm32 = nm;
    write_all(f.fd, &nm32, sizeof nm32);
    write_all(f.fd, &rec, sizeof rec);
    if (nm > 0) {
        write_all(f.fd, row_this, sizeof(int32_t) * (size_t)nm);
        write_all(f.fd, row_other, sizeof(int32_t) * (size_t)nm);
    }
    mt_close(&f);
    return 0;
}

int orc_mat_write_pmatch(const char* image1, const char* image2, int fc1, int fc2,
                         int nm, const int32_t* idx1, const int32_t* idx2, int reservation)
{
    if (!image1 || !image2 || !image1[0] || !image2[0]) return -1;
    int r = mt_write_one(image1, image2, fc2, nm, idx1, idx2, reservation);
    if (r) return r;
    return mt_write_one(image2, image1, fc1, nm, idx2, idx1, reservation);
}

int orc_mat_get_pmatch(const char* image, const char* other, int fc_other,
                       int32_t* idx_this, int32_t* idx_other, int cap)
{
    mt_file f;
    if (!mt_open(&f, image, 0)) return 0;
    int nm = 0;
    if (f.valid) {
        char rel[MT_MAX_PATH];
        mt_relative_path(image, other, rel);
        mt_recloc* loc = mt_find(&f, rel, NULL);
        if (loc && loc->fcount == fc_other) {
            int32_t stored = 0;
            lseek(f.fd, loc->read_loc, SEEK_SET);
            if (read_all(f.fd, &stored, sizeof stored) == (ssize_t)sizeof stored &&
                stored > 0 && stored <= fc_other) {
                lseek(f.fd, (off_t)sizeof(mt_record), SEEK_CUR);
                nm = stored;
                int take = nm < cap ? nm : cap;
                if (idx_this) read_all(f.fd, idx_this, sizeof(int32_t) * (size_t)take);
                lseek(f.fd, loc->read_loc + 4 + (off_t)sizeof(mt_record) + (off_t)sizeof(int32_t) * nm, SEEK_SET);
                if (idx_other) read_all(f.fd, idx_other, sizeof(int32_t) * (size_t)take);
            }
        }
    }
    mt_close(&f);
    return nm;
}

/* MatchFile::WriteIMatch(image_match, reverse, NM = 0, tvg, inliers) (MatchFile.cpp:660-748) for the geometry
 * WriteFMatch builds (:596-605: TwoViewGeometry(NF, F), MatchFile.h:85-92), one file.  Returns 1, or 0 when the file
 * has no record of `other`.  The record on disk: version, the geometry as SetGeometry / SetGeometryR leave it
 * (MatchFile.h:121-160; F transposed in the partner's file), everything else zero (the reference leaves the unset
 * members of its stack object as they happen to be).
 * One deviation, in the re-allocation branch (:690-719) when the record moves to the end of the file: the reference
 * writes only the geometry and the inlier lists at the new place, so that NM and the putative lists stay behind and
 * its readers find NM = 0; here the count and the lists move with the record. */
static int mt_write_inliers(const char* image, const char* other, int reverse, int nf,
                            const int32_t* row_a, const int32_t* row_b, const float* F)
{
    mt_file f;
    if (!mt_open(&f, image, 1)) return -1;
    if (!f.valid) { mt_close(&f); return 0; }
    char rel[MT_MAX_PATH];
    mt_relative_path(image, other, rel);                                                /* :666-668 */
    mt_recloc* last = NULL;
    mt_recloc* loc = mt_find(&f, rel, &last);
    if (!loc) { mt_close(&f); return 0; }
    int32_t nm = 0;
    lseek(f.fd, loc->read_loc, SEEK_SET);                                               /* :676-680 */
    if (read_all(f.fd, &nm, sizeof nm) != (ssize_t)sizeof nm || nm < 0) { mt_close(&f); return 0; }
    const int szmin = (1 + (nm + nf) * 2) * (int)sizeof(int32_t) + (int)sizeof(mt_record);   /* :688 */
    if (szmin > loc->block_size) {                                                      /* :690 */
        const int bs = (int)sizeof(mt_record) + ((nm + nf) > 8 ? (nm + nf) : 8) * (int)sizeof(int32_t) * 4;
        if (loc->block_size == 0 || loc == last) {                                      /* :695-699 */
            loc->block_size = bs;
        } else if (szmin > loc->block_size + loc->trash_size) {                         /* :700-706 */
            int32_t* lists = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * nm + 1));
            lseek(f.fd, loc->read_loc + 4 + (off_t)sizeof(mt_record), SEEK_SET);
            read_all(f.fd, lists, sizeof(int32_t) * (size_t)(2 * nm));
            int* off = (int*)malloc(sizeof(int) * MT_MAX_RECORDS);
            int n = mt_list(&f, off, MT_MAX_RECORDS), idx = -1;
            for (int i = 0; i < n; ++i)
                if ((mt_recloc*)(f.def + off[i]) == loc) idx = i;
            free(off);
            loc = mt_move_record_to_end(&f, idx);
            if (!loc) { free(lists); mt_close(&f); return -2; }
            loc->block_size = bs;
            lseek(f.fd, loc->read_loc, SEEK_SET);                                       /* the deviation stated above */
            write_all(f.fd, &nm, sizeof nm);
            lseek(f.fd, (off_t)sizeof(mt_record), SEEK_CUR);
            write_all(f.fd, lists, sizeof(int32_t) * (size_t)(2 * nm));
            free(lists);
        } else {                                                                        /* :707-711 */
            loc->block_size += loc->trash_size;
            loc->trash_size = 0;
        }
        lseek(f.fd, 0, SEEK_SET);                                                       /* :713-718 */
        write_all(f.fd, &f.h, sizeof f.h);
        write_all(f.fd, f.def, (size_t)f.h.definition_size);
    }
    mt_record rec;
    memset(&rec, 0, sizeof rec);
    rec.version = MR_VERSION_3;
    rec.GE = 100.0f;
    rec.NF = nf;
    if (nf > 0)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) rec.F[i][j] = reverse ? F[3 * j + i] : F[3 * i + j];
    lseek(f.fd, loc->read_loc + (off_t)sizeof(int32_t), SEEK_SET);                      /* :683 / :717 */
    write_all(f.fd, &rec, sizeof rec);                                                  /* :727 / :739 */
    lseek(f.fd, (off_t)nm * 8, SEEK_CUR);                                               /* :728 / :741 */
    if (nf > 0) {                                                                       /* :729-733 / :742-746 */
        write_all(f.fd, reverse ? row_b : row_a, sizeof(int32_t) * (size_t)nf);
        write_all(f.fd, reverse ? row_a : row_b, sizeof(int32_t) * (size_t)nf);
    }
    mt_close(&f);
    return 1;
}

/* MatchFile::WriteFMatch (MatchFile.cpp:596-605) -> WriteIMatch (:643-658): the first image's file, then the second's */
int orc_mat_write_fmatch(const char* image1, const char* image2, int nf, const int32_t* idx1, const int32_t* idx2,
                         const float* F)
{
    if (!image1 || !image2 || !image1[0] || !image2[0]) return -1;
    const int a = mt_write_inliers(image1, image2, 0, nf, idx1, idx2, F);
    if (a < 0) return a;
    const int b = mt_write_inliers(image2, image1, 1, nf, idx1, idx2, F);
    if (b < 0) return b;
    return a == 1 && b == 1;
}
