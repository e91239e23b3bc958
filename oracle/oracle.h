/*
 * CPU oracle for the descriptor-matching hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the scalar, obviously-correct restatement that the HIP path is checked
 * against.  It is never linked into, loaded by, or called from the product library;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * What the reference pins and what it does not:
 *   - file layouts (.sift, match txt, .mat PMatch records) are restated from
 *     Others/visualSfM_Sift_file/Match_modified/{FeaturePoints,MatchFile}.{h,cpp}
 *     and Others/SuperPoint_data/main.py.  The C++ reference classes need Windows
 *     <io.h>, a missing stdafx.h and a missing NamedMutex.h, so they are not built
 *     here; the layouts are pinned by byte sizes observed when the reference was
 *     probed (3-feature .sift = 468 bytes, first one-record .mat = 5912 bytes).
 *   - mutual-nearest-neighbour semantics, "-1" sentinels and first-index tie-break
 *     follow Others/SuperPoint_data/models/superglue.py:276-286.
 *   - brute-force L2 1-NN follows Graph_Idea/model/model_2_0.py:43-78.
 *   - the 2-NN + Lowe-ratio matcher itself has NO counterpart in the reference
 *     ("parity unpinned"): this file is its normative definition.
 */
#ifndef VSM_ORACLE_H
#define VSM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NO_SECOND INT32_MAX /* d2 when the train set has fewer than 2 descriptors */

typedef struct orc_params {
    int32_t ratio_num;  /* accept iff d1 * den^2 < num^2 * d2 (squared distances); num<=0 disables */
    int32_t ratio_den;
    int32_t mutual;     /* 1: keep (i, j) only if i is also the nearest query of j */
    int32_t reserved;
    int64_t max_dist_sq; /* u8: accept iff d1 <= max_dist_sq; <0 disables */
    float   min_score;   /* f32 dot metric: accept iff best score > min_score */
    float   max_dist_sq_f; /* f32 L2 metric: accept iff d1 <= max_dist_sq_f; <0 disables */
} orc_params;

enum { ORC_METRIC_DOT = 0, ORC_METRIC_L2 = 1 };
enum { ORC_LAYOUT_ROWS = 0 /* [n][dim] */, ORC_LAYOUT_DIM_MAJOR = 1 /* [dim][n] */ };

/* ---- SIFT-128 u8, exact int32 squared L2 ---------------------------------- */
/* For every query row i of A[n][128]: best train row of B[m][128] (lowest index on
 * ties), its squared distance, and the smallest squared distance over j != best. */
void orc_knn2_u8(const uint8_t* A, int n, const uint8_t* B, int m, int threads,
                 int32_t* idx1, int32_t* d1, int32_t* d2);
/* For every train row j: nearest query row (lowest index on ties) and distance. */
void orc_colbest_u8(const uint8_t* A, int n, const uint8_t* B, int m, int threads,
                    int32_t* cidx, int32_t* cd);
/* Full matcher: 2-NN, ratio, distance cap, mutual check; pairs ascending in idx0.
 * Returns the number of matches (<= cap written). */
int orc_match_u8(const uint8_t* A, int n, const uint8_t* B, int m, const orc_params* p,
                 int threads, int32_t* idx0, int32_t* idx1, int32_t* dist, int cap);
/* Same decision, dense form with -1 sentinels (superglue.py:285-290 convention). */
void orc_match_dense_u8(const uint8_t* A, int n, const uint8_t* B, int m, const orc_params* p,
                        int threads, int32_t* matches0, int32_t* matches1);

/* ---- float32 descriptors ---------------------------------------------------- */
/* dot(a, b) as the k-ascending single-precision fused-multiply-add chain
 * s = fmaf(a[k], b[k], s), s0 = 0. */
float orc_dot_f32(const float* a, const float* b, int dim);
/* key(i, j): DOT -> dot(a_i, b_j) (larger is better);
 *            L2  -> fmaf(-2, dot(a_i, b_j), |b_j|^2) (smaller is better), and the
 *                   reported distance is max(key + |a_i|^2, 0). */
void orc_knn2_f32(const float* A, int n, const float* B, int m, int dim, int metric,
                  int threads, int32_t* idx1, float* s1, float* s2);
void orc_colbest_f32(const float* A, int n, const float* B, int m, int dim, int metric,
                     int threads, int32_t* cidx, float* cs);
int orc_match_f32(const float* A, int n, const float* B, int m, int dim, int metric,
                  const orc_params* p, int threads,
                  int32_t* idx0, int32_t* idx1, float* score, int cap);
void orc_match_dense_f32(const float* A, int n, const float* B, int m, int dim, int metric,
                         const orc_params* p, int threads,
                         int32_t* matches0, int32_t* matches1, float* best0, float* best1);
/* model_2_0.py:43-78: x is [batch][dim][npts], emb is [dim][k]; argmin[b][p]. */
void orc_nearest_embed_f32(const float* x, int batch, int dim, int npts,
                           const float* emb, int k, int threads,
                           int64_t* argmin, float* result /* [batch][dim][npts] or NULL */);
/* Merge of per-slice top-2 lists (database sliced by contiguous index ranges):
 * in[s][q] = {key1, key2, idx1(global), idx2(global)}; smaller key wins for L2,
 * larger for DOT; lower global index wins ties. */
void orc_merge_top2_f32(const float* k1, const float* k2, const int32_t* i1, const int32_t* i2,
                        int nslices, int nq, int metric,
                        float* ok1, float* ok2, int32_t* oi1, int32_t* oi2);

/* ---- image-pair shortlist (shortlist_oracle.c) ------------------------------- */
long long orc_shortlist(const int32_t* words, const int64_t* offsets, int nimg, int nwords, int topk,
                        int32_t* pairs, long long cap, int32_t* scores);

int orc_train_vocabulary(const uint8_t* rows, long long total, int nwords, int iterations, int threads,
                         uint8_t* words);

/* ---- formats ---------------------------------------------------------------- */
/* FeaturePoints.cpp:228-280 / :388-446.  Returns bytes written or <0. */
long orc_sift_write(const char* path, const float* loc, int nloc_dim,
                    const uint8_t* des, int npoint);
/* Returns npoint (>0) on success, 0 on failure, like the reference reader.
 * Pass NULL buffers to query sizes only. */
int orc_sift_read(const char* path, float* loc, uint8_t* des, int nmax, int* nloc_dim);
/* main.py:114-130 + :143-155: formats one record into buf, returns its length.
 * matches0[n] holds the image-1 index or -1. */
long orc_matchtxt_record(const char* name0, const char* name1, const int32_t* matches0, int n,
                         char* buf, size_t bufsz);
/* main.py:157-169 pair list over n images -> pairs[][2]; returns the count. */
int orc_image_pairs(int n, int consecutive, const int32_t* discrete, int ndiscrete,
                    int32_t* pairs, int cap);
/* MatchFile.cpp:751-832: appends one putative-match record to <image1>.mat and
 * <image2>.mat (paths without extension).  Only the no-regrow case of
 * AddImageMatch is restated (returns -2 when the directory would have to grow). */
int orc_mat_write_pmatch(const char* image1, const char* image2, int fc1, int fc2,
                         int nm, const int32_t* idx1, const int32_t* idx2, int reservation);
/* MatchFile.cpp:856-895: returns NM (0 when absent or rejected). */
/* MatchFile::WriteFMatch (MatchFile.cpp:596-605, 643-748), the record re-allocation of :690-719 included */
int orc_mat_write_fmatch(const char* image1, const char* image2, int nf, const int32_t* idx1, const int32_t* idx2,
                         const float* F);
int orc_mat_get_pmatch(const char* image, const char* other, int fc_other,
                       int32_t* idx_this, int32_t* idx_other, int cap);

#ifdef __cplusplus
}
#endif
#endif
