/*
 * vsm — MI355X descriptor-matching engine, C ABI.
 *
 * The reference (thuanaislab/visual_slam) has no plugin or FFI interface on this
 * path; its boundary is a set of file formats plus plain C++ calls that take
 * caller-owned raw buffers.  Each entry point below names the reference code it
 * stands in for.  Conventions follow the reference's own:
 *   - caller allocates and frees every host buffer (FeaturePoints.cpp:516-531);
 *   - readers return 1/0 or a count >= 0 (FeaturePoints.cpp:392,440; MatchFile.cpp:886);
 *     everything else returns a count >= 0 or a negative VSM_E* code; nothing throws;
 *   - a match list is two equal-length int32 arrays, image-0 index ascending and the
 *     matching image-1 index, i.e. the two rows of Points<int>(NM, 2)
 *     (MatchFile.h:55-65) and the two index lines of the match txt (main.py:114-155);
 *   - dense results use -1 for "no match" (superglue.py:285-290);
 *   - a context is single-threaded; distinct contexts may be used concurrently.
 *
 * All compute entry points require the HIP device the context was created on; there is
 * no CPU fallback.
 */
#ifndef VSM_H
#define VSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSM_DESC_DIM_U8 128

enum {
    VSM_OK = 0,
    VSM_EINVAL = -1,   /* bad argument */
    VSM_ENODEV = -2,   /* no usable HIP device / HIP call failed */
    VSM_ENOMEM = -3,   /* device or host allocation failed */
    VSM_ECAP = -4,     /* caller buffer too small */
    VSM_EIO = -5,      /* file error */
    VSM_EFORMAT = -6,  /* malformed file */
    VSM_EUNSUPPORTED = -7
};

enum { VSM_METRIC_DOT = 0, VSM_METRIC_L2 = 1 };
enum { VSM_LAYOUT_ROWS = 0 /* [n][dim] */, VSM_LAYOUT_DIM_MAJOR = 1 /* [dim][n] */ };
enum { VSM_NO_SECOND = INT32_MAX };

typedef struct vsm_ctx vsm_ctx;

typedef struct vsm_params {
    int32_t ratio_num;     /* Lowe test on squared distances: accept iff d1*den^2 < num^2*d2; */
    int32_t ratio_den;     /* num <= 0 disables it.  0 < num, den <= 16384.                    */
    int32_t mutual;        /* keep (i, j) only if i is also the nearest image-0 row of j        */
    int32_t reserved;
    int64_t max_dist_sq;   /* u8 sets: accept iff d1 <= max_dist_sq; < 0 disables               */
    float   min_score;     /* f32 DOT: accept iff best score > min_score                        */
    float   max_dist_sq_f; /* f32 L2: accept iff d1 <= max_dist_sq_f; < 0 disables              */
} vsm_params;

/* ---- context -------------------------------------------------------------- */
vsm_ctx* vsm_create(int device);
void vsm_destroy(vsm_ctx* ctx);
/* Message for the last failure on this context (or the last failed vsm_create
 * when ctx is NULL).  Counterpart of the reference's redirectable printf
 * (MatchFile.cpp:44-45, 62-65). */
const char* vsm_last_error(const vsm_ctx* ctx);
/* Blocks until all work queued by this context has finished. */
int vsm_synchronize(vsm_ctx* ctx);
/* Native stream handle (hipStream_t) the context launches on. */
void* vsm_stream(vsm_ctx* ctx);

/* ---- descriptor sets -------------------------------------------------------- */
/* SIFT descriptors as FeatureData::ReadSIFTB(file, locbuf, desbuf, nmax) hands them
 * over (FeaturePoints.cpp:516-531): row-major uint8 [n][128].  Returns a set id. */
int vsm_upload_u8(vsm_ctx* ctx, const uint8_t* des, int n);
/* Same, from a device pointer on the context's device (no host round trip).  The buffer must be
 * 16-byte aligned (VSM_EINVAL otherwise); it may be reused as soon as the call returns. */
int vsm_upload_u8_device(vsm_ctx* ctx, const void* des_dev, int n);
/* SuperPoint-style float descriptors.  VSM_LAYOUT_DIM_MAJOR is the reference's
 * native [256][N] tensor (superpoint.py:196-202), VSM_LAYOUT_ROWS the transposed
 * form main.py:255 makes.  1 <= dim <= 256. */
int vsm_upload_f32(vsm_ctx* ctx, const float* des, int n, int dim, int layout);
/* From a device pointer on the context's device, either layout; 4-byte aligned. */
int vsm_upload_f32_device(vsm_ctx* ctx, const void* des_dev, int n, int dim, int layout);
/* How float sets are searched.  Both modes return the same bits: scores are the
 * k-ascending single-precision fma chain, ties go to the lowest index (the torch
 * max/argmin convention of superglue.py:276-279 and model_2_0.py:64).
 *   VSM_F32_FAST_EXACT (default): bf16 matrix-core pass that keeps a proven superset of
 *     each query's two best rows, then an exact re-rank of the survivors; falls back to
 *     the f32 kernel by itself when a survivor list overflows (massively tied scores).
 *   VSM_F32_MFMA_F32: every score on the f32-input matrix cores. */
enum { VSM_F32_FAST_EXACT = 0, VSM_F32_MFMA_F32 = 1 };
int vsm_set_f32_mode(vsm_ctx* ctx, int mode);
/* Which of the two ran in the last float search on this context (-1: none yet). */
int vsm_last_f32_path(const vsm_ctx* ctx);
/* Mutual check of float pair submissions (fast mode).  The reverse searches of a submission can run after the
 * forward ones, over only the train rows that candidates which passed the score / distance / ratio tests name
 * (VSM_F32_REVERSE_CANDIDATES; on the dot-product ranking they also look at the candidates' own rows of the
 * first image only, since no other row can outscore a candidate in its column), or together with them over
 * every train row (VSM_F32_REVERSE_ALL).  Same lists
 * either way.  VSM_F32_REVERSE_AUTO (default) takes the first for large submissions whose rule has such a
 * test and the second otherwise (one launch less per stage matters for a single small pair). */
enum { VSM_F32_REVERSE_AUTO = 0, VSM_F32_REVERSE_CANDIDATES = 1, VSM_F32_REVERSE_ALL = 2 };
int vsm_set_f32_reverse(vsm_ctx* ctx, int mode);
/* Which of the two the last float pair submission on this context took (VSM_F32_REVERSE_CANDIDATES or
 * VSM_F32_REVERSE_ALL; -1: none yet, or no mutual check). */
int vsm_last_f32_reverse(const vsm_ctx* ctx);
/* How u8 (SIFT) sets are searched.  Both forms return the same bits (exact integer distances, lowest
 * index on ties).
 *   VSM_U8_AUTO (default): pair submissions run the tile-minimum query pass (the matrix cores produce
 *     half-distances directly, each lane keeps the minimum per 32-row tile) and settle row, exact distance and tests
 *     for the candidates afterwards, from the 8-row group that holds the minimum (every row is a candidate when the
 *     rule has neither a ratio test nor a distance cap); rows that group cannot settle (ties between groups) take a
 *     rescan pass, see vsm_last_u8_rescanned.  With a ratio test or cap the mutual check looks only at the image-0
 *     rows near enough to anything to matter.  The raw 2-NN call (vsm_knn2) runs the packed form: the tile-minimum
 *     pass knows the runner-up distance only up to ties.
 *     In a submission of eight pairs or more whose sets all have at most 8192 rows, the settling of a strip of 512
 *     query rows runs in the tail of the query kernel itself, while the train image is still in the cache.
 *     A call with ONE pair of modest size (rows x rows up to 2^29) runs the packed form: such a call is bound by its
 *     launches, of which that form has fewer.
 *   VSM_U8_PACKED: the packed form only (value and row in one key for every distance).
 *   VSM_U8_TILE_SEPARATE / VSM_U8_TILE_FUSED: as VSM_U8_AUTO, with the settling always in passes of its own /
 *     in the query kernel's tail whenever the sets' sizes allow it, whatever the number of pairs, and without
 *     VSM_U8_AUTO's change to the packed form on data full of duplicates.
 * Setting a mode also makes VSM_U8_AUTO forget what it learnt from earlier submissions. */
enum { VSM_U8_AUTO = 0, VSM_U8_PACKED = 1, VSM_U8_TILE_SEPARATE = 2, VSM_U8_TILE_FUSED = 3 };
int vsm_set_u8_mode(vsm_ctx* ctx, int mode);
/* Which query pass the last u8 submission ran: 0 = tile-minimum, 1 = packed, 2 = tile-minimum with the candidates
 * settled in its tail (-1: none yet). */
int vsm_last_u8_path(const vsm_ctx* ctx);
/* Rows of the last finished u8 submission (collected, or matched by a call that waits) that the tile-minimum form
 * could not settle from its query pass and sent through the rescan pass: a second, packed-form pass over the whole
 * train set for those rows only.  0 for the packed form; -1: none yet.  A handful per pair on ordinary data; every row
 * on data full of exact duplicates (e.g. the all-ones descriptors of FeaturePoints.cpp:358-367), which is the worst
 * case of that form: the two passes together.  Under VSM_U8_AUTO a submission in which more than one row in eight
 * went that way makes the next 31 submissions of the context run the packed form, after which the tile-minimum form
 * is tried again. */
int vsm_last_u8_rescanned(const vsm_ctx* ctx);
int vsm_release_set(vsm_ctx* ctx, int set);
int vsm_set_count(const vsm_ctx* ctx, int set);

/* ---- matching ----------------------------------------------------------------- */
/* One image pair: 2-NN of every set0 row in set1, ratio / distance / mutual tests,
 * matches ascending in idx0.  The position the reference fills with
 * matching(...) .. get_list_matching_pairs (main.py:241-273).  dist may be NULL
 * (int32 squared L2 for u8 sets, float score/squared distance for f32 sets).
 * Returns the match count, or VSM_ECAP if it exceeds cap. */
int vsm_match_pair(vsm_ctx* ctx, int set0, int set1, const vsm_params* p, int metric,
                   int32_t* idx0, int32_t* idx1, void* dist, int cap);
/* Dense form: matches0[n0], matches1[n1] with -1 (superglue.py:276-293). */
int vsm_match_pair_dense(vsm_ctx* ctx, int set0, int set1, const vsm_params* p, int metric,
                         int32_t* matches0, int32_t* matches1);
/* No counterpart in the reference.  vsm_match_pair for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): one pair of
 * resident u8 sets, same match list, but every intermediate lives in a fixed 4 MiB array that the
 * library holds once per GPU, the call makes no hipMalloc / hipFree / host-to-device copy, and the
 * results go straight to the caller's DEVICE buffers: idx0_dev / idx1_dev (cap >= rows of set0 entries
 * each; the first *count_dev are valid) and count_dev (one int32).  The work is queued on the context's
 * stream (vsm_stream) and the call returns without waiting; VSM_ECAP when a set has more than 16384
 * rows.  Every call on a GPU shares that GPU's array: the library orders them behind one another with an
 * event per device (a call waits on the device for the one queued before it, whatever context or stream
 * that came from), so calls from several contexts do not corrupt each other, and do not overlap either. */
int vsm_match_pair_noalloc(vsm_ctx* ctx, int set0, int set1, const vsm_params* p,
                           int32_t* idx0_dev, int32_t* idx1_dev, int32_t* count_dev, int cap);
/* Many pairs in one submission (the per-pair loop of main.py:220-280).  pairs is
 * [npairs][2] set ids.  offsets[npairs+1] receives where each pair's matches start
 * in idx0/idx1 (cap entries each).  Returns the total or VSM_ECAP. */
long long vsm_match_pairs(vsm_ctx* ctx, const int32_t* pairs, int npairs, const vsm_params* p,
                          int metric, int64_t* offsets, int32_t* idx0, int32_t* idx1,
                          long long cap);
/* The two halves of vsm_match_pairs for callers that keep the GPU fed across calls: submit queues one
 * submission (it must fit the scratch budget) and returns at once; collect waits for the OLDEST
 * submission in flight and copies its lists (offsets[npairs+1] relative to that submission).  At most
 * two submissions are in flight; the pattern is submit(0), submit(1), collect(0), submit(2), ...
 * Float pairs are searched inside submit and cannot be queued behind another submission. */
int vsm_submit_pairs(vsm_ctx* ctx, const int32_t* pairs, int npairs, const vsm_params* p, int metric);
long long vsm_collect_matches(vsm_ctx* ctx, int64_t* offsets, int32_t* idx0, int32_t* idx1, long long cap);
/* Device scratch vsm_match_pairs may use per submission (default 8 GiB, at least 1 MiB): it cuts
 * the pair list into submissions of about half that and keeps two of them in flight (the next one is
 * queued before the lists of the last one are fetched).  A single pair is never cut. */
int vsm_set_scratch_budget(vsm_ctx* ctx, long long bytes);
/* Same work, results left on the device; returns the total match count.  Fetch with
 * vsm_fetch_matches before the next submission on this context. */
long long vsm_match_pairs_device(vsm_ctx* ctx, const int32_t* pairs, int npairs,
                                 const vsm_params* p, int metric);
long long vsm_fetch_matches(vsm_ctx* ctx, int64_t* offsets, int32_t* idx0, int32_t* idx1,
                            long long cap);

/* ---- raw k-NN ------------------------------------------------------------------- */
/* For every query row: nearest database row (lowest index on ties), its key, and the
 * runner-up key.  u8: d1/d2 are int32 squared L2 (VSM_NO_SECOND when the database has
 * one row).  f32: float; DOT keys are dot products (larger is better), L2 keys are
 * squared distances: the two best rows are found by the key |b|^2 - 2 a.b the matrix cores
 * compute, then ordered, and their distances reported, in difference form sum (a[k] - b[k])^2
 * (what the reference ranks by, model_2_0.py:68; exact to 1e-5 relative for near-duplicates too).
 * Three or more rows within the key's rounding error (about 1e-7 of |a|^2 + |b|^2) of one another are beyond what
 * this settles: only the key's two best rows are compared in difference form.
 * idx2 may be NULL.  Brute-force 1-NN of
 * NearestEmbedFunc.forward (model_2_0.py:43-78) is the k=1 part of this call. */
int vsm_knn2(vsm_ctx* ctx, int set_q, int set_db, int metric,
             int32_t* idx1, int32_t* idx2, void* d1, void* d2);
/* Merge of per-slice vsm_knn2 results for a database cut into contiguous index
 * ranges (slice s owns global rows [base[s], base[s+1])).  Inputs are host arrays
 * [nslices][nq] with slice-local indices; outputs are global. */
int vsm_merge_top2_f32(const float* k1, const float* k2, const int32_t* i1, const int32_t* i2,
                       const int64_t* base, int nslices, int nq, int metric,
                       float* ok1, float* ok2, int64_t* oi1, int64_t* oi2);
/* The same two steps without the host in between, for a database sliced over GPUs: vsm_knn2_device
 * leaves one slice's result in a caller-owned device buffer of 4 * nq 32-bit words (best key,
 * runner-up key as float bits, best row, runner-up row; float sets only) that can go straight into an
 * all-gather; vsm_merge_top2_device merges the gathered [nslices][4][nq] buffer on the device and
 * returns the global lists.  The caller orders its own stream against these calls (both return
 * after their work has finished). */
int vsm_knn2_device(vsm_ctx* ctx, int set_q, int set_db, int metric, void* packed_dev);
int vsm_merge_top2_device(vsm_ctx* ctx, const void* packed_dev, const int64_t* base, int nslices, int nq,
                          int metric, float* ok1, float* ok2, int64_t* oi1, int64_t* oi2);
/* A database sliced over the GPUs of this machine, in one process (the multi-GPU form of the same brute-force
 * search; the reference searches its 512 codes on one device, model_2_0.py:43-78).  devices[ndev] are distinct HIP
 * devices; the database rows are cut into nslices contiguous ranges of (nearly) equal size, nslices a multiple of
 * ndev, device d holding slices d * nslices / ndev ... (a context of its own per device, created here).
 * vsm_dbq_load uploads the rows (any earlier database is dropped).  vsm_dbq_query uploads the queries to every
 * device, searches every slice (vsm_knn2_device), brings the per-slice lists together with an RCCL all-gather over
 * the devices (librccl is loaded when the first handle is created; VSM_ENODEV when it is not installed) and merges
 * them on the first device (vsm_merge_top2_device): k1 / k2 / i1 / i2 [nq] are keys and global rows of the nearest
 * row and the runner-up, as vsm_knn2 over the whole database returns them (-1 when the database has fewer rows).
 * Returns nq or a negative code; vsm_dbq_last_error(NULL) describes a failed create.  A handle is single-threaded. */
typedef struct vsm_dbq vsm_dbq;
vsm_dbq* vsm_dbq_create(const int* devices, int ndev, int nslices);
void vsm_dbq_destroy(vsm_dbq* q);
const char* vsm_dbq_last_error(const vsm_dbq* q);
int vsm_dbq_load(vsm_dbq* q, const float* rows, long long n, int dim, int layout);
/* Number of slices; base[nslices + 1] (may be NULL) receives the first database row of every slice. */
int vsm_dbq_slices(const vsm_dbq* q, int64_t* base);
int vsm_dbq_query(vsm_dbq* q, const float* queries, int nq, int layout, int metric,
                  float* k1, float* k2, int64_t* i1, int64_t* i2);
/* NearestEmbedFunc.forward (model_2_0.py:43-78): x [batch][dim][npts], emb [dim][k];
 * argmin [batch][npts] int64, result (may be NULL) [batch][dim][npts]. */
int vsm_nearest_embed(vsm_ctx* ctx, const float* x, int batch, int dim, int npts,
                      const float* emb, int k, int64_t* argmin, float* result);

/* ---- geometric verification of putative matches ----------------------------------- */
/* The step that follows matching in the workflow the reference feeds: a fundamental matrix and the
 * inlier matches per image pair, the content of a match record's TwoViewGeometry and inlier lists
 * (MatchFile.h:68-161, WriteFMatch / WriteIMatch MatchFile.cpp:596-748); the reference's Python side
 * gets it from cv2.findEssentialMat(..., RANSAC) (models/utils.py:288-312).  The reference holds no
 * estimator of its own; this one is RANSAC over the normalised eight-point solve with Sampson-distance
 * inliers, deterministic for a given seed (samples come from a counter-based generator keyed by seed,
 * the pair's key and the sample number).  x1^T F x0 = 0 for a keypoint x0 of the pair's first
 * image and its match x1 in the second; F has unit Frobenius norm and rank 2. */
typedef struct vsm_geom_params {
    float max_sampson;   /* inlier threshold on the Sampson distance, in pixels (> 0) */
    int hypotheses;      /* samples per pair; 0 = 256 */
    unsigned seed;
    int min_inliers;     /* a pair with fewer inliers reports none (NF = 0, F = 0) */
    long long first_pair; /* sampling key of the call's first pair; pair k of the call uses first_pair + k, so
                           * a pair list cut into calls (or shards) in any way gives the same results */
} vsm_geom_params;
/* Keypoint locations of one image, n x (x, y) floats (the first two columns of a .sift location
 * record, FeaturePoints.cpp:388-446).  Returns a set id (released with vsm_release_set). */
int vsm_upload_xy(vsm_ctx* ctx, const float* xy, int n);
/* xy_pairs[npairs][2]: keypoint sets of each pair's two images.  The putative matches come as
 * offsets[npairs + 1] / idx0 / idx1 (the layout vsm_match_pairs returns), or, with all three NULL, are
 * the lists of the last collected submission of npairs pairs, read where they lie on the device (they
 * stay there until the second submission after it; one later submission may be in flight).
 * Out: F[npairs][9] (row-major, may be NULL), nf[npairs] (may be NULL), in_offsets[npairs + 1] and
 * the inlier matches of all pairs in in_idx0 / in_idx1 (feature indices, in match order).
 * Returns the total number of inlier matches, or VSM_ECAP with in_offsets filled. */
long long vsm_verify_matches(vsm_ctx* ctx, const int32_t* xy_pairs, int npairs, const int64_t* offsets,
                             const int32_t* idx0, const int32_t* idx1, const vsm_geom_params* g,
                             float* F, int32_t* nf, int64_t* in_offsets, int32_t* in_idx0,
                             int32_t* in_idx1, long long cap);

/* Relative pose of each pair from its fundamental matrix and the two cameras' intrinsics: what the
 * reference's estimate_pose gets from cv2.findEssentialMat + cv2.recoverPose (models/utils.py:288-312)
 * and what a match record's TwoViewGeometry holds as NE, R, T (MatchFile.h:68-83).  E = K1^T F K0 is
 * decomposed into its four (R, +-t) candidates; the one with most inlier matches in front of both
 * cameras wins.  cameras[npairs][8] = fx, fy, cx, cy of the pair's first image, then of its second;
 * F[npairs][9] and the inlier lists as vsm_verify_matches returned them.  Out: R[npairs][9] (row-major),
 * T[npairs][3] with X1 = R X0 + T for a point X0 in the first camera's frame and |T| = 1 (the scale is
 * not observable), ne[npairs] = matches in front of both cameras (0: no pose, R = T = 0).  The reference
 * holds no estimator of its own (OpenCV's is not installed here): the procedure is this library's,
 * deterministic, and stated step by step beside the kernel.  Returns the number of pairs with a pose. */
int vsm_recover_pose(vsm_ctx* ctx, const int32_t* xy_pairs, int npairs, const float* cameras, const float* F,
                     const int64_t* in_offsets, const int32_t* in_idx0, const int32_t* in_idx1,
                     float* R, float* T, int32_t* ne);

/* ---- measurement ----------------------------------------------------------------- */
/* With timing on, the distance kernel of each submission is bracketed by HIP events on
 * the context's stream; vsm_last_kernel_ms returns the last bracket (synchronises). */
int vsm_set_timing(vsm_ctx* ctx, int on);
float vsm_last_kernel_ms(vsm_ctx* ctx);
/* Pairs in the submission that bracket belongs to (vsm_match_pairs may cut its list into several). */
int vsm_last_submission_pairs(const vsm_ctx* ctx);
/* Same for the mutual-check pass of a u8 submission (-1 when it did not run). */
float vsm_last_verify_ms(vsm_ctx* ctx);

/* ---- VisualSfM interchange files (host only, no device needed) ------------------- */
/* FeatureData::ReadSIFTB(file, locbuf, desbuf, nmax) (FeaturePoints.cpp:516-531) with
 * the header checks of the member reader (:388-446).  With a buffer it returns what that call returns: the rows
 * now in the buffers, min(nmax, rows in the file) (0 on failure, and for nmax <= 0); locbuf receives that many
 * location rows (the reference call copies the locations of ALL rows of the file into locbuf whatever nmax is).
 * With both buffers NULL it only reads the header and returns the rows in the file; *npoint and *nloc_dim (each
 * may be NULL) receive the header's values either way. */
int vsm_sift_read(const char* path, float* locbuf, uint8_t* desbuf, int nmax,
                  int* npoint, int* nloc_dim);
/* FeatureData::saveSIFTB2 (FeaturePoints.cpp:228-280).  Returns 1/0. */
int vsm_sift_write(const char* path, const float* loc, int nloc_dim,
                   const uint8_t* des, int npoint);
/* Lowe's ASCII keypoint file, FeatureData::ReadSIFTA (FeaturePoints.cpp:54-125): "npoint 128", then per
 * feature "row col scale orientation" and 128 integers.  locbuf[n][5] receives (col + 0.5, row + 0.5, 0,
 * scale, -orientation), desbuf the low 8 bits of every value.  NULL buffers query the count.  Returns the
 * number of features in the file (0 on failure; only 128-value descriptors, as the binary reader). */
int vsm_sift_read_ascii(const char* path, float* locbuf, uint8_t* desbuf, int nmax, int* npoint);
/* FeatureData::saveSIFTA (FeaturePoints.cpp:205-226) for loc[n][5].  Returns the count or a negative code;
 * writes nothing for an empty set, as there. */
int vsm_sift_write_ascii(const char* path, const float* loc, const uint8_t* des, int npoint);
/* FeatureData::SortSIFT (FeaturePoints.cpp:726-765) in place: descending scale (location column 3), features
 * of equal scale in descending order of their old position (what its bubble sort of the reversed sequence
 * leaves).  des may be NULL; order[npoint] (may be NULL) receives the old position of every new row. */
int vsm_sift_sort_by_scale(float* loc, int nloc_dim, uint8_t* des, int npoint, int32_t* order);
/* The in-memory part of FeatureData::ConvertA2B (FeaturePoints.cpp:655-695): if the first descriptor's norm
 * is below 400, every descriptor is scaled to norm 512 and rounded, and location values 0 and 1 change
 * places.  Returns 1 when it converted, 0 when the set was left alone.  An all-zero descriptor stays zero
 * (the reference divides by zero there). */
int vsm_sift_renormalise(float* loc, int nloc_dim, uint8_t* des, int npoint);
/* ReadSIFTA + ConvertA2B (FeaturePoints.cpp:54-72, 655-710): the binary file takes the ASCII file's place
 * ("x.sift.sift" -> "x.sift" and the ASCII file is removed; "x.sift" is overwritten; other names: nothing is
 * written).  out_path[cap] (may be NULL) receives the name written.  Returns the feature count or < 0. */
int vsm_sift_ascii_to_binary(const char* path, char* out_path, int cap);
/* save_feature_locations (main.py:133-140) / ReadLocFromText (FeaturePoints.cpp:326-369). */
int vsm_feature_txt_write(const char* path, const float* xy, int m);
int vsm_feature_txt_read(const char* path, float* xy, int cap);
/* The two conversions of the reference's converter program (test_main.cpp:30-60).
 * vsm_sift_to_txt: FeatureData::ReadSIFTB + Sift2Txt (FeaturePoints.cpp:306-325) for VSM_TXT_BOTH: one
 * row per feature, the location values (as `ofstream << float` prints them) then the 128 descriptor
 * bytes, every value followed by a blank; write_out_data (:281-305) for VSM_TXT_LOCATIONS /
 * VSM_TXT_DESCRIPTORS.  vsm_txt_to_sift: ReadLocFromText (:326-369) + saveSIFTB2 (:228-280): locations
 * (x, y, 0, 0, 0), descriptors of all ones.  Both return the feature count or a negative code. */
enum { VSM_TXT_BOTH = 0, VSM_TXT_LOCATIONS = 1, VSM_TXT_DESCRIPTORS = 2 };
int vsm_sift_to_txt(const char* sift_path, const char* txt_path, int what);
int vsm_txt_to_sift(const char* txt_path, const char* sift_path);
/* get_list_matching_pairs + save_matching_sfm (main.py:114-155): appends one record
 * built from the dense matches0[n] (-1 = none).  Returns the match count. */
int vsm_matchtxt_append_dense(const char* path, const char* name0, const char* name1,
                              const int32_t* matches0, int n);
/* Same record from a match list; n0 is image 0's keypoint count (it decides the
 * trailing-space quirk of main.py:124-129). */
int vsm_matchtxt_append(const char* path, const char* name0, const char* name1,
                        const int32_t* idx0, const int32_t* idx1, int nm, int n0);
/* Buffered form of the same writer for long runs (one open file, large writes); the bytes
 * equal those of repeated vsm_matchtxt_append calls.  truncate != 0 starts a new file. */
typedef struct vsm_txtwriter vsm_txtwriter;
vsm_txtwriter* vsm_matchtxt_open(const char* path, int truncate);
int vsm_matchtxt_write(vsm_txtwriter* w, const char* name0, const char* name1,
                       const int32_t* idx0, const int32_t* idx1, int nm, int n0);
int vsm_matchtxt_close(vsm_txtwriter* w);
/* generate_imagePairs (main.py:157-169).  Returns the pair count. */
int vsm_image_pairs(int nimages, int consecutive, const int32_t* discrete, int ndiscrete,
                    int32_t* pairs, int cap);
/* All i<j pairs (exhaustive matching).  Returns the pair count. */
long long vsm_image_pairs_exhaustive(int nimages, int32_t* pairs, long long cap);
/* Pair list by content, for collections that are not a sequence (generate_imagePairs, main.py:157-169,
 * goes by position only; the reference has no counterpart of this).  sets[nimg] are resident
 * descriptor sets, vocab_set a resident set of visual words of the same kind and dimension.  Every
 * descriptor row takes its nearest word (squared L2, lowest row among equals); a word held by df of
 * the nimg images weighs 1 + floor(log2(nimg / df)) (integer quotient); the score of two images is
 * the sum over words of weight * min(count in one, count in the other); every image chooses its
 * topk partners by (score descending, position ascending), scores of 0 excluded.  pairs receives the
 * union of the choices as positions in sets[], i < j, in ascending (i, j) order; scores (may be NULL)
 * the [nimg][nimg] score matrix.  All integer arithmetic.  Returns the pair count. */
long long vsm_shortlist_pairs(vsm_ctx* ctx, const int32_t* sets, int nimg, int vocab_set, int topk,
                              int32_t* pairs, long long cap, int32_t* scores);
/* A vocabulary for vsm_shortlist_pairs by k-means over all rows of the resident u8 sets (no counterpart in
 * the reference either; NearestEmbed's codebook, model_2_0.py:84-103, is learned by gradient descent).
 * K = min(nwords, total rows); word k starts as row floor(k * total / K) of the concatenation of the
 * sets; every iteration assigns each row to its nearest word (squared L2, lowest word among equals) and
 * replaces each word by the mean of its rows, per component (2 * sum + count) / (2 * count) in integers;
 * a word without rows keeps its value.  words receives K rows of 128 bytes.  Returns K. */
int vsm_train_vocabulary(vsm_ctx* ctx, const int32_t* sets, int nimg, int nwords, int iterations,
                         uint8_t* words);
/* A new resident u8 set holding rows 0, stride, 2 * stride, ... of the concatenation of the sets: a
 * subsample to train a vocabulary on (vsm_train_vocabulary over that one set).  Returns the set id;
 * release it with vsm_release_set. */
int vsm_sample_rows_u8(vsm_ctx* ctx, const int32_t* sets, int nimg, int stride);
/* MatchFile::WritePMatch(image1, image2, fc1, fc2, NM, matches) (MatchFile.cpp:751-832):
 * image paths without extension; writes <image1>.mat and <image2>.mat.  Re-writing a pair with more
 * matches than its block holds moves the record to the end of the file as MoveRecordToEnd does
 * (MatchFile.cpp:422-453), bookkeeping quirks included; a full directory doubles as in AddImageMatch
 * (:494-560).
 * reservation = MatchFile::SetRecordReservation (MatchFile.h:244); pass 0 for the
 * reference default of 20. */
int vsm_mat_write_pmatch(const char* image1, const char* image2, int fc1, int fc2,
                         int nm, const int32_t* idx1, const int32_t* idx2, int reservation);
/* MatchFile::multi_thread_mode (MatchFile.cpp:37, 221-231; switched by MatchFile.h:242-244): with it on the
 * reference holds a mutex named after the .mat file from OpenMatchFile to CloseMatchFile.  Here: on != 0 makes every
 * vsm_mat_* call of the process hold an advisory lock (flock) on a .mat file while it has it open, exclusive for the
 * writers, shared for the readers, one file at a time, so that several processes (one per GPU, each matching its
 * slice of the pair list) may write records of the same images with vsm_mat_write_pmatch / _fmatch / _ematch.  The
 * files then hold every record, in the order the writers happened to arrive.  Not for vsm_mat_writer_*: it keeps
 * directories in memory until it is closed and is for one process.  Returns the previous setting. */
int vsm_mat_set_locking(int on);
/* The same writer for long runs (SfM over thousands of pairs): every touched file's directory is
 * read once and kept in memory, at most max_open (<= 0: 256) descriptors stay open, directories are
 * written back by vsm_mat_writer_close.  The files end up byte for byte as repeated
 * vsm_mat_write_pmatch calls leave them; until close they are incomplete. */
typedef struct vsm_matwriter vsm_matwriter;
vsm_matwriter* vsm_mat_writer_open(int reservation, int max_open);
int vsm_mat_writer_put(vsm_matwriter* w, const char* image1, const char* image2, int fc1, int fc2,
                       int nm, const int32_t* idx1, const int32_t* idx2);
int vsm_mat_writer_close(vsm_matwriter* w);
/* MatchFile(image).GetPMatch(other, FC, NM, matches) (MatchFile.cpp:856-895). */
int vsm_mat_get_pmatch(const char* image, const char* other, int fc_other,
                       int32_t* idx_this, int32_t* idx_other, int cap);
/* MatchFile(image).GetPMatchR (MatchFile.cpp:897-936): the same record with the two rows
 * exchanged: row0 = indices in `other`, row1 = indices in `image`. */
int vsm_mat_get_pmatch_r(const char* image, const char* other, int fc_other,
                         int32_t* row0, int32_t* row1, int cap);
/* MatchFile::GetMatchCount (MatchFile.cpp:938-958). Returns 1/0. */
int vsm_mat_get_match_count(const char* image, const char* other, int* nm, int* nf);

/* MatchFile::WriteFMatch (MatchFile.cpp:596-605 -> WriteIMatch :643-748): geometry (NF, F) and the
 * inlier index lists into the record of the pair in both images' files (F transposed and the rows
 * swapped in the second, as SetGeometryR and :735-746 do).  The pair's putative record must exist
 * (vsm_mat_write_pmatch).  Returns 1, 0 if a file has no such record, or a negative code.
 * A record whose block cannot hold the inlier lists (a file from elsewhere, or nf above the record's NM; no record
 * written by vsm_mat_write_pmatch with nf <= NM) goes through the re-allocation branch (MatchFile.cpp:690-719): it
 * takes a larger block where it lies when it is the last of the file, the dead bytes behind it when those suffice,
 * and otherwise moves to the end of the file.  In the last case the reference writes geometry and inliers at the new
 * place and leaves NM and the putative lists at the old one (its own readers then find NM = 0); here they move with
 * the record. */
int vsm_mat_write_fmatch(const char* image1, const char* image2, int nf, const int32_t* idx1,
                         const int32_t* idx2, const float* F);
/* MatchFile::WriteIMatch (MatchFile.cpp:643-748) with the whole two-view geometry: NF, F and the inlier
 * lists as vsm_mat_write_fmatch writes them, plus NE, R, T and the two focal lengths F1, F2
 * (MatchFile.h:68-83).  The second image's file gets the exchanged view, as SetGeometryR does
 * (MatchFile.h:121-141): F and R transposed, T = -R^T T, F1 and F2 swapped.  ne = 0 writes no pose. */
int vsm_mat_write_ematch(const char* image1, const char* image2, int nf, const int32_t* idx1, const int32_t* idx2,
                         const float* F, int ne, const float* R, const float* T, float f1, float f2);
int vsm_mat_writer_put_ematch(vsm_matwriter* w, const char* image1, const char* image2, int nf,
                              const int32_t* idx1, const int32_t* idx2, const float* F, int ne, const float* R,
                              const float* T, float f1, float f2);
/* The pose fields of the record of `other` in `image`'s file, as stored there: *ne, R[9], T[3],
 * f12[2] = F1, F2.  Returns NM of the record (0 when absent or rejected by the reference's checks). */
int vsm_mat_get_pose(const char* image, const char* other, int fc_other, int* ne, float* R, float* T, float* f12);
/* MatchFile::GetIMatch / GetIMatchR (MatchFile.cpp:961-1084): F[9], *nf and up to cap inlier pairs;
 * returns NM of the record (0 when absent or rejected by the reference's checks). */
/* The same through the cached writer that vsm_mat_writer_open returns. */
int vsm_mat_writer_put_fmatch(vsm_matwriter* w, const char* image1, const char* image2, int nf,
                              const int32_t* idx1, const int32_t* idx2, const float* F);
int vsm_mat_get_imatch(const char* image, const char* other, int fc_other, float* F, int* nf,
                       int32_t* row0, int32_t* row1, int cap);
int vsm_mat_get_imatch_r(const char* image, const char* other, int fc_other, float* F, int* nf,
                         int32_t* row0, int32_t* row1, int cap);

#ifdef __cplusplus
}
#endif
#endif
