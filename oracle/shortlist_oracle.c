/*
 * CPU oracle of the image-pair shortlist.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference picks pairs by position in the sequence only (generate_imagePairs,
 * Others/SuperPoint_data/main.py:157-169); choosing them by content has no counterpart there
 * ("parity unpinned"): this file is the normative definition of what the library computes.
 *
 *   words[i][r]   nearest vocabulary row of descriptor r of image i (orc_knn2_u8 / orc_knn2_f32 with
 *                 the L2 metric supply it: lowest row among equal distances)
 *   h[i][w]       how many rows of image i took word w
 *   df[w]         how many images hold word w;  weight[w] = 0 if df == 0, else
 *                 1 + floor(log2(nimg / df)) with the integer quotient
 *   score[i][j]   sum over w of weight[w] * min(h[i][w], h[j][w])
 *   choice        every image takes its topk partners by (score descending, position ascending),
 *                 scores of 0 excluded; the result is the union of the choices as (i, j), i < j, in
 *                 ascending order.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

static int floor_log2(int v)
{
    int l = 0;
    while (v > 1) {
        v >>= 1;
        ++l;
    }
    return l;
}

/* words: concatenated per image, offsets[nimg + 1].  scores: [nimg][nimg] or NULL.
 * Returns the pair count (all of them counted, at most cap written). */
long long orc_shortlist(const int32_t* words, const int64_t* offsets, int nimg, int nwords, int topk,
                        int32_t* pairs, long long cap, int32_t* scores)
{
    if (nimg <= 0) return 0;
    uint32_t* h = (uint32_t*)calloc((size_t)nimg * nwords, sizeof(uint32_t));
    uint32_t* weight = (uint32_t*)calloc((size_t)nwords, sizeof(uint32_t));
    int32_t* sc = (int32_t*)calloc((size_t)nimg * nimg, sizeof(int32_t));
    unsigned char* mark = (unsigned char*)calloc((size_t)nimg * nimg, 1);
    unsigned char* taken = (unsigned char*)malloc((size_t)nimg);
    for (int i = 0; i < nimg; ++i)
        for (int64_t r = offsets[i]; r < offsets[i + 1]; ++r) h[(size_t)i * nwords + words[r]] += 1;
    for (int w = 0; w < nwords; ++w) {
        int df = 0;
        for (int i = 0; i < nimg; ++i) df += h[(size_t)i * nwords + w] != 0;
        weight[w] = df ? (uint32_t)(1 + floor_log2(nimg / df)) : 0;
    }
    for (int i = 0; i < nimg; ++i)
        for (int j = 0; j < nimg; ++j) {
            int64_t s = 0;
            for (int w = 0; w < nwords; ++w) {
                const uint32_t a = h[(size_t)i * nwords + w], b = h[(size_t)j * nwords + w];
                s += (int64_t)weight[w] * (a < b ? a : b);
            }
            sc[(size_t)i * nimg + j] = (int32_t)s;
        }
    for (int i = 0; i < nimg; ++i) {
        memset(taken, 0, (size_t)nimg);
        for (int r = 0; r < topk; ++r) {
            int best = -1;
            for (int j = 0; j < nimg; ++j) {
                if (j == i || taken[j] || sc[(size_t)i * nimg + j] <= 0) continue;
                if (best < 0 || sc[(size_t)i * nimg + j] > sc[(size_t)i * nimg + best]) best = j;
            }
            if (best < 0) break;
            taken[best] = 1;
            if (i < best) mark[(size_t)i * nimg + best] = 1;
            else mark[(size_t)best * nimg + i] = 1;
        }
    }
    long long n = 0;
    for (int i = 0; i < nimg; ++i)
        for (int j = i + 1; j < nimg; ++j)
            if (mark[(size_t)i * nimg + j]) {
                if (n < cap) {
                    pairs[2 * n] = i;
                    pairs[2 * n + 1] = j;
                }
                ++n;
            }
    if (scores) memcpy(scores, sc, (size_t)nimg * nimg * sizeof(int32_t));
    free(h);
    free(weight);
    free(sc);
    free(mark);
    free(taken);
    return n;
}

/* k-means vocabulary over rows[total][128] (the concatenation of the images' descriptors): word k starts
 * as row floor(k * total / K), K = min(nwords, total); every iteration assigns each row to its nearest
 * word (orc_knn2_u8: squared L2, lowest word among equals) and replaces each word by the rounded mean of
 * its rows, (2 * sum + count) / (2 * count) per component; a word without rows keeps its value.
 * Returns K. */
int orc_train_vocabulary(const uint8_t* rows, long long total, int nwords, int iterations, int threads,
                         uint8_t* words)
{
    if (total <= 0 || nwords < 1) return 0;
    const int K = (int)(nwords < total ? nwords : total);
    for (int k = 0; k < K; ++k) {
        const long long g = (long long)(((__int128)k * total) / K);
        memcpy(words + (size_t)k * 128, rows + (size_t)g * 128, 128);
    }
    int32_t* idx = (int32_t*)malloc((size_t)total * sizeof(int32_t));
    int32_t* d1 = (int32_t*)malloc((size_t)total * sizeof(int32_t));
    int32_t* d2 = (int32_t*)malloc((size_t)total * sizeof(int32_t));
    uint64_t* sums = (uint64_t*)malloc((size_t)K * 128 * sizeof(uint64_t));
    uint64_t* counts = (uint64_t*)malloc((size_t)K * sizeof(uint64_t));
    for (int it = 0; it < iterations; ++it) {
        orc_knn2_u8(rows, (int)total, words, K, threads, idx, d1, d2);
        memset(sums, 0, (size_t)K * 128 * sizeof(uint64_t));
        memset(counts, 0, (size_t)K * sizeof(uint64_t));
        for (long long r = 0; r < total; ++r) {
            counts[idx[r]] += 1;
            for (int k = 0; k < 128; ++k) sums[(size_t)idx[r] * 128 + k] += rows[(size_t)r * 128 + k];
        }
        for (int w = 0; w < K; ++w)
            if (counts[w])
                for (int k = 0; k < 128; ++k)
                    words[(size_t)w * 128 + k] = (uint8_t)((2 * sums[(size_t)w * 128 + k] + counts[w]) / (2 * counts[w]));
    }
    free(idx);
    free(d1);
    free(d2);
    free(sums);
    free(counts);
    return K;
}
