// Calls every compute entry point of the C ABI once or twice on small random inputs (no result
// checks: the parity tests do those).  Built with AddressSanitizer on the host side by
// tools/build_asan_cli.sh, it looks for host memory errors in the ABI layer.
#include "vsm.h"

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        long long r_ = (long long)(x);                                                   \
        if (r_ < 0) {                                                                    \
            fprintf(stderr, "%s -> %lld: %s\n", #x, r_, vsm_last_error(ctx));            \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

int main()
{
    vsm_ctx* ctx = vsm_create(0);
    if (!ctx) {
        fprintf(stderr, "vsm_create: %s\n", vsm_last_error(nullptr));
        return 1;
    }
    std::mt19937 rng(11);
    std::normal_distribution<float> gauss;
    auto floats = [&](int n, int dim) {
        std::vector<float> v((size_t)n * dim + 1);
        for (int i = 0; i < n; ++i) {
            double s = 0;
            for (int d = 0; d < dim; ++d) { v[(size_t)i * dim + d] = gauss(rng); s += (double)v[(size_t)i * dim + d] * v[(size_t)i * dim + d]; }
            for (int d = 0; d < dim; ++d) v[(size_t)i * dim + d] /= (float)std::sqrt(s);
        }
        return v;
    };
    auto bytes = [&](int n) {
        std::vector<uint8_t> v((size_t)n * 128 + 1);
        for (auto& b : v) b = (uint8_t)(rng() % 120);
        return v;
    };
    vsm_params p{4, 5, 1, 0, -1, -1e30f, -1.0f};
    vsm_set_timing(ctx, 1);
    // u8
    const int n0 = 700, n1 = 1300;
    auto a = bytes(n0), b = bytes(n1);
    int s0 = vsm_upload_u8(ctx, a.data(), n0), s1 = vsm_upload_u8(ctx, b.data(), n1), se = vsm_upload_u8(ctx, nullptr, 0);
    CHECK(s0); CHECK(s1); CHECK(se);
    std::vector<int32_t> i0(n1 + 1), i1(n1 + 1), d(n1 + 1), m0(n1 + 1), m1(n1 + 1);
    CHECK(vsm_match_pair(ctx, s0, s1, &p, VSM_METRIC_L2, i0.data(), i1.data(), d.data(), n0));
    CHECK(vsm_match_pair(ctx, s0, se, &p, VSM_METRIC_L2, i0.data(), i1.data(), nullptr, n0));
    CHECK(vsm_match_pair_dense(ctx, s0, s1, &p, VSM_METRIC_L2, m0.data(), m1.data()));
    CHECK(vsm_knn2(ctx, s0, s1, VSM_METRIC_L2, i0.data(), nullptr, d.data(), m0.data()));
    const int32_t pairs[] = {s0, s1, s1, s0, s0, se, se, s1, s0, s0};
    std::vector<int64_t> off(6);
    std::vector<int32_t> t0(4 * n1), t1(4 * n1);
    CHECK(vsm_match_pairs(ctx, pairs, 5, &p, VSM_METRIC_L2, off.data(), t0.data(), t1.data(), 4 * n1));
    CHECK(vsm_submit_pairs(ctx, pairs, 5, &p, VSM_METRIC_L2));
    CHECK(vsm_submit_pairs(ctx, pairs + 2, 3, &p, VSM_METRIC_L2));
    CHECK(vsm_collect_matches(ctx, off.data(), t0.data(), t1.data(), 4 * n1));
    CHECK(vsm_collect_matches(ctx, off.data(), t0.data(), t1.data(), 4 * n1));
    long long tot = vsm_match_pairs_device(ctx, pairs, 5, &p, VSM_METRIC_L2);
    CHECK(tot);
    CHECK(vsm_fetch_matches(ctx, off.data(), t0.data(), t1.data(), 4 * n1));
    // geometry on the lists just produced and on host lists
    std::vector<float> xy((size_t)n1 * 2);
    for (auto& v : xy) v = (float)(rng() % 2000);
    int x0 = vsm_upload_xy(ctx, xy.data(), n0), x1 = vsm_upload_xy(ctx, xy.data(), n1), xe = vsm_upload_xy(ctx, nullptr, 0);
    CHECK(x0); CHECK(x1); CHECK(xe);
    const int32_t xp[] = {x0, x1, x1, x0, x0, xe, xe, x1, x0, x0};
    vsm_geom_params g{2.0f, 64, 3u, 0, 17};
    std::vector<float> F(45);
    std::vector<int32_t> nf(5), g0(4 * n1), g1(4 * n1);
    std::vector<int64_t> goff(6);
    CHECK(vsm_verify_matches(ctx, xp, 5, nullptr, nullptr, nullptr, &g, F.data(), nf.data(), goff.data(), g0.data(), g1.data(), 4 * n1));
    CHECK(vsm_verify_matches(ctx, xp, 5, off.data(), t0.data(), t1.data(), &g, F.data(), nf.data(), goff.data(), g0.data(), g1.data(), 4 * n1));
    // floats, both layouts and both search modes
    for (int dim : {256, 100, 7}) {
        const int q = 500, r = 900;
        auto fa = floats(q, dim), fb = floats(r, dim);
        std::vector<float> fbt((size_t)r * dim);
        for (int i = 0; i < r; ++i)
            for (int k = 0; k < dim; ++k) fbt[(size_t)k * r + i] = fb[(size_t)i * dim + k];
        int f0 = vsm_upload_f32(ctx, fa.data(), q, dim, VSM_LAYOUT_ROWS), f1 = vsm_upload_f32(ctx, fbt.data(), r, dim, VSM_LAYOUT_DIM_MAJOR);
        CHECK(f0); CHECK(f1);
        std::vector<int32_t> j1(r + 1), j2(r + 1), w0(r + 1), w1(r + 1);
        std::vector<float> k1(r + 1), k2(r + 1);
        for (int mode : {VSM_F32_FAST_EXACT, VSM_F32_MFMA_F32}) {
            CHECK(vsm_set_f32_mode(ctx, mode));
            vsm_params pf{0, 1, 1, 0, -1, 0.1f, -1.0f};
            CHECK(vsm_knn2(ctx, f0, f1, VSM_METRIC_DOT, j1.data(), j2.data(), k1.data(), k2.data()));
            CHECK(vsm_knn2(ctx, f0, f1, VSM_METRIC_L2, j1.data(), nullptr, k1.data(), nullptr));
            CHECK(vsm_match_pair(ctx, f0, f1, &pf, VSM_METRIC_DOT, w0.data(), w1.data(), k1.data(), q));
            CHECK(vsm_match_pair_dense(ctx, f0, f1, &pf, VSM_METRIC_DOT, w0.data(), w1.data()));
            const int32_t fp[] = {f0, f1, f1, f0, f0, f0};
            std::vector<int64_t> fo(4);
            std::vector<int32_t> u0(3 * r), u1(3 * r);
            for (int rev : {VSM_F32_REVERSE_ALL, VSM_F32_REVERSE_CANDIDATES, VSM_F32_REVERSE_AUTO}) {
                CHECK(vsm_set_f32_reverse(ctx, rev));
                CHECK(vsm_match_pairs(ctx, fp, 3, &pf, VSM_METRIC_DOT, fo.data(), u0.data(), u1.data(), 3 * r));
                vsm_params pl = pf;
                pl.ratio_num = 4; pl.ratio_den = 5;
                CHECK(vsm_match_pairs(ctx, fp, 3, &pl, VSM_METRIC_L2, fo.data(), u0.data(), u1.data(), 3 * r));
            }
        }
        CHECK(vsm_set_f32_mode(ctx, VSM_F32_FAST_EXACT));
        // sliced database: device-resident per-slice results, device merge
        void* packed = nullptr;
        if (hipMalloc(&packed, (size_t)2 * 4 * q * 4) != hipSuccess) return 1;
        CHECK(vsm_knn2_device(ctx, f0, f1, VSM_METRIC_L2, packed));
        CHECK(vsm_knn2_device(ctx, f0, f1, VSM_METRIC_L2, (char*)packed + (size_t)4 * q * 4));
        const int64_t base[3] = {0, r, 2 * r};
        std::vector<float> ok1(q), ok2(q);
        std::vector<int64_t> oi1(q), oi2(q);
        CHECK(vsm_merge_top2_device(ctx, packed, base, 2, q, VSM_METRIC_L2, ok1.data(), ok2.data(), oi1.data(), oi2.data()));
        (void)hipFree(packed);
        CHECK(vsm_release_set(ctx, f0));
        CHECK(vsm_release_set(ctx, f1));
    }
    {   // NearestEmbedFunc.forward shape
        const int batch = 2, dim = 64, npts = 33, k = 50;
        auto x = floats(batch * npts, dim), emb = floats(k, dim);
        std::vector<int64_t> arg((size_t)batch * npts);
        std::vector<float> res((size_t)batch * dim * npts);
        CHECK(vsm_nearest_embed(ctx, x.data(), batch, dim, npts, emb.data(), k, arg.data(), res.data()));
    }
    {   // pair shortlist over u8 sets (one of them empty) with a small vocabulary, with and without the score matrix
        auto words = bytes(37);
        const int vs = vsm_upload_u8(ctx, words.data(), 37);
        CHECK(vs);
        const int32_t imgs[] = {s0, s1, se, s0};
        std::vector<int32_t> sp(2 * 6), sc(16);
        CHECK(vsm_shortlist_pairs(ctx, imgs, 4, vs, 2, sp.data(), 6, sc.data()));
        CHECK(vsm_shortlist_pairs(ctx, imgs, 4, vs, 1, sp.data(), 6, nullptr));
        std::vector<uint8_t> trained(50 * 128);
        CHECK(vsm_train_vocabulary(ctx, imgs, 4, 50, 2, trained.data()));
        const int sampled = vsm_sample_rows_u8(ctx, imgs, 4, 9);
        CHECK(sampled);
        CHECK(vsm_train_vocabulary(ctx, &sampled, 1, 20, 1, trained.data()));
        vsm_release_set(ctx, sampled);
    }
    printf("kernel ms %.3f, verify ms %.3f, last pairs %d\n", vsm_last_kernel_ms(ctx), vsm_last_verify_ms(ctx), vsm_last_submission_pairs(ctx));
    vsm_destroy(ctx);
    puts("abi exerciser done");
    return 0;
}
