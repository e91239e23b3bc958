#include "vsm.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
int main(int argc, char** argv)
{
    std::mt19937 rng(5);
    const std::string dir = argc > 1 ? argv[1] : "/tmp/vsm_san_work";
    if (system(("rm -rf " + dir + " && mkdir -p " + dir + "/sub").c_str())) return 9;
    // .sift round trip, several sizes incl. 0
    for (int n : {0, 1, 3, 777}) {
        std::vector<float> loc((size_t)n * 5 + 1);
        std::vector<uint8_t> des((size_t)n * 128 + 1);
        for (auto& v : loc) v = (float)(rng() % 1000);
        for (auto& v : des) v = (uint8_t)rng();
        const std::string p = dir + "/" + std::to_string(n) + ".sift";
        if (!vsm_sift_write(p.c_str(), loc.data(), 5, des.data(), n)) return 1;
        int np = 0, nl = 0;
        if (n && !vsm_sift_read(p.c_str(), nullptr, nullptr, 0, &np, &nl)) return 2;
        std::vector<float> l2((size_t)n * 5 + 1);
        std::vector<uint8_t> d2((size_t)n * 128 + 1);
        if (n && !vsm_sift_read(p.c_str(), l2.data(), d2.data(), n, nullptr, nullptr)) return 3;
        if (n && (memcmp(l2.data(), loc.data(), (size_t)n * 20) || memcmp(d2.data(), des.data(), (size_t)n * 128))) return 4;
        // truncated read
        if (n > 2) vsm_sift_read(p.c_str(), l2.data(), d2.data(), 2, nullptr, nullptr);
    }
    // match txt
    vsm_txtwriter* tw = vsm_matchtxt_open((dir + "/m.txt").c_str(), 1);
    for (int k = 0; k < 50; ++k) {
        const int nm = (int)(rng() % 300);
        std::vector<int32_t> a(nm + 1), b(nm + 1);
        for (int i = 0; i < nm; ++i) { a[i] = i * 3; b[i] = (int32_t)(rng() % 100000); }
        if (nm) a[nm - 1] = (k & 1) ? 999 : a[nm - 1];
        vsm_matchtxt_write(tw, "a.jpg", "bb.jpg", a.data(), b.data(), nm, 1000);
    }
    vsm_matchtxt_close(tw);
    // .mat: many partners (directory regrow), rewrites with larger lists (move to end), inlier records, readers
    const int N = 60;
    vsm_matwriter* w = vsm_mat_writer_open(0, 4);
    std::vector<std::vector<int32_t>> keep0, keep1;
    for (int round = 0; round < 2; ++round)
        for (int i = 0; i < N; ++i)
            for (int j = i + 1; j < N; j += 1 + (int)(rng() % 3)) {
                const int nm = (int)(rng() % (round ? 90 : 40));
                std::vector<int32_t> a(nm + 1), b(nm + 1);
                for (int t = 0; t < nm; ++t) { a[t] = t; b[t] = (int32_t)(rng() % 500); }
                const std::string p0 = dir + "/" + std::to_string(i), p1 = dir + (j % 2 ? "/sub/" : "/") + std::to_string(j);
                if (vsm_mat_writer_put(w, p0.c_str(), p1.c_str(), 500, 500, nm, a.data(), b.data()) < 0) return 5;
                float F[9] = {1, 2, 3, 4, 5, 6, 7, 8, 9};
                const int nf = nm / 2;
                if (vsm_mat_writer_put_fmatch(w, p0.c_str(), p1.c_str(), nf, a.data(), b.data(), F) < 0) return 6;
            }
    if (vsm_mat_writer_close(w) != VSM_OK) return 7;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            const std::string p0 = dir + "/" + std::to_string(i), p1 = dir + (j % 2 ? "/sub/" : "/") + std::to_string(j);
            std::vector<int32_t> a(600), b(600);
            float F[9];
            int nf = 0, nm = 0, nf2 = 0;
            vsm_mat_get_pmatch(p0.c_str(), p1.c_str(), 500, a.data(), b.data(), 600);
            vsm_mat_get_pmatch_r(p1.c_str(), p0.c_str(), 500, a.data(), b.data(), 3);
            vsm_mat_get_imatch(p0.c_str(), p1.c_str(), 500, F, &nf, a.data(), b.data(), 600);
            vsm_mat_get_imatch_r(p1.c_str(), p0.c_str(), 500, F, &nf, a.data(), b.data(), 5);
            vsm_mat_get_match_count(p0.c_str(), p1.c_str(), &nm, &nf2);
        }
    // direct writers
    int32_t a[4] = {0, 1, 2, 3}, b[4] = {3, 2, 1, 0};
    float F[9] = {0};
    vsm_mat_write_pmatch((dir + "/x").c_str(), (dir + "/y").c_str(), 10, 10, 4, a, b, 0);
    vsm_mat_write_fmatch((dir + "/x").c_str(), (dir + "/y").c_str(), 2, a, b, F);
    vsm_mat_write_fmatch((dir + "/x").c_str(), (dir + "/nope").c_str(), 2, a, b, F);
    // keypoint txt, pair lists
    float xy[6] = {1.5f, 2.5f, 3, 4, 5, 6}, back[6];
    vsm_feature_txt_write((dir + "/k.txt").c_str(), xy, 3);
    vsm_feature_txt_read((dir + "/k.txt").c_str(), back, 3);
    vsm_feature_txt_read((dir + "/k.txt").c_str(), back, 1);
    // the converter's two directions, and a keypoint txt that announces more rows than it holds
    if (vsm_txt_to_sift((dir + "/k.txt").c_str(), (dir + "/k.sift").c_str()) != 3) return 1;
    for (int what = 0; what < 3; ++what)
        if (vsm_sift_to_txt((dir + "/k.sift").c_str(), (dir + "/k_back.txt").c_str(), what) != 3) return 1;
    {
        FILE* fp = fopen((dir + "/short.txt").c_str(), "w");
        fputs("5 0\n1 2\n", fp);
        fclose(fp);
        if (vsm_txt_to_sift((dir + "/short.txt").c_str(), (dir + "/short.sift").c_str()) >= 0) return 1;
        if (vsm_sift_to_txt((dir + "/short.txt").c_str(), (dir + "/o.txt").c_str(), 0) >= 0) return 1;
    }
    int32_t disc[2] = {7, 15};
    std::vector<int32_t> pairs(2 * 5000);
    vsm_image_pairs(1000, 3, disc, 2, pairs.data(), 5000);
    vsm_image_pairs_exhaustive(50, pairs.data(), 5000);
    puts("sanitizer driver done");
    return 0;
}
