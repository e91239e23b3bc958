// vsm_match_pairs — command-line driver: a folder of VisualSfM .sift files in, matching.txt
// (and optionally .mat putative-match records) out.
//
// It is the counterpart of the reference's two drivers around the matcher: the batch loop of
// Others/visualSfM_Sift_file/Match_modified/test_main.cpp:46-60 (files named <i>.sift, i = 0..N-1)
// and the per-pair loop of Others/SuperPoint_data/main.py:208-280 (pair list, one three-line record
// per pair appended to matching.txt, image names "<i>.jpg" as main.py:64 assigns them).
#include "../../include/vsm.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Options {
    std::string sift_dir, out = "matching.txt", name_ext = ".jpg", sift_ext = ".sift";
    int num = 0, batch = 4096;
    std::vector<int> devices{0};   // one matching context, with its own copy of every descriptor set, per entry
    int shard = 0, nshards = 1;
    bool exhaustive = true, mutual = true, write_mat = false, append = false;
    bool by_vocab = false;        // pair list from a visual vocabulary (vsm_shortlist_pairs)
    int vocab_size = 4096, topk = 20, kmeans = 0, kmeans_stride = 1;
    std::string vocab_file;       // .sift file whose descriptors are the words; empty: rows drawn from the images
    std::string pairs_file;       // explicit pair list: one "i j" per line
    int consecutive = 3;
    std::vector<int32_t> discrete{7, 15};
    int ratio_num = 4, ratio_den = 5;
    long long max_dist = -1;
    float verify = 0.0f;          // > 0: geometric verification, Sampson threshold in pixels
    int samples = 256;
    unsigned seed = 0;
    int min_inliers = 0;
    std::string inlier_out;       // match txt of the verified matches
    bool have_camera = false;     // --camera: relative pose per verified pair
    float camera[4] = {0, 0, 0, 0};   // fx, fy, cx, cy (one camera for the whole collection)
};

void usage()
{
    fprintf(stderr,
            "usage: vsm_match_pairs --sift-dir DIR --num N [options]\n"
            "  --out FILE            match txt to write (default matching.txt)\n"
            "  --append              append to FILE instead of starting it\n"
            "  --pairs exhaustive|window|vocab   all i<j pairs (default), the reference's window list, or every\n"
            "                        image's --topk most similar images by visual-word histograms\n"
            "  --pairs-file FILE     the pair list itself: one \"i j\" (image numbers) per line, matched in file order\n"
            "  --topk T              vocab list: partners chosen per image (default 20)\n"
            "  --vocab FILE.sift     vocab list: the words; without it --vocab-size K (default 4096) rows are\n"
            "                        taken from the images themselves, evenly spread\n"
            "  --kmeans N            vocab list without --vocab: the words are the result of N k-means iterations\n"
            "                        over all descriptors (default 0: the drawn rows as they are);\n"
            "                        --kmeans-stride S trains on every S-th descriptor only\n"
            "  --consecutive K       window list: pairs (i, i+1..i+K)          (default 3)\n"
            "  --discrete A,B,...    window list: pairs (i, i+A), (i, i+B) ... (default 7,15)\n"
            "  --ratio NUM/DEN       Lowe ratio, exact rational (default 4/5; 0/1 disables)\n"
            "  --max-dist D          keep matches with squared distance <= D\n"
            "  --no-mutual           skip the mutual-nearest-neighbour check\n"
            "  --mat                 also write <DIR>/<i>.mat putative-match records (with --shard every process locks a\n"
            "                        file while it updates it: all records arrive, in no fixed order within a file)\n"
            "  --verify PIXELS       geometric verification of every pair's matches (fundamental-matrix RANSAC,\n"
            "                        Sampson-distance threshold); with --mat the inlier records are written too\n"
            "  --samples N           RANSAC samples per pair (default 256); --seed S; --min-inliers K\n"
            "  --inlier-out FILE     match txt of the verified matches (same format as --out)\n"
            "  --camera FX,FY,CX,CY  intrinsics of the collection's camera: with --verify, the relative pose of every\n"
            "                        verified pair (E = K^T F K decomposed); with --mat it goes into the records (NE, R, T)\n"
            "  --name-ext EXT        image-name suffix in the match txt (default .jpg)\n"
            "  --device N, --batch P (pairs per GPU submission, default 4096; two are in flight)\n"
            "  --devices A,B,...     several GPUs of this machine in one process: every one holds all descriptors and takes\n"
            "                        the next submission of the pair list when it is free; one writer thread emits the\n"
            "                        results in pair order, so FILE (and the .mat records, the verification outputs) are\n"
            "                        byte for byte those of a one-GPU run.  A device may be named twice (two contexts)\n"
            "  --shard R/W           match only the R-th of W contiguous slices of the pair list, cut at equal\n"
            "                        shares of the work (rows x rows per pair) (one process\n"
            "                        per GPU); with W > 1 the text goes to FILE.partR: concatenating the parts in\n"
            "                        order gives the file of an unsharded run\n");
}

bool parse(int argc, char** argv, Options& o)
{
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&](const char* what) -> const char* {
            if (i + 1 >= argc) {
                fprintf(stderr, "missing value for %s\n", what);
                return nullptr;
            }
            return argv[++i];
        };
        const char* v = nullptr;
        if (a == "--sift-dir") { if (!(v = next("--sift-dir"))) return false; o.sift_dir = v; }
        else if (a == "--num") { if (!(v = next("--num"))) return false; o.num = atoi(v); }
        else if (a == "--out") { if (!(v = next("--out"))) return false; o.out = v; }
        else if (a == "--append") o.append = true;
        else if (a == "--pairs") {
            if (!(v = next("--pairs"))) return false;
            o.by_vocab = strcmp(v, "vocab") == 0;
            o.exhaustive = !o.by_vocab && strcmp(v, "window") != 0;
        }
        else if (a == "--pairs-file") { if (!(v = next("--pairs-file"))) return false; o.pairs_file = v; }
        else if (a == "--topk") { if (!(v = next("--topk"))) return false; o.topk = atoi(v); }
        else if (a == "--vocab") { if (!(v = next("--vocab"))) return false; o.vocab_file = v; }
        else if (a == "--kmeans-stride") { if (!(v = next("--kmeans-stride"))) return false; o.kmeans_stride = atoi(v); }
        else if (a == "--kmeans") { if (!(v = next("--kmeans"))) return false; o.kmeans = atoi(v); }
        else if (a == "--vocab-size") { if (!(v = next("--vocab-size"))) return false; o.vocab_size = atoi(v); }
        else if (a == "--consecutive") { if (!(v = next("--consecutive"))) return false; o.consecutive = atoi(v); }
        else if (a == "--discrete") {
            if (!(v = next("--discrete"))) return false;
            o.discrete.clear();
            for (const char* p = v; *p;) {
                o.discrete.push_back((int32_t)strtol(p, const_cast<char**>(&p), 10));
                if (*p == ',') ++p;
            }
        }
        else if (a == "--ratio") {
            if (!(v = next("--ratio"))) return false;
            if (sscanf(v, "%d/%d", &o.ratio_num, &o.ratio_den) != 2) { fprintf(stderr, "--ratio wants NUM/DEN\n"); return false; }
        }
        else if (a == "--max-dist") { if (!(v = next("--max-dist"))) return false; o.max_dist = atoll(v); }
        else if (a == "--no-mutual") o.mutual = false;
        else if (a == "--mat") o.write_mat = true;
        else if (a == "--verify") { if (!(v = next("--verify"))) return false; o.verify = (float)atof(v); }
        else if (a == "--samples") { if (!(v = next("--samples"))) return false; o.samples = atoi(v); }
        else if (a == "--seed") { if (!(v = next("--seed"))) return false; o.seed = (unsigned)strtoul(v, nullptr, 10); }
        else if (a == "--min-inliers") { if (!(v = next("--min-inliers"))) return false; o.min_inliers = atoi(v); }
        else if (a == "--inlier-out") { if (!(v = next("--inlier-out"))) return false; o.inlier_out = v; }
        else if (a == "--camera") {
            if (!(v = next("--camera"))) return false;
            if (sscanf(v, "%f,%f,%f,%f", &o.camera[0], &o.camera[1], &o.camera[2], &o.camera[3]) != 4 ||
                !(o.camera[0] > 0.0f) || !(o.camera[1] > 0.0f)) {
                fprintf(stderr, "--camera takes FX,FY,CX,CY with positive focal lengths\n");
                return false;
            }
            o.have_camera = true;
        }
        else if (a == "--name-ext") { if (!(v = next("--name-ext"))) return false; o.name_ext = v; }
        else if (a == "--device") { if (!(v = next("--device"))) return false; o.devices.assign(1, atoi(v)); }
        else if (a == "--devices") {
            if (!(v = next("--devices"))) return false;
            o.devices.clear();
            for (const char* p = v; *p;) {
                char* end = nullptr;
                const long d = strtol(p, &end, 10);
                if (end == p || d < 0) { fprintf(stderr, "--devices wants a list of device numbers, e.g. 0,1,2,3\n"); return false; }
                o.devices.push_back((int)d);
                p = *end == ',' ? end + 1 : end;
                if (*end && *end != ',') { fprintf(stderr, "--devices wants a list of device numbers, e.g. 0,1,2,3\n"); return false; }
            }
            if (o.devices.empty()) { fprintf(stderr, "--devices wants at least one device\n"); return false; }
        }
        else if (a == "--batch") { if (!(v = next("--batch"))) return false; o.batch = atoi(v); }
        else if (a == "--shard") {
            if (!(v = next("--shard"))) return false;
            if (sscanf(v, "%d/%d", &o.shard, &o.nshards) != 2 || o.nshards < 1 || o.shard < 0 || o.shard >= o.nshards) {
                fprintf(stderr, "--shard wants R/W with 0 <= R < W\n");
                return false;
            }
        }
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return false; }
    }
    if (o.have_camera && !(o.verify > 0.0f)) {
        fprintf(stderr, "--camera needs --verify\n");
        return false;
    }
    if (!o.inlier_out.empty() && !(o.verify > 0.0f)) {
        fprintf(stderr, "--inlier-out needs --verify\n");
        return false;
    }
    return !o.sift_dir.empty() && o.num > 0 && o.batch > 0 && o.samples > 0 && o.topk > 0 && o.vocab_size > 0 && o.kmeans_stride > 0;
}

}  // namespace

// One matching context: a GPU (or a second context on one), its copy of every image's descriptors and keypoints.
struct Device {
    int dev = 0;
    vsm_ctx* ctx = nullptr;
    std::vector<int> set_id, xy_id;
};

// The results of one submission on their way from the GPU side to the writer.
struct Batch {
    long long index = -1, lo = 0;
    int take = 0;
    bool ready = false;   // filled and waiting for the writer (guarded by the driver's mutex)
    std::vector<int32_t> idx0, idx1;
    std::vector<int64_t> off;
    // geometric verification
    std::vector<float> F;
    std::vector<int32_t> nf, in0, in1;
    std::vector<int64_t> in_off;
    // relative pose (--camera)
    std::vector<float> R, T, cams;
    std::vector<int32_t> ne;
};

int main(int argc, char** argv)
{
    Options o;
    if (!parse(argc, argv, o)) {
        usage();
        return 2;
    }
    const int ndev = (int)o.devices.size();
    std::vector<Device> devs(ndev);
    for (int d = 0; d < ndev; ++d) {
        devs[d].dev = o.devices[d];
        devs[d].ctx = vsm_create(o.devices[d]);
        if (!devs[d].ctx) {
            fprintf(stderr, "vsm_create(%d): %s\n", o.devices[d], vsm_last_error(nullptr));
            return 1;
        }
        devs[d].set_id.assign(o.num, -1);
        devs[d].xy_id.assign(o.num, -1);
    }
    vsm_ctx* ctx = devs[0].ctx;   // pair lists by content are made on the first context
    const auto t_start = std::chrono::steady_clock::now();

    // ---- load and upload every image's descriptors once: files are read by a few threads, 256 at a
    // time, and uploaded in order, every context by a thread of its own
    std::vector<int> count(o.num, 0);
    const bool verify = o.verify > 0.0f;
    // words drawn from the images: vocab_size / num rows of each (one more from the first vocab_size % num), evenly spread
    const bool draw_words = o.by_vocab && o.vocab_file.empty() && o.kmeans <= 0;
    std::vector<uint8_t> words;
    {
        const int kChunk = 256;
        const int nthreads = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        std::vector<std::vector<uint8_t>> des(kChunk);
        std::vector<std::vector<float>> loc(kChunk), xy(kChunk);
        std::vector<int> np(kChunk), bad(kChunk), nlocs(kChunk);
        for (int base = 0; base < o.num; base += kChunk) {
            const int m = std::min(kChunk, o.num - base);
            std::vector<std::thread> readers;
            for (int t = 0; t < nthreads; ++t)
                readers.emplace_back([&, t]() {
                    for (int k = t; k < m; k += nthreads) {
                        const std::string path = o.sift_dir + "/" + std::to_string(base + k) + o.sift_ext;
                        int n = 0, nloc = 0;
                        if (!vsm_sift_read(path.c_str(), nullptr, nullptr, 0, &n, &nloc)) n = 0;  // unreadable or empty
                        des[k].resize((size_t)n * VSM_DESC_DIM_U8 + 1);
                        if (verify) loc[k].resize((size_t)n * std::max(nloc, 2) + 1);
                        bad[k] = n > 0 && !vsm_sift_read(path.c_str(), verify ? loc[k].data() : nullptr, des[k].data(), n, nullptr, nullptr);
                        if (verify && n > 0 && nloc < 2) bad[k] = 1;
                        np[k] = n;
                        nlocs[k] = nloc;
                        if (verify && !bad[k]) {
                            // x, y are the first two columns of a location record
                            xy[k].resize((size_t)n * 2 + 1);
                            for (int r = 0; r < n; ++r) {
                                xy[k][2 * r] = loc[k][(size_t)r * nloc];
                                xy[k][2 * r + 1] = loc[k][(size_t)r * nloc + 1];
                            }
                        }
                    }
                });
            for (auto& r : readers) r.join();
            for (int k = 0; k < m; ++k) {
                const int i = base + k;
                if (bad[k]) {
                    fprintf(stderr, "short read: %s/%d%s\n", o.sift_dir.c_str(), i, o.sift_ext.c_str());
                    return 1;
                }
                count[i] = np[k];
                if (draw_words) {
                    const int q = std::min(np[k], o.vocab_size / o.num + (i < o.vocab_size % o.num ? 1 : 0));
                    for (int j = 0; j < q; ++j) {
                        const uint8_t* row = des[k].data() + (size_t)((long long)j * np[k] / q) * VSM_DESC_DIM_U8;
                        words.insert(words.end(), row, row + VSM_DESC_DIM_U8);
                    }
                }
            }
            std::vector<std::string> up_err(ndev);
            auto upload = [&](int d) {
                for (int k = 0; k < m; ++k) {
                    const int i = base + k;
                    devs[d].set_id[i] = vsm_upload_u8(devs[d].ctx, des[k].data(), np[k]);
                    if (devs[d].set_id[i] < 0) {
                        up_err[d] = "upload " + std::to_string(i) + o.sift_ext + ": " + vsm_last_error(devs[d].ctx);
                        return;
                    }
                    if (verify) {
                        devs[d].xy_id[i] = vsm_upload_xy(devs[d].ctx, xy[k].data(), np[k]);
                        if (devs[d].xy_id[i] < 0) {
                            up_err[d] = "upload keypoints " + std::to_string(i) + ": " + vsm_last_error(devs[d].ctx);
                            return;
                        }
                    }
                }
            };
            std::vector<std::thread> uploaders;
            for (int d = 1; d < ndev; ++d) uploaders.emplace_back(upload, d);
            upload(0);
            for (auto& u : uploaders) u.join();
            for (int d = 0; d < ndev; ++d)
                if (!up_err[d].empty()) {
                    fprintf(stderr, "%s\n", up_err[d].c_str());
                    return 1;
                }
        }
    }
    const std::vector<int>& set_id = devs[0].set_id;

    const double load_secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();

    // ---- pair list
    std::vector<int32_t> pairs;
    long long npairs;
    if (!o.pairs_file.empty()) {
        FILE* fp = fopen(o.pairs_file.c_str(), "r");
        if (!fp) {
            fprintf(stderr, "cannot open %s\n", o.pairs_file.c_str());
            return 1;
        }
        long long a = 0, b = 0;
        int got;
        while ((got = fscanf(fp, "%lld %lld", &a, &b)) == 2) {
            if (a < 0 || b < 0 || a >= o.num || b >= o.num) {
                fprintf(stderr, "%s: pair %lld %lld names an image outside 0..%d\n", o.pairs_file.c_str(), a, b, o.num - 1);
                fclose(fp);
                return 1;
            }
            pairs.push_back((int32_t)a);
            pairs.push_back((int32_t)b);
        }
        const bool clean = got == EOF && feof(fp);
        fclose(fp);
        if (!clean) {
            fprintf(stderr, "%s: not a list of \"i j\" lines\n", o.pairs_file.c_str());
            return 1;
        }
        npairs = (long long)pairs.size() / 2;
        pairs.resize((size_t)npairs * 2 + 2);
    } else if (o.by_vocab) {
        if (o.vocab_file.empty() && o.kmeans > 0) {
            words.resize((size_t)o.vocab_size * VSM_DESC_DIM_U8);
            int k;
            if (o.kmeans_stride > 1) {
                const int sample = vsm_sample_rows_u8(ctx, set_id.data(), o.num, o.kmeans_stride);
                k = sample < 0 ? sample : vsm_train_vocabulary(ctx, &sample, 1, o.vocab_size, o.kmeans, words.data());
                if (sample >= 0) vsm_release_set(ctx, sample);
            } else {
                k = vsm_train_vocabulary(ctx, set_id.data(), o.num, o.vocab_size, o.kmeans, words.data());
            }
            if (k < 0) {
                fprintf(stderr, "vsm_train_vocabulary: %s\n", vsm_last_error(ctx));
                return 1;
            }
            words.resize((size_t)k * VSM_DESC_DIM_U8);
        } else if (!draw_words) {
            int n = 0;
            if (!vsm_sift_read(o.vocab_file.c_str(), nullptr, nullptr, 0, &n, nullptr) || n <= 0) {
                fprintf(stderr, "cannot read the vocabulary %s\n", o.vocab_file.c_str());
                return 1;
            }
            words.resize((size_t)n * VSM_DESC_DIM_U8);
            if (!vsm_sift_read(o.vocab_file.c_str(), nullptr, words.data(), n, nullptr, nullptr)) {
                fprintf(stderr, "short read: %s\n", o.vocab_file.c_str());
                return 1;
            }
        }
        const int nwords = (int)(words.size() / VSM_DESC_DIM_U8);
        const int vocab_id = nwords > 0 ? vsm_upload_u8(ctx, words.data(), nwords) : VSM_EINVAL;
        if (vocab_id < 0) {
            fprintf(stderr, "vocabulary of %d words: %s\n", nwords, nwords > 0 ? vsm_last_error(ctx) : "no descriptors to draw from");
            return 1;
        }
        const long long cap = std::min<long long>((long long)o.num * std::min(o.topk, o.num), (long long)o.num * (o.num - 1) / 2);
        pairs.resize((size_t)cap * 2 + 2);
        npairs = vsm_shortlist_pairs(ctx, set_id.data(), o.num, vocab_id, o.topk, pairs.data(), cap, nullptr);
        if (npairs < 0) {
            fprintf(stderr, "vsm_shortlist_pairs: %s\n", vsm_last_error(ctx));
            return 1;
        }
        vsm_release_set(ctx, vocab_id);
        fprintf(stderr, "vocabulary of %d words, %d partners per image: %lld pairs\n", nwords, o.topk, npairs);
    } else if (o.exhaustive) {
        npairs = vsm_image_pairs_exhaustive(o.num, nullptr, 0);
        pairs.resize((size_t)npairs * 2 + 2);
        vsm_image_pairs_exhaustive(o.num, pairs.data(), npairs);
    } else {
        npairs = vsm_image_pairs(o.num, o.consecutive, o.discrete.data(), (int)o.discrete.size(), nullptr, 0);
        pairs.resize((size_t)npairs * 2 + 2);
        vsm_image_pairs(o.num, o.consecutive, o.discrete.data(), (int)o.discrete.size(), pairs.data(), (int)npairs);
    }

    // this process's slice of the pair list
    long long first_of_shard = 0;
    if (o.nshards > 1) {
        // contiguous slices of about equal work (rows of image 0 x rows of image 1 per pair), so that the
        // parts still concatenate in pair order; every process finds the same cuts
        std::vector<double> upto((size_t)npairs + 1, 0.0);
        for (long long k = 0; k < npairs; ++k)
            upto[k + 1] = upto[k] + ((double)count[pairs[2 * k]] + 1.0) * ((double)count[pairs[2 * k + 1]] + 1.0);
        auto cut = [&](int r) {
            if (r <= 0) return 0ll;
            if (r >= o.nshards) return npairs;
            const double want = upto[npairs] * r / o.nshards;
            return (long long)(std::lower_bound(upto.begin(), upto.end(), want) - upto.begin());
        };
        const long long lo = cut(o.shard), hi = std::max(lo, cut(o.shard + 1));
        first_of_shard = lo;
        pairs.erase(pairs.begin(), pairs.begin() + 2 * lo);
        pairs.resize((size_t)(hi - lo) * 2 + 2);
        npairs = hi - lo;
        if (o.out != "/dev/null") o.out += ".part" + std::to_string(o.shard);
    }

    vsm_txtwriter* tw = vsm_matchtxt_open(o.out.c_str(), o.append ? 0 : 1);
    if (!tw) {
        fprintf(stderr, "cannot open %s\n", o.out.c_str());
        return 1;
    }
    vsm_txtwriter* tw_in = nullptr;
    if (!o.inlier_out.empty()) {
        std::string path = o.inlier_out;
        if (o.nshards > 1 && path != "/dev/null") path += ".part" + std::to_string(o.shard);
        tw_in = vsm_matchtxt_open(path.c_str(), o.append ? 0 : 1);
        if (!tw_in) {
            fprintf(stderr, "cannot open %s\n", path.c_str());
            return 1;
        }
    }
    vsm_geom_params gprm{};
    gprm.max_sampson = o.verify;
    gprm.hypotheses = o.samples;
    gprm.seed = o.seed;
    gprm.min_inliers = o.min_inliers;
    vsm_params prm{};
    prm.ratio_num = o.ratio_num;
    prm.ratio_den = o.ratio_den;
    prm.mutual = o.mutual ? 1 : 0;
    prm.max_dist_sq = o.max_dist;
    prm.min_score = -1e30f;
    prm.max_dist_sq_f = -1.0f;

    // The GPU side (a thread per context) and the writer (text formatting, .mat records) overlap.  Submissions are
    // handed out in pair-list order to whichever context is free; every context keeps two of them in flight (the
    // next one is queued before the lists of the last one are collected) and has two result buffers; the one writer
    // thread takes the finished submissions in pair-list order, so the files do not depend on which GPU matched what.
    std::mutex mu;
    std::condition_variable cv;
    std::map<long long, Batch*> ready;
    std::vector<Batch> batches((size_t)ndev * 2);
    // set by whoever fails, read by everyone: the message is published before the flag (release / acquire)
    std::atomic<bool> failed{false};
    std::string fail_msg;
    auto fail = [&](const std::string& msg) {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!failed.load(std::memory_order_relaxed)) fail_msg = msg;
            failed.store(true, std::memory_order_release);
        }
        cv.notify_all();
    };
    const long long nbatches = (npairs + o.batch - 1) / o.batch;

    // One process: the writer that keeps every file's directory in memory.  Several (--shard): every record is
    // written through to the files under a lock per file, since the other processes update the same files.
    const bool shared_mat = o.write_mat && o.nshards > 1;
    if (shared_mat) vsm_mat_set_locking(1);
    vsm_matwriter* mw = o.write_mat && !shared_mat ? vsm_mat_writer_open(o.num, 0) : nullptr;
    std::thread writer([&]() {
        for (long long next = 0; next < nbatches; ++next) {
            Batch* bp = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return ready.count(next) != 0 || failed.load(std::memory_order_acquire); });
                if (failed.load(std::memory_order_acquire)) return;
                bp = ready[next];
                ready.erase(next);
            }
            Batch& b = *bp;
            for (int k = 0; k < b.take && !failed.load(std::memory_order_acquire); ++k) {
                const int ia = pairs[2 * (b.lo + k)], ib = pairs[2 * (b.lo + k) + 1];
                const int nm = (int)(b.off[k + 1] - b.off[k]);
                const std::string n0 = std::to_string(ia) + o.name_ext, n1 = std::to_string(ib) + o.name_ext;
                if (vsm_matchtxt_write(tw, n0.c_str(), n1.c_str(), b.idx0.data() + b.off[k], b.idx1.data() + b.off[k], nm,
                                       count[ia]) < 0) {
                    fail("write error on " + o.out);
                    break;
                }
                if (o.write_mat) {
                    const std::string p0 = o.sift_dir + "/" + std::to_string(ia), p1 = o.sift_dir + "/" + std::to_string(ib);
                    const int rc = shared_mat
                        ? vsm_mat_write_pmatch(p0.c_str(), p1.c_str(), count[ia], count[ib], nm, b.idx0.data() + b.off[k],
                                               b.idx1.data() + b.off[k], o.num)
                        : vsm_mat_writer_put(mw, p0.c_str(), p1.c_str(), count[ia], count[ib], nm,
                                             b.idx0.data() + b.off[k], b.idx1.data() + b.off[k]);
                    if (rc < 0) {
                        fail("cannot write .mat for pair " + std::to_string(ia) + " " + std::to_string(ib));
                        break;
                    }
                }
                if (verify) {
                    const int nin = (int)(b.in_off[k + 1] - b.in_off[k]);
                    const int32_t* q0 = b.in0.data() + b.in_off[k];
                    const int32_t* q1 = b.in1.data() + b.in_off[k];
                    if (tw_in && vsm_matchtxt_write(tw_in, n0.c_str(), n1.c_str(), q0, q1, nin, count[ia]) < 0) {
                        fail("write error on " + o.inlier_out);
                        break;
                    }
                    if (o.write_mat) {
                        const std::string p0 = o.sift_dir + "/" + std::to_string(ia), p1 = o.sift_dir + "/" + std::to_string(ib);
                        int rc;
                        if (shared_mat)
                            rc = o.have_camera
                                ? vsm_mat_write_ematch(p0.c_str(), p1.c_str(), nin, q0, q1, b.F.data() + (size_t)k * 9, b.ne[k],
                                                       b.R.data() + (size_t)k * 9, b.T.data() + (size_t)k * 3, o.camera[0], o.camera[0])
                                : vsm_mat_write_fmatch(p0.c_str(), p1.c_str(), nin, q0, q1, b.F.data() + (size_t)k * 9);
                        else
                            rc = o.have_camera
                                ? vsm_mat_writer_put_ematch(mw, p0.c_str(), p1.c_str(), nin, q0, q1, b.F.data() + (size_t)k * 9, b.ne[k],
                                                            b.R.data() + (size_t)k * 9, b.T.data() + (size_t)k * 3, o.camera[0], o.camera[0])
                                : vsm_mat_writer_put_fmatch(mw, p0.c_str(), p1.c_str(), nin, q0, q1, b.F.data() + (size_t)k * 9);
                        if (rc < 0) {
                            fail("cannot write inlier record for pair " + std::to_string(ia) + " " + std::to_string(ib));
                            break;
                        }
                    }
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                b.ready = false;
            }
            cv.notify_all();
        }
    });

    std::atomic<long long> next_batch{0};
    std::atomic<long long> total{0}, total_inliers{0}, pairs_with_geometry{0}, pairs_with_pose{0};
    auto gpu_side = [&](int d) {
        Device& D = devs[d];
        std::vector<int32_t> ids, xy_ids;
        auto grab = [&]() -> long long {
            if (failed.load(std::memory_order_acquire)) return -1;
            const long long k = next_batch.fetch_add(1);
            return k < nbatches ? k : -1;
        };
        auto submit = [&](long long k) {
            const long long lo = k * o.batch;
            const int take = (int)std::min<long long>(o.batch, npairs - lo);
            ids.resize((size_t)take * 2);
            for (int q = 0; q < take; ++q) {
                ids[2 * q] = D.set_id[pairs[2 * (lo + q)]];
                ids[2 * q + 1] = D.set_id[pairs[2 * (lo + q) + 1]];
            }
            if (vsm_submit_pairs(D.ctx, ids.data(), take, &prm, VSM_METRIC_L2) != VSM_OK) {
                fail(std::string("vsm_submit_pairs: ") + vsm_last_error(D.ctx));
                return false;
            }
            return true;
        };
        vsm_geom_params g = gprm;
        long long cur = grab();
        if (cur < 0 || !submit(cur)) return;
        for (int slot = 0; cur >= 0; slot ^= 1) {
            const long long nxt = grab();
            if (nxt >= 0 && !submit(nxt)) return;
            Batch& b = batches[(size_t)d * 2 + slot];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !b.ready || failed.load(std::memory_order_acquire); });
            }
            if (failed.load(std::memory_order_acquire)) return;
            const long long lo = cur * o.batch;
            const int take = (int)std::min<long long>(o.batch, npairs - lo);
            size_t cap = 1;
            for (int q = 0; q < take; ++q) cap += (size_t)count[pairs[2 * (lo + q)]];
            b.index = cur;
            b.lo = lo;
            b.take = take;
            b.idx0.resize(cap);
            b.idx1.resize(cap);
            b.off.resize((size_t)take + 1);
            const long long got = vsm_collect_matches(D.ctx, b.off.data(), b.idx0.data(), b.idx1.data(), (long long)cap);
            if (got < 0) {
                fail(std::string("vsm_collect_matches: ") + vsm_last_error(D.ctx));
                return;
            }
            total += got;
            if (verify) {
                // the lists just collected are still on the device: verify them there
                xy_ids.resize((size_t)take * 2);
                for (int q = 0; q < take; ++q) {
                    xy_ids[2 * q] = D.xy_id[pairs[2 * (lo + q)]];
                    xy_ids[2 * q + 1] = D.xy_id[pairs[2 * (lo + q) + 1]];
                }
                b.F.resize((size_t)take * 9);
                b.nf.resize((size_t)take);
                b.in_off.resize((size_t)take + 1);
                b.in0.resize((size_t)got + 1);
                b.in1.resize((size_t)got + 1);
                g.first_pair = first_of_shard + lo;   // global position in the pair list: sharding and batching do not change the samples
                const long long gin = vsm_verify_matches(D.ctx, xy_ids.data(), take, nullptr, nullptr, nullptr, &g, b.F.data(),
                                                         b.nf.data(), b.in_off.data(), b.in0.data(), b.in1.data(), got);
                if (gin < 0) {
                    fail(std::string("vsm_verify_matches: ") + vsm_last_error(D.ctx));
                    return;
                }
                total_inliers += gin;
                long long with = 0;
                for (int q = 0; q < take; ++q) with += b.nf[q] > 0;
                pairs_with_geometry += with;
                if (o.have_camera) {
                    b.R.resize((size_t)take * 9);
                    b.T.resize((size_t)take * 3);
                    b.ne.resize((size_t)take);
                    b.cams.resize((size_t)take * 8);
                    for (int q = 0; q < take; ++q)
                        for (int e = 0; e < 8; ++e) b.cams[(size_t)q * 8 + e] = o.camera[e & 3];
                    const int np = vsm_recover_pose(D.ctx, xy_ids.data(), take, b.cams.data(), b.F.data(), b.in_off.data(), b.in0.data(),
                                                    b.in1.data(), b.R.data(), b.T.data(), b.ne.data());
                    if (np < 0) {
                        fail(std::string("vsm_recover_pose: ") + vsm_last_error(D.ctx));
                        return;
                    }
                    pairs_with_pose += np;
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                b.ready = true;
                ready[cur] = &b;
            }
            cv.notify_all();
            cur = nxt;
        }
    };
    std::vector<std::thread> gpu_threads;
    for (int d = 1; d < ndev; ++d) gpu_threads.emplace_back(gpu_side, d);
    gpu_side(0);
    for (auto& t : gpu_threads) t.join();
    writer.join();
    if (failed.load(std::memory_order_acquire)) {
        fprintf(stderr, "%s\n", fail_msg.c_str());
        return 1;
    }
    if (mw && vsm_mat_writer_close(mw) != VSM_OK) {
        fprintf(stderr, "error writing .mat files\n");
        return 1;
    }
    if (vsm_matchtxt_close(tw) != VSM_OK) {
        fprintf(stderr, "write error on %s\n", o.out.c_str());
        return 1;
    }
    if (tw_in && vsm_matchtxt_close(tw_in) != VSM_OK) {
        fprintf(stderr, "write error on %s\n", o.inlier_out.c_str());
        return 1;
    }
    if (verify)
        fprintf(stderr, "geometric verification: %lld pairs with a fundamental matrix, %lld inlier matches\n",
                pairs_with_geometry.load(), total_inliers.load());
    if (o.have_camera) fprintf(stderr, "relative pose: %lld pairs\n", pairs_with_pose.load());
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    fprintf(stderr, "%d images, %lld pairs, %lld matches, %.3f s (%.0f pairs/s) on %d context%s; reading + upload %.3f s, matching + writing %.3f s\n",
            o.num, npairs, total.load(), secs, npairs / secs, ndev, ndev == 1 ? "" : "s", load_secs, secs - load_secs);
    for (Device& D : devs) vsm_destroy(D.ctx);
    return 0;
}
