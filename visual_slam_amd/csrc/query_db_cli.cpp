// vsm_query_db — command-line driver of the sliced map-database query: float descriptors of a database and of a
// set of queries in (NumPy .npy files, float32, [rows][dim]), the two nearest database rows of every query out.
//
// The search it runs is the brute-force nearest-neighbour search of NearestEmbedFunc.forward
// (Graph_Idea/model/model_2_0.py:43-78), with the database cut into slices over the GPUs named by --devices;
// the per-slice lists meet in an RCCL all-gather and are merged on the device (vsm_dbq_* in include/vsm.h).
#include "../../include/vsm.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// float32 C-order 2-D .npy (format 1.0 / 2.0 / 3.0)
bool read_npy(const std::string& path, std::vector<float>& data, long long& rows, int& cols)
{
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) {
        fprintf(stderr, "cannot open %s\n", path.c_str());
        return false;
    }
    unsigned char head[12];
    bool ok = fread(head, 1, 10, fp) == 10 && memcmp(head, "\x93NUMPY", 6) == 0;
    size_t hlen = 0;
    if (ok) {
        if (head[6] == 1) hlen = head[8] | (head[9] << 8);
        else ok = fread(head + 10, 1, 2, fp) == 2, hlen = head[8] | (head[9] << 8) | (head[10] << 16) | ((size_t)head[11] << 24);
    }
    std::string dict(hlen, '\0');
    ok = ok && hlen > 0 && hlen < (1u << 20) && fread(&dict[0], 1, hlen, fp) == hlen;
    long long r = 0, c = 0;
    if (ok) {
        const size_t sh = dict.find("'shape'");
        ok = dict.find("'<f4'") != std::string::npos && dict.find("'fortran_order': False") != std::string::npos &&
             sh != std::string::npos && sscanf(dict.c_str() + sh, "'shape': (%lld, %lld)", &r, &c) == 2 && r >= 0 && c >= 1 && c <= 256;
    }
    if (!ok) {
        fprintf(stderr, "%s is not a float32 C-order .npy file of shape (rows, dim <= 256)\n", path.c_str());
        fclose(fp);
        return false;
    }
    data.resize((size_t)r * c);
    if (fread(data.data(), sizeof(float), data.size(), fp) != data.size()) {
        fprintf(stderr, "short read: %s\n", path.c_str());
        fclose(fp);
        return false;
    }
    fclose(fp);
    rows = r;
    cols = (int)c;
    return true;
}

// unit rows from a counter-based generator (splitmix64 -> Box-Muller): the same on every machine
void synth_rows(std::vector<float>& out, long long n, int dim, unsigned long long seed)
{
    out.resize((size_t)n * dim);
    auto mix = [](unsigned long long z) {
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    for (long long i = 0; i < n; ++i) {
        double nrm = 0;
        for (int k = 0; k < dim; ++k) {
            const unsigned long long a = mix(seed + ((unsigned long long)i * dim + k) * 2), b = mix(seed + ((unsigned long long)i * dim + k) * 2 + 1);
            const double u1 = ((a >> 11) + 1.0) / 9007199254740993.0, u2 = (b >> 11) / 9007199254740992.0;
            const double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
            out[(size_t)i * dim + k] = (float)g;
            nrm += g * g;
        }
        const float inv = (float)(1.0 / sqrt(nrm > 0 ? nrm : 1.0));
        for (int k = 0; k < dim; ++k) out[(size_t)i * dim + k] *= inv;
    }
}

void usage()
{
    fprintf(stderr,
            "usage: vsm_query_db --db DB.npy --queries Q.npy --out FILE [options]\n"
            "       vsm_query_db --synthetic NQ,NDB,DIM [--reps R] [options]\n"
            "  --devices A,B,...   GPUs that hold the database (default 0); --slices S (a multiple of their number;\n"
            "                      default: one per device)\n"
            "  --metric l2|dot     squared L2 distance (default) or dot product\n"
            "  FILE gets one line per query: row and key of the nearest database row, then of the runner-up\n");
}

}  // namespace

int main(int argc, char** argv)
{
    std::string db_path, q_path, out_path;
    std::vector<int> devices{0};
    int slices = 0, metric = VSM_METRIC_L2, reps = 5;
    long long syn_q = 0, syn_db = 0;
    int syn_dim = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        const char* v = i + 1 < argc ? argv[i + 1] : nullptr;
        auto need = [&]() { if (!v) { usage(); exit(2); } ++i; return v; };
        if (a == "--db") db_path = need();
        else if (a == "--queries") q_path = need();
        else if (a == "--out") out_path = need();
        else if (a == "--slices") slices = atoi(need());
        else if (a == "--reps") reps = atoi(need());
        else if (a == "--metric") { const std::string m = need(); metric = m == "dot" ? VSM_METRIC_DOT : VSM_METRIC_L2; }
        else if (a == "--devices") {
            devices.clear();
            for (const char* p = need(); *p;) {
                char* end = nullptr;
                devices.push_back((int)strtol(p, &end, 10));
                if (end == p) { usage(); return 2; }
                p = *end == ',' ? end + 1 : end;
            }
        }
        else if (a == "--synthetic") {
            if (sscanf(need(), "%lld,%lld,%d", &syn_q, &syn_db, &syn_dim) != 3 || syn_q < 1 || syn_db < 1 || syn_dim < 1 || syn_dim > 256) { usage(); return 2; }
        }
        else { usage(); return 2; }
    }
    const bool synthetic = syn_dim > 0;
    if (devices.empty() || (!synthetic && (db_path.empty() || q_path.empty() || out_path.empty()))) {
        usage();
        return 2;
    }
    if (slices <= 0) slices = (int)devices.size();
    std::vector<float> db, qs;
    long long ndb = 0, nq = 0;
    int dim = 0, qdim = 0;
    if (synthetic) {
        ndb = syn_db; nq = syn_q; dim = qdim = syn_dim;
        synth_rows(db, ndb, dim, 1000);
        synth_rows(qs, nq, dim, 99);
    } else if (!read_npy(db_path, db, ndb, dim) || !read_npy(q_path, qs, nq, qdim)) {
        return 1;
    }
    if (dim != qdim || nq > INT32_MAX) {
        fprintf(stderr, "database rows have %d components, queries %d\n", dim, qdim);
        return 1;
    }
    vsm_dbq* h = vsm_dbq_create(devices.data(), (int)devices.size(), slices);
    if (!h) {
        fprintf(stderr, "vsm_dbq_create: %s\n", vsm_dbq_last_error(nullptr));
        return 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (vsm_dbq_load(h, db.data(), ndb, dim, VSM_LAYOUT_ROWS) != VSM_OK) {
        fprintf(stderr, "vsm_dbq_load: %s\n", vsm_dbq_last_error(h));
        return 1;
    }
    const double load_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<float> k1((size_t)nq), k2((size_t)nq);
    std::vector<int64_t> i1((size_t)nq), i2((size_t)nq);
    const int runs = synthetic ? std::max(reps, 1) + 1 : 1;
    double best = 1e30;
    for (int r = 0; r < runs; ++r) {
        const auto t1 = std::chrono::steady_clock::now();
        if (vsm_dbq_query(h, qs.data(), (int)nq, VSM_LAYOUT_ROWS, metric, k1.data(), k2.data(), i1.data(), i2.data()) < 0) {
            fprintf(stderr, "vsm_dbq_query: %s\n", vsm_dbq_last_error(h));
            return 1;
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        if (r > 0 || runs == 1) best = std::min(best, s);
    }
    if (!out_path.empty()) {
        FILE* fp = fopen(out_path.c_str(), "w");
        if (!fp) {
            fprintf(stderr, "cannot open %s\n", out_path.c_str());
            return 1;
        }
        for (long long q = 0; q < nq; ++q)
            fprintf(fp, "%lld %.9g %lld %.9g\n", (long long)i1[q], (double)k1[q], (long long)i2[q], (double)k2[q]);
        if (fclose(fp) != 0) {
            fprintf(stderr, "write error on %s\n", out_path.c_str());
            return 1;
        }
    }
    fprintf(stderr, "%lld queries x %lld rows x %d, %d slice%s on %zu device%s: upload %.3f s, query %.3f ms (%.1f G evals/s, "
                    "queries uploaded and lists fetched inside)\n",
            nq, ndb, dim, slices, slices == 1 ? "" : "s", devices.size(), devices.size() == 1 ? "" : "s", load_s, best * 1e3,
            (double)nq * (double)ndb / best / 1e9);
    vsm_dbq_destroy(h);
    return 0;
}
