// VisualSfM interchange files: .sift feature files, the three-line-per-pair match txt, the
// keypoint txt, and .mat putative-match records.  Host code only.
//
// Layouts follow the reference's readers/writers:
//   .sift      Others/visualSfM_Sift_file/Match_modified/FeaturePoints.{h,cpp}
//   .mat       Others/visualSfM_Sift_file/Match_modified/MatchFile.{h,cpp}
//   match txt  Others/SuperPoint_data/main.py:114-155
//   keypoints  Others/SuperPoint_data/main.py:133-140, FeaturePoints.cpp:326-369
#include "../../include/vsm.h"

#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

// ---- small I/O helpers --------------------------------------------------------------
class Fd {
public:
    explicit Fd(int fd = -1) : fd_(fd) {}
    ~Fd() { if (fd_ >= 0) ::close(fd_); }
    Fd(const Fd&) = delete;
    Fd& operator=(const Fd&) = delete;
    bool ok() const { return fd_ >= 0; }
    int get() const { return fd_; }
    bool write_all(const void* p, size_t n) const
    {
        const char* c = static_cast<const char*>(p);
        while (n) {
            ssize_t w = ::write(fd_, c, n);
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) return false;
            c += w;
            n -= (size_t)w;
        }
        return true;
    }
    // returns bytes read (short at EOF) or -1
    ssize_t read_upto(void* p, size_t n) const
    {
        char* c = static_cast<char*>(p);
        size_t got = 0;
        while (got < n) {
            ssize_t r = ::read(fd_, c + got, n - got);
            if (r < 0 && errno == EINTR) continue;
            if (r < 0) return -1;
            if (r == 0) break;
            got += (size_t)r;
        }
        return (ssize_t)got;
    }
    bool read_exact(void* p, size_t n) const { return read_upto(p, n) == (ssize_t)n; }
    bool seek(off_t o) const { return ::lseek(fd_, o, SEEK_SET) >= 0; }
private:
    int fd_;
};

constexpr int32_t fourcc(char a, char b, char c, char d)
{
    return (int32_t)((uint32_t)(uint8_t)a | ((uint32_t)(uint8_t)b << 8) | ((uint32_t)(uint8_t)c << 16) |
                     ((uint32_t)(uint8_t)d << 24));
}

// ---- .sift (FeaturePoints.h:55-90) ----------------------------------------------------
constexpr int32_t kSiftName = fourcc('S', 'I', 'F', 'T');
constexpr int32_t kMserName = fourcc('M', 'S', 'E', 'R');
constexpr int32_t kSiftV4 = fourcc('V', '4', '.', '0');
constexpr int32_t kSiftV5 = fourcc('V', '5', '.', '0');
constexpr int32_t kSiftEof = fourcc('\xff', 'E', 'O', 'F');

struct SiftHeader {
    int32_t name, version, npoint, nloc_dim, ndes_dim;
};

// ---- decimal formatting -----------------------------------------------------------
inline char* put_int(char* p, int v)
{
    char tmp[12];
    int n = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do {
        tmp[n++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    if (v < 0) *p++ = '-';
    while (n) *p++ = tmp[--n];
    return p;
}

bool append_file(const char* path, const std::string& bytes)
{
    // main.py:143-155: append if the file exists, else create
    Fd f(::open(path, O_CREAT | O_WRONLY | O_APPEND, 0644));
    return f.ok() && f.write_all(bytes.data(), bytes.size());
}

}  // namespace

// =====================================================================================
// .sift
// =====================================================================================
extern "C" int vsm_sift_read(const char* path, float* locbuf, uint8_t* desbuf, int nmax,
                             int* npoint, int* nloc_dim)
{
    if (npoint) *npoint = 0;
    if (nloc_dim) *nloc_dim = 0;
    if (!path) return 0;
    Fd f(::open(path, O_RDONLY));
    if (!f.ok()) return 0;
    SiftHeader h;
    if (!f.read_exact(&h, sizeof h)) return 0;
    const bool name_ok = h.name == kSiftName || h.name == kMserName;
    const bool ver_ok = h.version == kSiftV4 || h.version == kSiftV5;
    // FeaturePoints.cpp:396-402: the reader accepts only 128-d descriptors
    if (!name_ok || !ver_ok || h.npoint <= 0 || h.nloc_dim <= 0 || h.ndes_dim != VSM_DESC_DIM_U8) return 0;
    if (npoint) *npoint = h.npoint;
    if (nloc_dim) *nloc_dim = h.nloc_dim;
    if (!locbuf && !desbuf) return h.npoint;
    const int take = std::min(std::max(nmax, 0), h.npoint);
    const size_t loc_bytes = sizeof(float) * (size_t)h.npoint * h.nloc_dim;
    if (locbuf) {
        // the location block is stored for all points; hand back the first `take`
        const size_t want = sizeof(float) * (size_t)take * h.nloc_dim;
        if (!f.read_exact(locbuf, want)) return 0;
    }
    if (desbuf) {
        if (!f.seek((off_t)(sizeof h + loc_bytes))) return 0;
        if (!f.read_exact(desbuf, (size_t)take * VSM_DESC_DIM_U8)) return 0;
    }
    return take;   // FeaturePoints.cpp:524-529: the rows now in the caller's buffers, min(nmax, npoint)
}

extern "C" int vsm_sift_write(const char* path, const float* loc, int nloc_dim,
                              const uint8_t* des, int npoint)
{
    if (!path || npoint < 0 || nloc_dim <= 0 || (npoint > 0 && (!loc || !des))) return 0;
    // FeaturePoints.cpp:233: created owner read/write
    Fd f(::open(path, O_CREAT | O_WRONLY | O_TRUNC, S_IRUSR | S_IWUSR));
    if (!f.ok()) return 0;
    const SiftHeader h = {kSiftName, kSiftV4, npoint, nloc_dim, VSM_DESC_DIM_U8};
    const int32_t eof = kSiftEof;
    bool ok = f.write_all(&h, sizeof h);
    ok = ok && f.write_all(loc, sizeof(float) * (size_t)npoint * nloc_dim);
    ok = ok && f.write_all(des, (size_t)npoint * VSM_DESC_DIM_U8);
    ok = ok && f.write_all(&eof, sizeof eof);
    return ok ? 1 : 0;
}

// ---- ASCII .sift (Lowe's keypoint text; FeaturePoints.cpp:54-125, 205-226) ---------------------------
namespace {

// Whitespace-separated tokens of a whole file held in memory.
class Tokens {
public:
    explicit Tokens(std::string text) : text_(std::move(text)), p_(text_.c_str()) {}
    bool next_float(float* v)
    {
        char* end = nullptr;
        skip();
        if (!*p_) return false;
        *v = strtof(p_, &end);
        if (end == p_) return false;
        p_ = end;
        return true;
    }
    bool next_uint(unsigned* v)
    {
        char* end = nullptr;
        skip();
        if (*p_ < '0' || *p_ > '9') return false;
        *v = (unsigned)strtoul(p_, &end, 10);
        p_ = end;
        return true;
    }
    bool next_int(int* v)
    {
        char* end = nullptr;
        skip();
        if (!*p_) return false;
        const long x = strtol(p_, &end, 10);
        if (end == p_) return false;
        *v = (int)x;
        p_ = end;
        return true;
    }
private:
    void skip() { while (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r' || *p_ == '\f' || *p_ == '\v') ++p_; }
    std::string text_;
    const char* p_;
};

bool slurp(const char* path, std::string* out)
{
    Fd f(::open(path, O_RDONLY));
    if (!f.ok()) return false;
    struct stat st;
    if (fstat(f.get(), &st) != 0 || st.st_size < 0) return false;
    out->resize((size_t)st.st_size);
    return f.read_upto(&(*out)[0], out->size()) == (ssize_t)out->size();
}

}  // namespace

// FeatureData::ReadSIFTA + ReadFeatureDataA with four location values per feature (FeaturePoints.cpp:54-125):
// "npoint ndim", then per feature "row col scale orientation" and ndim descriptor integers.  A stored
// location is (col + 0.5, row + 0.5, 0, scale, -orientation); a descriptor value keeps its low 8 bits.
extern "C" int vsm_sift_read_ascii(const char* path, float* locbuf, uint8_t* desbuf, int nmax, int* npoint)
{
    if (npoint) *npoint = 0;
    std::string text;
    if (!path || !slurp(path, &text)) return 0;
    const size_t file_bytes = text.size();
    Tokens tok(std::move(text));
    int n = 0, nd = 0;
    if (!tok.next_int(&n) || !tok.next_int(&nd) || n <= 0 || nd != VSM_DESC_DIM_U8) return 0;
    // a feature is 4 + 128 values, each at least one character and a separator: a count the file cannot hold is a
    // malformed header (and would otherwise size the caller's buffers)
    if ((size_t)n > file_bytes / (2 * (4 + VSM_DESC_DIM_U8))) return 0;
    if (npoint) *npoint = n;
    if (!locbuf && !desbuf) return n;
    const int take = std::min(std::max(nmax, 0), n);
    for (int i = 0; i < take; ++i) {
        float row, col, scale, ori;
        if (!tok.next_float(&row) || !tok.next_float(&col) || !tok.next_float(&scale) || !tok.next_float(&ori)) return 0;
        if (locbuf) {
            float* l = locbuf + (size_t)i * 5;
            l[0] = col + 0.5f;
            l[1] = row + 0.5f;
            l[2] = 0.0f;
            l[3] = scale;
            l[4] = -ori;
        }
        for (int k = 0; k < VSM_DESC_DIM_U8; ++k) {
            unsigned v;
            if (!tok.next_uint(&v)) return 0;
            if (desbuf) desbuf[(size_t)i * VSM_DESC_DIM_U8 + k] = (uint8_t)v;
        }
    }
    return n;
}

// FeatureData::saveSIFTA (FeaturePoints.cpp:205-226): "n 128", then per feature "loc[1] loc[0] loc[3] loc[4]"
// (as `ofstream << float` prints them) and the descriptor, every value followed by a blank, a line break after
// every 20th value and one more after the last.  Nothing is written for an empty set, as there.
extern "C" int vsm_sift_write_ascii(const char* path, const float* loc, const uint8_t* des, int npoint)
{
    if (!path || npoint < 0 || (npoint > 0 && (!loc || !des))) return VSM_EINVAL;
    if (npoint == 0) return 0;
    std::string out;
    out.reserve((size_t)npoint * 600 + 32);
    char num[64];
    char* p = put_int(num, npoint);
    *p++ = ' ';
    p = put_int(p, VSM_DESC_DIM_U8);
    *p++ = '\n';
    out.append(num, p - num);
    for (int i = 0; i < npoint; ++i) {
        const float* l = loc + (size_t)i * 5;
        const int len = snprintf(num, sizeof num, "%g %g %g %g\n", (double)l[1], (double)l[0], (double)l[3], (double)l[4]);
        out.append(num, len);
        for (int k = 0; k < VSM_DESC_DIM_U8; ++k) {
            p = put_int(num, des[(size_t)i * VSM_DESC_DIM_U8 + k]);
            *p++ = ' ';
            if ((k + 1) % 20 == 0) *p++ = '\n';
            out.append(num, p - num);
        }
        out.push_back('\n');
    }
    Fd f(::open(path, O_CREAT | O_WRONLY | O_TRUNC, 0644));
    if (!f.ok() || !f.write_all(out.data(), out.size())) return VSM_EIO;
    return npoint;
}

// FeatureData::SortSIFT (FeaturePoints.cpp:726-765): features by descending scale (location column 3).  The
// reference bubble-sorts the REVERSED sequence with a strict comparison, so features of equal scale end up
// in descending order of their old position; a stable sort of the reversed positions gives the same order.
extern "C" int vsm_sift_sort_by_scale(float* loc, int nloc_dim, uint8_t* des, int npoint, int32_t* order)
{
    if (npoint < 0 || nloc_dim < 4 || (npoint > 0 && !loc)) return VSM_EINVAL;
    std::vector<int32_t> ord((size_t)npoint);
    for (int i = 0; i < npoint; ++i) ord[i] = npoint - 1 - i;
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
        return loc[(size_t)a * nloc_dim + 3] > loc[(size_t)b * nloc_dim + 3];
    });
    std::vector<float> l2((size_t)npoint * nloc_dim);
    for (int i = 0; i < npoint; ++i)
        memcpy(&l2[(size_t)i * nloc_dim], loc + (size_t)ord[i] * nloc_dim, sizeof(float) * nloc_dim);
    if (npoint) memcpy(loc, l2.data(), l2.size() * sizeof(float));
    if (des) {
        std::vector<uint8_t> d2((size_t)npoint * VSM_DESC_DIM_U8);
        for (int i = 0; i < npoint; ++i)
            memcpy(&d2[(size_t)i * VSM_DESC_DIM_U8], des + (size_t)ord[i] * VSM_DESC_DIM_U8, VSM_DESC_DIM_U8);
        if (npoint) memcpy(des, d2.data(), d2.size());
    }
    if (order && npoint) memcpy(order, ord.data(), sizeof(int32_t) * (size_t)npoint);
    return npoint;
}

// The in-memory part of FeatureData::ConvertA2B (FeaturePoints.cpp:655-695): when the FIRST descriptor's L2 norm
// is below 400 the set is taken to be in another convention: every descriptor is scaled to norm 512 and rounded
// (floor(x + 0.5), in double), and the first two location values of every feature change places.
extern "C" int vsm_sift_renormalise(float* loc, int nloc_dim, uint8_t* des, int npoint)
{
    if (npoint < 0 || (npoint > 0 && !des) || (loc && nloc_dim < 2)) return VSM_EINVAL;
    if (npoint == 0) return 0;
    double first = 0;
    for (int k = 0; k < VSM_DESC_DIM_U8; ++k) first += (double)(des[k] * des[k]);
    if (!(sqrt(first) < 400)) return 0;
    for (int i = 0; i < npoint; ++i) {
        uint8_t* d = des + (size_t)i * VSM_DESC_DIM_U8;
        double sum = 0;
        for (int k = 0; k < VSM_DESC_DIM_U8; ++k) sum += (double)(d[k] * d[k]);
        if (sum == 0) continue;   // the reference divides by zero here; an all-zero row stays as it is
        const double scale = 512.0 / sqrt(sum);
        for (int k = 0; k < VSM_DESC_DIM_U8; ++k) {
            const double v = floor(scale * d[k] + 0.5);
            d[k] = (uint8_t)(unsigned)v;          // 512 at most: the low 8 bits, as the reference's cast keeps
        }
    }
    if (loc)
        for (int i = 0; i < npoint; ++i) std::swap(loc[(size_t)i * nloc_dim], loc[(size_t)i * nloc_dim + 1]);
    return 1;
}

// ReadSIFTA followed by ConvertA2B (FeaturePoints.cpp:54-72, 655-710): the binary file takes the ASCII file's
// place; "x.sift.sift" becomes "x.sift" and the ASCII file is removed, "x.sift" is overwritten, any other name
// is left alone (nothing written, as there).  out_path receives the name written.
extern "C" int vsm_sift_ascii_to_binary(const char* path, char* out_path, int cap)
{
    if (out_path && cap > 0) out_path[0] = 0;
    if (!path) return VSM_EINVAL;
    int n = 0;
    if (!vsm_sift_read_ascii(path, nullptr, nullptr, 0, &n)) return VSM_EFORMAT;
    std::vector<float> loc;
    std::vector<uint8_t> des;
    try {   // nothing may be thrown through the C boundary
        loc.resize((size_t)n * 5);
        des.resize((size_t)n * VSM_DESC_DIM_U8);
    } catch (const std::bad_alloc&) {
        return VSM_ENOMEM;
    }
    if (!vsm_sift_read_ascii(path, loc.data(), des.data(), n, nullptr)) return VSM_EFORMAT;
    vsm_sift_renormalise(loc.data(), 5, des.data(), n);
    std::string name(path);
    const size_t len = name.size();
    bool remove_ascii = false;
    if (len > 10 && name.compare(len - 10, 10, ".sift.sift") == 0) {
        name.resize(len - 5);
        remove_ascii = true;
    } else if (!(len > 5 && name.compare(len - 5, 5, ".sift") == 0)) {
        return n;
    }
    if (!vsm_sift_write(name.c_str(), loc.data(), 5, des.data(), n)) return VSM_EIO;
    if (remove_ascii) ::remove(path);
    if (out_path && cap > 0) snprintf(out_path, (size_t)cap, "%s", name.c_str());
    return n;
}

// =====================================================================================
// keypoint txt
// =====================================================================================
extern "C" int vsm_feature_txt_write(const char* path, const float* xy, int m)
{
    if (!path || m < 0 || (m > 0 && !xy)) return VSM_EINVAL;
    // main.py:133-140: row 0 = "m 0", then "int(x) int(y)" (truncated toward zero)
    std::string out;
    out.reserve(16 * ((size_t)m + 1));
    char line[40];
    char* p = put_int(line, m);
    *p++ = ' '; *p++ = '0'; *p++ = '\n';
    out.append(line, p - line);
    for (int i = 0; i < m; ++i) {
        p = put_int(line, (int)xy[2 * i]);
        *p++ = ' ';
        p = put_int(p, (int)xy[2 * i + 1]);
        *p++ = '\n';
        out.append(line, p - line);
    }
    Fd f(::open(path, O_CREAT | O_WRONLY | O_TRUNC, 0644));
    if (!f.ok() || !f.write_all(out.data(), out.size())) return VSM_EIO;
    return m;
}

extern "C" int vsm_feature_txt_read(const char* path, float* xy, int cap)
{
    if (!path) return VSM_EINVAL;
    FILE* fp = fopen(path, "r");
    if (!fp) return VSM_EIO;
    int m = 0, tmp = 0;
    if (fscanf(fp, "%d %d", &m, &tmp) != 2 || m < 0) {
        fclose(fp);
        return VSM_EFORMAT;
    }
    // FeaturePoints.cpp:336-355: x, y per row
    int got = 0;
    for (; got < m; ++got) {
        float x, y;
        if (fscanf(fp, "%f %f", &x, &y) != 2) break;
        if (xy && got < cap) {
            xy[2 * got] = x;
            xy[2 * got + 1] = y;
        }
    }
    fclose(fp);
    if (got != m) return VSM_EFORMAT;
    return m;
}

// FeatureData::Sift2Txt (FeaturePoints.cpp:306-325) and write_out_data (:281-305): the features of a
// .sift file as text, one row per feature, every value followed by a blank.  Locations are written
// the way `ofstream << float` writes them (six significant digits, %g), descriptor bytes as integers.
extern "C" int vsm_sift_to_txt(const char* sift_path, const char* txt_path, int what)
{
    if (!sift_path || !txt_path || what < VSM_TXT_BOTH || what > VSM_TXT_DESCRIPTORS) return VSM_EINVAL;
    int n = 0, nloc = 0;
    if (!vsm_sift_read(sift_path, nullptr, nullptr, 0, &n, &nloc)) return VSM_EFORMAT;
    std::vector<float> loc((size_t)n * nloc + 1);
    std::vector<uint8_t> des((size_t)n * VSM_DESC_DIM_U8 + 1);
    if (n > 0 && !vsm_sift_read(sift_path, loc.data(), des.data(), n, nullptr, nullptr)) return VSM_EFORMAT;
    std::string out;
    out.reserve((size_t)n * (what == VSM_TXT_LOCATIONS ? 64 : 600));
    char num[48];
    for (int i = 0; i < n; ++i) {
        if (what != VSM_TXT_DESCRIPTORS)
            for (int j = 0; j < nloc; ++j) {
                const int len = snprintf(num, sizeof num, "%g ", (double)loc[(size_t)i * nloc + j]);
                out.append(num, len);
            }
        if (what != VSM_TXT_LOCATIONS)
            for (int j = 0; j < VSM_DESC_DIM_U8; ++j) {
                char* p = put_int(num, des[(size_t)i * VSM_DESC_DIM_U8 + j]);
                *p++ = ' ';
                out.append(num, p - num);
            }
        out.push_back('\n');
    }
    Fd f(::open(txt_path, O_CREAT | O_WRONLY | O_TRUNC, 0644));
    if (!f.ok() || !f.write_all(out.data(), out.size())) return VSM_EIO;
    return n;
}

// txt2sift of test_main.cpp:46-60: ReadLocFromText (FeaturePoints.cpp:326-369: locations x, y, 0, 0, 0
// and descriptors of all ones) followed by saveSIFTB2.
extern "C" int vsm_txt_to_sift(const char* txt_path, const char* sift_path)
{
    if (!txt_path || !sift_path) return VSM_EINVAL;
    const int m = vsm_feature_txt_read(txt_path, nullptr, 0);
    if (m < 0) return m;
    std::vector<float> xy((size_t)m * 2 + 1), loc((size_t)m * 5 + 1, 0.0f);
    if (m > 0 && vsm_feature_txt_read(txt_path, xy.data(), m) != m) return VSM_EFORMAT;
    for (int i = 0; i < m; ++i) {
        loc[(size_t)i * 5] = xy[2 * i];
        loc[(size_t)i * 5 + 1] = xy[2 * i + 1];
    }
    const std::vector<uint8_t> des((size_t)m * VSM_DESC_DIM_U8 + 1, 1);
    return vsm_sift_write(sift_path, loc.data(), 5, des.data(), m) ? m : VSM_EIO;
}

// =====================================================================================
// match txt
// =====================================================================================
// "00".."99": two digits per table look-up
struct DigitPairs {
    char d[200];
    DigitPairs()
    {
        for (int i = 0; i < 100; ++i) {
            d[2 * i] = (char)('0' + i / 10);
            d[2 * i + 1] = (char)('0' + i % 10);
        }
    }
};
static const DigitPairs kDigitPairs;

// keypoint indices are small non-negative numbers: at most six digits on the short path
static inline char* put_index(char* p, int32_t v)
{
    if (v < 0 || v >= 1000000) return put_int(p, v);
    const unsigned u = (unsigned)v;
    const char* t = kDigitPairs.d;
    if (u < 100) {
        if (u >= 10) *p++ = t[2 * u];
        *p++ = t[2 * u + 1];
        return p;
    }
    if (u < 10000) {
        const unsigned hi = u / 100, lo = u % 100;
        if (hi >= 10) *p++ = t[2 * hi];
        *p++ = t[2 * hi + 1];
        *p++ = t[2 * lo];
        *p++ = t[2 * lo + 1];
        return p;
    }
    const unsigned top = u / 10000, rest = u % 10000;
    const unsigned hi = rest / 100, lo = rest % 100;
    if (top >= 10) *p++ = t[2 * top];
    *p++ = t[2 * top + 1];
    *p++ = t[2 * hi];
    *p++ = t[2 * hi + 1];
    *p++ = t[2 * lo];
    *p++ = t[2 * lo + 1];
    return p;
}

static void matchtxt_format(std::string& rec, const char* name0, const char* name1,
                            const int32_t* idx0, const int32_t* idx1, int nm, int n0)
{
    const size_t l0 = strlen(name0), l1 = strlen(name1);
    const size_t start = rec.size();
    // worst case: every entry 11 characters + separator, three line ends, the header
    rec.resize(start + l0 + l1 + 16 + 2 * ((size_t)nm * 12 + 1));
    char* const base = &rec[start];
    char* p = base;
    memcpy(p, name0, l0); p += l0;
    *p++ = ' ';
    memcpy(p, name1, l1); p += l1;
    *p++ = ' ';
    p = put_int(p, nm);
    *p++ = '\n';
    for (int line = 0; line < 2; ++line) {
        const int32_t* v = line ? idx1 : idx0;
        for (int k = 0; k < nm; ++k) {
            p = put_index(p, v[k]);
            // main.py:124-129: no separator after the entry of the last keypoint
            if (idx0[k] != n0 - 1) *p++ = ' ';
        }
        *p++ = '\n';
    }
    rec.resize(start + (size_t)(p - base));
}

// Buffered writer for long runs: one open file, records appended in call order, flushed in
// large writes.  Same bytes as repeated vsm_matchtxt_append calls.
struct vsm_txtwriter {
    int fd = -1;
    std::string buf;
    bool failed = false;
    bool flush()
    {
        size_t off = 0;
        while (off < buf.size()) {
            ssize_t w = ::write(fd, buf.data() + off, buf.size() - off);
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) { failed = true; return false; }
            off += (size_t)w;
        }
        buf.clear();
        return true;
    }
};

extern "C" vsm_txtwriter* vsm_matchtxt_open(const char* path, int truncate)
{
    if (!path) return nullptr;
    const int fd = ::open(path, O_CREAT | O_WRONLY | (truncate ? O_TRUNC : O_APPEND), 0644);
    if (fd < 0) return nullptr;
    vsm_txtwriter* w = new vsm_txtwriter();
    w->fd = fd;
    w->buf.reserve(8u << 20);
    return w;
}

extern "C" int vsm_matchtxt_write(vsm_txtwriter* w, const char* name0, const char* name1,
                                  const int32_t* idx0, const int32_t* idx1, int nm, int n0)
{
    if (!w || !name0 || !name1 || nm < 0 || (nm > 0 && (!idx0 || !idx1))) return VSM_EINVAL;
    if (w->failed) return VSM_EIO;
    matchtxt_format(w->buf, name0, name1, idx0, idx1, nm, n0);
    if (w->buf.size() >= (6u << 20) && !w->flush()) return VSM_EIO;
    return nm;
}

extern "C" int vsm_matchtxt_close(vsm_txtwriter* w)
{
    if (!w) return VSM_EINVAL;
    const bool ok = !w->failed && w->flush();
    ::close(w->fd);
    delete w;
    return ok ? VSM_OK : VSM_EIO;
}

static int matchtxt_emit(const char* path, const char* name0, const char* name1,
                         const int32_t* idx0, const int32_t* idx1, int nm, int n0)
{
    std::string rec;
    rec.reserve(64 + (size_t)nm * 14);
    matchtxt_format(rec, name0, name1, idx0, idx1, nm, n0);
    return append_file(path, rec) ? nm : VSM_EIO;
}

extern "C" int vsm_matchtxt_append(const char* path, const char* name0, const char* name1,
                                   const int32_t* idx0, const int32_t* idx1, int nm, int n0)
{
    if (!path || !name0 || !name1 || nm < 0 || (nm > 0 && (!idx0 || !idx1))) return VSM_EINVAL;
    return matchtxt_emit(path, name0, name1, idx0, idx1, nm, n0);
}

extern "C" int vsm_matchtxt_append_dense(const char* path, const char* name0, const char* name1,
                                         const int32_t* matches0, int n)
{
    if (!path || !name0 || !name1 || n < 0 || (n > 0 && !matches0)) return VSM_EINVAL;
    std::vector<int32_t> i0, i1;
    for (int i = 0; i < n; ++i)
        if (matches0[i] != -1) {
            i0.push_back(i);
            i1.push_back(matches0[i]);
        }
    return matchtxt_emit(path, name0, name1, i0.data(), i1.data(), (int)i0.size(), n);
}

// =====================================================================================
// pair lists
// =====================================================================================
extern "C" int vsm_image_pairs(int nimages, int consecutive, const int32_t* discrete, int ndiscrete,
                               int32_t* pairs, int cap)
{
    if (nimages < 0 || consecutive < 0 || ndiscrete < 0 || (ndiscrete > 0 && !discrete)) return VSM_EINVAL;
    int count = 0;
    auto emit = [&](int a, int b) {
        if (pairs && count < cap) {
            pairs[2 * count] = a;
            pairs[2 * count + 1] = b;
        }
        ++count;
    };
    for (int i = 0; i < nimages; ++i) {
        for (int step = 1; step <= consecutive; ++step)
            if (i + step < nimages) emit(i, i + step);
        for (int d = 0; d < ndiscrete; ++d)
            if (i + discrete[d] < nimages) emit(i, i + discrete[d]);
    }
    return count;
}

extern "C" long long vsm_image_pairs_exhaustive(int nimages, int32_t* pairs, long long cap)
{
    if (nimages < 0) return VSM_EINVAL;
    long long count = 0;
    for (int i = 0; i < nimages; ++i)
        for (int j = i + 1; j < nimages; ++j) {
            if (pairs && count < cap) {
                pairs[2 * count] = i;
                pairs[2 * count + 1] = j;
            }
            ++count;
        }
    return count;
}

// =====================================================================================
// .mat
// =====================================================================================
namespace {

constexpr int32_t kMatVersion = fourcc('M', 'T', '0', '3');
constexpr int32_t kRecVersion = fourcc('M', 'R', 'V', '3');
constexpr int kMaxPath = 260;  // MAX_PATH of the reference build

struct MatHeader {  // MatchFile.h:189-196
    int32_t version, file_count, definition_size, definition_buf, feature_count;
};
struct MatRecLoc {  // MatchFile.h:198-206 (file_name continues for extra_size bytes)
    int32_t fcount, read_loc, block_size, trash_size, extra_size;
    char file_name[4];
};
struct MatGeometry {  // TwoViewGeometry, MatchFile.h:68-83
    int32_t NF, NE, NH, NH2;
    float F[3][3], R[3][3], T[3], F1, F2, H[3][3], GE, AA;
};
struct MatRecord {  // MatchRecordV3, MatchFile.h:174-184
    int32_t version;
    MatGeometry tvg;
    int32_t extra[7];
};
static_assert(sizeof(MatHeader) == 20 && sizeof(MatRecLoc) == 24, "on-disk sizes");
static_assert(sizeof(MatGeometry) == 152 && sizeof(MatRecord) == 184, "on-disk sizes");

// MatchFile.cpp:372-406 (POSIX separators)
std::string relative_name(const std::string& self, const std::string& other)
{
    size_t i = 0;
    while (i < self.size() && i < other.size() && self[i] == other[i]) ++i;
    while (i > 0 && (i >= self.size() || self[i] != '/')) --i;
    if (i == 0) return other;
    std::string rel;
    int up_self = 0, down_other = 0;
    for (size_t p = i + 1; p < self.size(); ++p)
        if (self[p] == '/' && (p + 1 >= self.size() || self[p + 1] != '/')) {
            rel += "../";
            ++up_self;
        }
    for (size_t p = i + 1; p < other.size(); ++p)
        if (other[p] == '/' && (p + 1 >= other.size() || other[p + 1] != '/')) ++down_other;
    if (up_self <= 1 || down_other <= 1) return rel + other.substr(i + 1);
    return other;
}

// MatchFile::multi_thread_mode (MatchFile.cpp:37, :221-231): with it on, the reference takes a mutex named after
// the file from OpenMatchFile to CloseMatchFile.  Here an advisory lock on the file itself (flock) takes that place:
// exclusive for writers, shared for readers, held from open to close, across processes.
std::atomic<int> g_mat_locking{0};

// One image's match database, directory held in memory.
class MatFile {
public:
    bool open(const std::string& image, bool for_write)
    {
        image_ = image;
        const std::string path = image + ".mat";
        fd_ = ::open(path.c_str(), for_write ? (O_CREAT | O_RDWR) : O_RDONLY, S_IRUSR | S_IWUSR);
        if (fd_ < 0) return false;
        if (g_mat_locking.load(std::memory_order_relaxed)) {
            int rc;
            do rc = ::flock(fd_, for_write ? LOCK_EX : LOCK_SH); while (rc != 0 && errno == EINTR);
            if (rc != 0) return false;
        }
        MatHeader h;
        ssize_t got = read_at(0, &h, sizeof h);
        if (got == (ssize_t)sizeof h && h.definition_size > 0 && h.definition_buf >= h.definition_size &&
            h.version == kMatVersion && h.file_count >= 0) {
            hdr_ = h;
            dir_.assign((size_t)h.definition_buf, 0);
            if (read_at(sizeof h, dir_.data(), (size_t)h.definition_size) != h.definition_size) return false;
            index_directory();
        } else {
            hdr_ = MatHeader{kMatVersion, 0, 0, 0, 0};  // MatchFile.cpp:563-573
            dir_.clear();
            recs_.clear();
            by_name_.clear();
        }
        return true;
    }
    ~MatFile() { close_fd(); }
    // The directory stays in memory; the descriptor can be dropped and taken again (long runs over
    // more files than a process may hold open).
    void close_fd()
    {
        if (fd_ >= 0) ::close(fd_);
        fd_ = -1;
    }
    bool ensure_fd()
    {
        if (fd_ >= 0) return true;
        fd_ = ::open((image_ + ".mat").c_str(), O_CREAT | O_RDWR, S_IRUSR | S_IWUSR);
        return fd_ >= 0;
    }

    // first record with this name (the reference searches linearly, MatchFile.cpp:347-369)
    int find(const std::string& rel) const
    {
        auto it = by_name_.find(rel);
        return it == by_name_.end() ? -1 : it->second;
    }
    const MatRecLoc* rec(size_t i) const { return reinterpret_cast<const MatRecLoc*>(dir_.data() + recs_[i]); }
    MatRecLoc* rec(size_t i) { return reinterpret_cast<MatRecLoc*>(dir_.data() + recs_[i]); }
    size_t count() const { return recs_.size(); }
    const std::string& image() const { return image_; }

    // MatchFile.cpp:455-561.  Returns the new record's index or -1.
    int add(const std::string& rel, int reservation)
    {
        const int slen = ((int)rel.size() / 4 + 1) * 4;
        const int elen = slen - 4;
        const int rlen = elen + (int)sizeof(MatRecLoc);
        if (hdr_.file_count == 0) {
            hdr_.version = kMatVersion;
            hdr_.definition_buf = std::max(20, reservation) * (kMaxPath + (int)sizeof(MatRecLoc));
            dir_.assign((size_t)hdr_.definition_buf, 0);
            hdr_.definition_size = 0;
            recs_.clear();
            by_name_.clear();
            append_record(rel, elen, rlen, (int32_t)sizeof(MatHeader) + hdr_.definition_buf);
            return 0;
        }
        if (hdr_.definition_size + rlen < hdr_.definition_buf) {
            const MatRecLoc* last = rec(recs_.size() - 1);
            append_record(rel, elen, rlen, last->read_loc + last->block_size);
            return (int)recs_.size() - 1;
        }
        return grow_and_add(rel, elen, rlen) ? (int)recs_.size() - 1 : -1;
    }

    bool flush_directory() const
    {
        return write_at(0, &hdr_, sizeof hdr_) && write_at(sizeof hdr_, dir_.data(), (size_t)hdr_.definition_size);
    }
    ssize_t read_at(off_t off, void* p, size_t n) const
    {
        if (::lseek(fd_, off, SEEK_SET) < 0) return -1;
        char* c = static_cast<char*>(p);
        size_t got = 0;
        while (got < n) {
            ssize_t r = ::read(fd_, c + got, n - got);
            if (r < 0 && errno == EINTR) continue;
            if (r < 0) return -1;
            if (r == 0) break;
            got += (size_t)r;
        }
        return (ssize_t)got;
    }
    bool write_at(off_t off, const void* p, size_t n) const
    {
        if (::lseek(fd_, off, SEEK_SET) < 0) return false;
        const char* c = static_cast<const char*>(p);
        while (n) {
            ssize_t w = ::write(fd_, c, n);
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) return false;
            c += w;
            n -= (size_t)w;
        }
        return true;
    }
    // Moves the directory record i behind the last one; the caller re-writes its data at the new place
    // (MatchFile.cpp:422-453, bookkeeping included: the predecessor's trash_size grows by the LENGTH OF THE
    // DIRECTORY RECORD, not by the size of the vacated block, and when record 0 moves the header's
    // definition_buf grows by that length instead; the moved record keeps its own trash_size).
    void move_to_end(int i)
    {
        if (i < 0 || i >= (int)recs_.size() - 1) return;
        const size_t a = recs_[i], b = recs_[i + 1], len = b - a;
        std::vector<char> tmp(dir_.begin() + a, dir_.begin() + b);
        dir_.erase(dir_.begin() + a, dir_.begin() + b);
        dir_.insert(dir_.begin() + (hdr_.definition_size - len), tmp.begin(), tmp.end());
        index_directory();
        if (i == 0) {
            hdr_.definition_buf += (int32_t)len;
            dir_.resize((size_t)hdr_.definition_buf, 0);
        } else {
            rec(i - 1)->trash_size += (int32_t)len;
        }
        MatRecLoc* last = rec(recs_.size() - 1);
        const MatRecLoc* prev = rec(recs_.size() - 2);
        last->read_loc = prev->read_loc + prev->block_size;
    }

private:
    void index_directory()
    {
        // MatchFile.cpp:275-299: stop at the first implausible record
        recs_.clear();
        size_t p = 0;
        for (int i = 0; i < hdr_.file_count; ++i) {
            if (p + sizeof(MatRecLoc) > dir_.size()) break;
            const MatRecLoc* r = reinterpret_cast<const MatRecLoc*>(dir_.data() + p);
            if (r->read_loc <= 0 || r->block_size == 0 || r->extra_size > kMaxPath || r->extra_size < 0 ||
                r->fcount > 1000000 || r->fcount < 0)
                break;
            recs_.push_back(p);
            p += sizeof(MatRecLoc) + (size_t)r->extra_size;
        }
        by_name_.clear();
        for (size_t i = 0; i < recs_.size(); ++i) by_name_.emplace(rec(i)->file_name, (int)i);   // keeps the first
        if ((int)recs_.size() != hdr_.file_count) {
            hdr_.file_count = (int)recs_.size();
            hdr_.definition_size = (int32_t)p;
        }
    }
    void append_record(const std::string& rel, int elen, int rlen, int32_t read_loc)
    {
        const size_t at = (size_t)hdr_.definition_size;
        MatRecLoc* r = reinterpret_cast<MatRecLoc*>(dir_.data() + at);
        memset(r, 0, (size_t)rlen);
        r->read_loc = read_loc;
        r->extra_size = elen;
        memcpy(r->file_name, rel.c_str(), rel.size() + 1);
        recs_.push_back(at);
        by_name_.emplace(rel, (int)recs_.size() - 1);
        hdr_.definition_size += rlen;
        hdr_.file_count = (int)recs_.size();
    }
    // The directory is full: double it, relocating the data blocks that sit where the
    // larger directory will be to the end of the file, and rotating their records to the
    // end of the directory so that the last record stays the last block (MatchFile.cpp:494-560).
    bool grow_and_add(const std::string& rel, int elen, int rlen)
    {
        const int extra_needed = rlen + hdr_.definition_buf;
        int moved = 0, space = 0;
        while (space < extra_needed && moved < (int)recs_.size()) {
            space += rec(moved)->block_size + rec(moved)->trash_size;
            ++moved;
        }
        if (space < extra_needed) return false;
        std::vector<char> data((size_t)space);
        const int32_t first_loc = rec(0)->read_loc;
        if (read_at(first_loc, data.data(), (size_t)space) < 0) return false;
        const MatRecLoc* last = rec(recs_.size() - 1);
        int32_t wloc = last->read_loc + last->block_size;
        size_t doff = 0;
        for (int i = 0; i < moved; ++i) {
            MatRecLoc* r = rec(i);
            if (!write_at(wloc, data.data() + doff, (size_t)r->block_size)) return false;
            doff += (size_t)r->block_size + (size_t)r->trash_size;
            r->read_loc = wloc;
            wloc += r->block_size;
            r->trash_size = 0;
        }
        if (moved < (int)recs_.size()) {
            const size_t cut = recs_[moved];
            std::rotate(dir_.begin(), dir_.begin() + cut, dir_.begin() + hdr_.definition_size);
        }
        hdr_.definition_buf = hdr_.definition_buf * 2 + rlen;
        dir_.resize((size_t)hdr_.definition_buf, 0);
        index_directory();
        append_record(rel, elen, rlen, wloc);
        return true;
    }

    int fd_ = -1;
    std::string image_;
    MatHeader hdr_{};
    std::vector<char> dir_;
    std::vector<size_t> recs_;  // byte offsets of the records inside dir_
    std::unordered_map<std::string, int> by_name_;
};

// One record into an open file.  flush_dir = false leaves the changed directory in memory only.
int mat_put(MatFile& mf, const std::string& other, int fc_other, int nm, const int32_t* row_this,
            const int32_t* row_other, int reservation, bool flush_dir)
{
    const std::string& image = mf.image();
    const std::string rel = relative_name(image, other);
    if ((int)rel.size() >= kMaxPath) return VSM_EINVAL;
    int idx = mf.find(rel);
    if (idx < 0) idx = mf.add(rel, reservation);
    if (idx < 0) return VSM_EIO;
    MatRecLoc* loc = mf.rec(idx);
    loc->fcount = fc_other;
    // MatchFile.cpp:781-805
    const int fresh_size = (int)sizeof(MatRecord) + std::max(nm, 8) * (int)sizeof(int32_t) * 6;
    if (loc->block_size == 0 || idx == (int)mf.count() - 1) {
        loc->block_size = fresh_size;
    } else {
        const int need = (int)sizeof(MatRecord) + std::max(nm, 0) * (int)sizeof(int32_t) * 4;
        if (need > loc->block_size) {
            if (need > loc->block_size + loc->trash_size) {
                mf.move_to_end(idx);   // the vacated block stays in the file as dead bytes
                idx = (int)mf.count() - 1;
                loc = mf.rec(idx);
                loc->block_size = fresh_size;
            } else {
                loc->block_size += loc->trash_size;
                loc->trash_size = 0;
            }
        }
    }
    if (flush_dir && !mf.flush_directory()) return VSM_EIO;
    // MatchFile.cpp:813-831
    MatRecord rec;
    memset(&rec, 0, sizeof rec);
    rec.version = kRecVersion;
    rec.tvg.GE = 100.0f;
    std::vector<char> blk(sizeof(int32_t) + sizeof rec + (size_t)std::max(nm, 0) * 8);
    const int32_t nm32 = nm;
    memcpy(blk.data(), &nm32, 4);
    memcpy(blk.data() + 4, &rec, sizeof rec);
    if (nm > 0) {
        memcpy(blk.data() + 4 + sizeof rec, row_this, (size_t)nm * 4);
        memcpy(blk.data() + 4 + sizeof rec + (size_t)nm * 4, row_other, (size_t)nm * 4);
    }
    return mf.write_at(loc->read_loc, blk.data(), blk.size()) ? VSM_OK : VSM_EIO;
}

int mat_write_one(const std::string& image, const std::string& other, int fc_other, int nm,
                  const int32_t* row_this, const int32_t* row_other, int reservation)
{
    MatFile mf;
    if (!mf.open(image, true)) return VSM_EIO;
    return mat_put(mf, other, fc_other, nm, row_this, row_other, reservation, true);
}

}  // namespace

// Writer for long runs (the reference re-opens both files, re-reads and re-writes their directories
// and searches the partner name linearly on every WritePMatch call, MatchFile.cpp:347-369, 751-832):
// every touched file's directory is parsed once and kept in memory, written back when the writer is
// closed; at most `max_open` descriptors stay open.  The files end up byte for byte as repeated
// vsm_mat_write_pmatch calls leave them.
struct vsm_matwriter {
    int reservation = 0;
    size_t max_open = 256;
    std::unordered_map<std::string, std::unique_ptr<MatFile>> files;
    std::list<MatFile*> open_order;   // least recently used first
    bool failed = false;

    MatFile* get(const std::string& image)
    {
        auto it = files.find(image);
        MatFile* mf;
        if (it == files.end()) {
            std::unique_ptr<MatFile> fresh(new MatFile());
            if (!fresh->open(image, true)) return nullptr;
            mf = fresh.get();
            files.emplace(image, std::move(fresh));
        } else {
            mf = it->second.get();
            open_order.remove(mf);
            if (!mf->ensure_fd()) return nullptr;
        }
        open_order.push_back(mf);
        while (open_order.size() > max_open) {
            open_order.front()->close_fd();
            open_order.pop_front();
        }
        return mf;
    }
};

extern "C" int vsm_mat_set_locking(int on)
{
    return g_mat_locking.exchange(on ? 1 : 0);
}

extern "C" vsm_matwriter* vsm_mat_writer_open(int reservation, int max_open)
{
    vsm_matwriter* w = new vsm_matwriter();
    w->reservation = reservation;
    if (max_open > 0) w->max_open = (size_t)max_open;
    return w;
}

extern "C" int vsm_mat_writer_put(vsm_matwriter* w, const char* image1, const char* image2, int fc1, int fc2,
                                  int nm, const int32_t* idx1, const int32_t* idx2)
{
    if (!w || !image1 || !image2 || !image1[0] || !image2[0] || nm < 0 || (nm > 0 && (!idx1 || !idx2))) return VSM_EINVAL;
    if (strlen(image1) + 5 > (size_t)kMaxPath || strlen(image2) + 5 > (size_t)kMaxPath) return VSM_EINVAL;
    if (w->failed) return VSM_EIO;
    MatFile* a = w->get(image1);
    int rc = a ? mat_put(*a, image2, fc2, nm, idx1, idx2, w->reservation, false) : VSM_EIO;
    if (rc == VSM_OK) {
        MatFile* b = w->get(image2);
        rc = b ? mat_put(*b, image1, fc1, nm, idx2, idx1, w->reservation, false) : VSM_EIO;
    }
    if (rc == VSM_EIO) w->failed = true;
    return rc;
}

extern "C" int vsm_mat_writer_close(vsm_matwriter* w)
{
    if (!w) return VSM_EINVAL;
    bool ok = !w->failed;
    for (auto& kv : w->files) {
        MatFile* mf = kv.second.get();
        if (mf->count() == 0) continue;
        if (!mf->ensure_fd() || !mf->flush_directory()) ok = false;
        mf->close_fd();
    }
    delete w;
    return ok ? VSM_OK : VSM_EIO;
}

extern "C" int vsm_mat_write_pmatch(const char* image1, const char* image2, int fc1, int fc2,
                                    int nm, const int32_t* idx1, const int32_t* idx2, int reservation)
{
    if (!image1 || !image2 || !image1[0] || !image2[0] || nm < 0 || (nm > 0 && (!idx1 || !idx2))) return VSM_EINVAL;
    if (strlen(image1) + 5 > (size_t)kMaxPath || strlen(image2) + 5 > (size_t)kMaxPath) return VSM_EINVAL;
    int rc = mat_write_one(image1, image2, fc2, nm, idx1, idx2, reservation);
    if (rc != VSM_OK) return rc;
    return mat_write_one(image2, image1, fc1, nm, idx2, idx1, reservation);
}

extern "C" int vsm_mat_get_pmatch(const char* image, const char* other, int fc_other,
                                  int32_t* idx_this, int32_t* idx_other, int cap)
{
    if (!image || !other) return 0;
    MatFile mf;
    if (!mf.open(image, false)) return 0;
    const int idx = mf.find(relative_name(image, other));
    if (idx < 0) return 0;
    const MatRecLoc* loc = mf.rec(idx);
    if (loc->fcount != fc_other) return 0;  // MatchFile.cpp:871
    int32_t nm = 0;
    if (mf.read_at(loc->read_loc, &nm, 4) != 4) return 0;
    if (!(nm > 0 && nm <= fc_other)) return 0;  // :877
    const int take = std::min(nm, std::max(cap, 0));
    const off_t rows = (off_t)loc->read_loc + 4 + (off_t)sizeof(MatRecord);
    if (idx_this && mf.read_at(rows, idx_this, (size_t)take * 4) != (ssize_t)take * 4) return 0;
    if (idx_other && mf.read_at(rows + (off_t)nm * 4, idx_other, (size_t)take * 4) != (ssize_t)take * 4) return 0;
    return nm;
}

extern "C" int vsm_mat_get_pmatch_r(const char* image, const char* other, int fc_other,
                                    int32_t* row0, int32_t* row1, int cap)
{
    // MatchFile.cpp:897-936: the same record, first stored row into matches[1], second into matches[0]
    return vsm_mat_get_pmatch(image, other, fc_other, row1, row0, cap);
}

extern "C" int vsm_mat_get_match_count(const char* image, const char* other, int* nm, int* nf)
{
    if (nm) *nm = 0;
    if (nf) *nf = 0;
    if (!image || !other) return 0;
    MatFile mf;
    if (!mf.open(image, false)) return 0;
    const int idx = mf.find(relative_name(image, other));
    if (idx < 0) return 0;
    int32_t counts[3];  // NM, record version, NF (MatchFile.cpp:950-954)
    if (mf.read_at(mf.rec(idx)->read_loc, counts, sizeof counts) != (ssize_t)sizeof counts) return 0;
    if (nm) *nm = counts[0];
    if (nf) *nf = counts[2];
    return 1;
}

// ---- inlier records ---------------------------------------------------------------------
namespace {

// MatchFile::WriteIMatch(image_match, reverse, NM = 0, tvg, inliers), MatchFile.cpp:660-748, for the
// geometry WriteFMatch builds (TwoViewGeometry(NF, F), MatchFile.h:85-92).  The record must exist.  Every record
// written by WritePMatch has room (block = 184 + max(NM, 8) * 24 bytes against 188 + (NM + NF) * 8 needed, NF <= NM);
// for one that has not (a file from elsewhere, or more inliers than putative matches) the re-allocation branch
// (:690-719) runs: the last record of the file, or one without a block, gets a block of 184 + max(NM + NF, 8) * 16
// bytes where it lies; one whose block and the dead bytes behind it suffice takes those; any other moves to the end
// of the file as MoveRecordToEnd does (:422-453) with a block of that size.  In that last case the reference writes
// only the geometry and the inlier lists at the new place and leaves the count and the putative lists behind, so that
// its own readers (GetPMatch :877, GetIMatch :980-990) find NM = 0 there; here both are carried along, and the record
// stays readable.  flush_dir = false leaves the changed directory in memory only (the cached writer).
struct MatPose {   // NE, R, T, F1, F2 of TwoViewGeometry (MatchFile.h:68-83); ne = 0: none
    int ne = 0;
    const float* R = nullptr;
    const float* T = nullptr;
    float f1 = 0.0f, f2 = 0.0f;
};

int mat_put_inliers(MatFile& mf, const std::string& other, bool reverse, int nf,
                    const int32_t* row_a, const int32_t* row_b, const float* F, const MatPose& pose = MatPose(),
                    bool flush_dir = true)
{
    int idx = mf.find(relative_name(mf.image(), other));
    if (idx < 0) return 0;
    MatRecLoc* loc = mf.rec(idx);
    int32_t nm = 0;
    if (mf.read_at(loc->read_loc, &nm, 4) != 4 || nm < 0) return 0;
    const long long need = (1 + ((long long)nm + nf) * 2) * 4 + (long long)sizeof(MatRecord);
    if (need > loc->block_size) {   // MatchFile.cpp:690-719
        const long long fresh = (long long)sizeof(MatRecord) + std::max<long long>((long long)nm + nf, 8) * 16;
        if (fresh > INT32_MAX) return VSM_EINVAL;
        if (loc->block_size == 0 || idx == (int)mf.count() - 1) {
            loc->block_size = (int32_t)fresh;
        } else if (need > (long long)loc->block_size + loc->trash_size) {
            // the count and the putative lists go with the record (the reference leaves them behind, see above)
            std::vector<char> lists(4 + (size_t)nm * 8);
            if (mf.read_at(loc->read_loc, lists.data(), 4) != 4 ||
                mf.read_at((off_t)loc->read_loc + 4 + (off_t)sizeof(MatRecord), lists.data() + 4, (size_t)nm * 8) != (ssize_t)nm * 8)
                return VSM_EIO;
            mf.move_to_end(idx);
            idx = (int)mf.count() - 1;
            loc = mf.rec(idx);
            loc->block_size = (int32_t)fresh;
            if (!mf.write_at(loc->read_loc, lists.data(), 4) ||
                !mf.write_at((off_t)loc->read_loc + 4 + (off_t)sizeof(MatRecord), lists.data() + 4, (size_t)nm * 8))
                return VSM_EIO;
        } else {
            loc->block_size += loc->trash_size;
            loc->trash_size = 0;
        }
        if (flush_dir && !mf.flush_directory()) return VSM_EIO;
    }
    MatRecord rec;
    memset(&rec, 0, sizeof rec);
    rec.version = kRecVersion;
    rec.tvg.GE = 100.0f;
    rec.tvg.NF = nf;
    if (nf > 0)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) rec.tvg.F[i][j] = reverse ? F[3 * j + i] : F[3 * i + j];   // SetGeometryR transposes
    // SetGeometry / SetGeometryR (MatchFile.h:121-161): NE and the focal lengths always, R and T only
    // beside an F; the exchanged view stores R^T and -R^T T, each component summed in the reference's order
    rec.tvg.NE = pose.ne;
    rec.tvg.F1 = reverse ? pose.f2 : pose.f1;
    rec.tvg.F2 = reverse ? pose.f1 : pose.f2;
    if (nf > 0 && pose.ne > 0) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) rec.tvg.R[i][j] = reverse ? pose.R[3 * j + i] : pose.R[3 * i + j];
        for (int i = 0; i < 3; ++i) {
            if (reverse) {
                float t = -rec.tvg.R[i][0] * pose.T[0];
                t = t - rec.tvg.R[i][1] * pose.T[1];
                t = t - rec.tvg.R[i][2] * pose.T[2];
                rec.tvg.T[i] = t;
            } else {
                rec.tvg.T[i] = pose.T[i];
            }
        }
    }
    if (!mf.write_at((off_t)loc->read_loc + 4, &rec, sizeof rec)) return VSM_EIO;
    if (nf > 0) {
        const off_t at = (off_t)loc->read_loc + 4 + (off_t)sizeof rec + (off_t)nm * 8;
        const int32_t* first = reverse ? row_b : row_a;
        const int32_t* second = reverse ? row_a : row_b;
        if (!mf.write_at(at, first, (size_t)nf * 4) || !mf.write_at(at + (off_t)nf * 4, second, (size_t)nf * 4)) return VSM_EIO;
    }
    return 1;
}

}  // namespace

extern "C" int vsm_mat_write_fmatch(const char* image1, const char* image2, int nf, const int32_t* idx1,
                                    const int32_t* idx2, const float* F)
{
    if (!image1 || !image2 || !image1[0] || !image2[0] || nf < 0 || (nf > 0 && (!idx1 || !idx2 || !F))) return VSM_EINVAL;
    if (strlen(image1) + 5 > (size_t)kMaxPath || strlen(image2) + 5 > (size_t)kMaxPath) return VSM_EINVAL;
    // one file at a time, as MatchFile.cpp:643-658 closes the first before it opens the second: with locking on,
    // two writers that take the same two files in opposite order must not each hold one and wait for the other
    int a, b;
    {
        MatFile m1;
        if (!m1.open(image1, true)) return VSM_EIO;
        a = mat_put_inliers(m1, image2, false, nf, idx1, idx2, F);
    }
    if (a < 0) return a;
    {
        MatFile m2;
        if (!m2.open(image2, true)) return VSM_EIO;
        b = mat_put_inliers(m2, image1, true, nf, idx1, idx2, F);
    }
    if (b < 0) return b;
    return a == 1 && b == 1 ? 1 : 0;
}

extern "C" int vsm_mat_writer_put_fmatch(vsm_matwriter* w, const char* image1, const char* image2, int nf,
                                         const int32_t* idx1, const int32_t* idx2, const float* F)
{
    if (!w || !image1 || !image2 || !image1[0] || !image2[0] || nf < 0 || (nf > 0 && (!idx1 || !idx2 || !F))) return VSM_EINVAL;
    if (w->failed) return VSM_EIO;
    MatFile* a = w->get(image1);
    int ra = a ? mat_put_inliers(*a, image2, false, nf, idx1, idx2, F, MatPose(), false) : VSM_EIO;
    if (ra < 0) return ra;
    MatFile* b = w->get(image2);
    int rb = b ? mat_put_inliers(*b, image1, true, nf, idx1, idx2, F, MatPose(), false) : VSM_EIO;
    if (rb < 0) return rb;
    return ra == 1 && rb == 1 ? 1 : 0;
}

static int pose_args_bad(int nf, int ne, const float* R, const float* T)
{
    return ne < 0 || (ne > 0 && (!R || !T || nf <= 0));
}

extern "C" int vsm_mat_write_ematch(const char* image1, const char* image2, int nf, const int32_t* idx1,
                                    const int32_t* idx2, const float* F, int ne, const float* R, const float* T,
                                    float f1, float f2)
{
    if (!image1 || !image2 || !image1[0] || !image2[0] || nf < 0 || (nf > 0 && (!idx1 || !idx2 || !F))) return VSM_EINVAL;
    if (pose_args_bad(nf, ne, R, T)) return VSM_EINVAL;
    if (strlen(image1) + 5 > (size_t)kMaxPath || strlen(image2) + 5 > (size_t)kMaxPath) return VSM_EINVAL;
    MatPose pose;
    pose.ne = ne; pose.R = R; pose.T = T; pose.f1 = f1; pose.f2 = f2;
    int a, b;
    {
        MatFile m1;
        if (!m1.open(image1, true)) return VSM_EIO;
        a = mat_put_inliers(m1, image2, false, nf, idx1, idx2, F, pose);
    }
    if (a < 0) return a;
    {
        MatFile m2;
        if (!m2.open(image2, true)) return VSM_EIO;
        b = mat_put_inliers(m2, image1, true, nf, idx1, idx2, F, pose);
    }
    if (b < 0) return b;
    return a == 1 && b == 1 ? 1 : 0;
}

extern "C" int vsm_mat_writer_put_ematch(vsm_matwriter* w, const char* image1, const char* image2, int nf,
                                         const int32_t* idx1, const int32_t* idx2, const float* F, int ne,
                                         const float* R, const float* T, float f1, float f2)
{
    if (!w || !image1 || !image2 || !image1[0] || !image2[0] || nf < 0 || (nf > 0 && (!idx1 || !idx2 || !F))) return VSM_EINVAL;
    if (pose_args_bad(nf, ne, R, T)) return VSM_EINVAL;
    if (w->failed) return VSM_EIO;
    MatPose pose;
    pose.ne = ne; pose.R = R; pose.T = T; pose.f1 = f1; pose.f2 = f2;
    MatFile* a = w->get(image1);
    int ra = a ? mat_put_inliers(*a, image2, false, nf, idx1, idx2, F, pose, false) : VSM_EIO;
    if (ra < 0) return ra;
    MatFile* b = w->get(image2);
    int rb = b ? mat_put_inliers(*b, image1, true, nf, idx1, idx2, F, pose, false) : VSM_EIO;
    if (rb < 0) return rb;
    return ra == 1 && rb == 1 ? 1 : 0;
}

extern "C" int vsm_mat_get_pose(const char* image, const char* other, int fc_other, int* ne, float* R, float* T,
                                float* f12)
{
    if (ne) *ne = 0;
    if (R) memset(R, 0, 9 * sizeof(float));
    if (T) memset(T, 0, 3 * sizeof(float));
    if (f12) f12[0] = f12[1] = 0.0f;
    if (!image || !other) return 0;
    MatFile mf;
    if (!mf.open(image, false)) return 0;
    const int idx = mf.find(relative_name(image, other));
    if (idx < 0) return 0;
    const MatRecLoc* loc = mf.rec(idx);
    if (loc->fcount != fc_other) return 0;   // MatchFile.cpp:975
    int32_t nm = 0;
    MatRecord rec;
    if (mf.read_at(loc->read_loc, &nm, 4) != 4) return 0;
    if (mf.read_at((off_t)loc->read_loc + 4, &rec, sizeof rec) != (ssize_t)sizeof rec) return 0;
    if (nm > loc->fcount || nm <= 0 || rec.version != kRecVersion || rec.tvg.NF > loc->fcount) return 0;   // :988, :995
    if (ne) *ne = rec.tvg.NE;
    if (R) memcpy(R, rec.tvg.R, 9 * sizeof(float));
    if (T) memcpy(T, rec.tvg.T, 3 * sizeof(float));
    if (f12) { f12[0] = rec.tvg.F1; f12[1] = rec.tvg.F2; }
    return nm;
}

static int mat_get_inliers(const char* image, const char* other, int fc_other, bool reverse, float* F, int* nf,
                           int32_t* row0, int32_t* row1, int cap)
{
    if (nf) *nf = 0;
    if (F) memset(F, 0, 9 * sizeof(float));
    if (!image || !other) return 0;
    MatFile mf;
    if (!mf.open(image, false)) return 0;
    const int idx = mf.find(relative_name(image, other));
    if (idx < 0) return 0;
    const MatRecLoc* loc = mf.rec(idx);
    if (loc->fcount != fc_other) return 0;   // MatchFile.cpp:975
    int32_t nm = 0;
    MatRecord rec;
    if (mf.read_at(loc->read_loc, &nm, 4) != 4) return 0;
    if (mf.read_at((off_t)loc->read_loc + 4, &rec, sizeof rec) != (ssize_t)sizeof rec) return 0;
    if (nm > loc->fcount || nm < 0) return 0;                                          // :988
    if (nm == 0 || rec.version != kRecVersion || rec.tvg.NF > loc->fcount) return 0;   // :995
    const int n = rec.tvg.NF;
    if (n > 0) {
        if (F)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) F[3 * i + j] = reverse ? rec.tvg.F[j][i] : rec.tvg.F[i][j];
        const int take = std::min(n, std::max(cap, 0));
        const off_t at = (off_t)loc->read_loc + 4 + (off_t)sizeof rec + (off_t)nm * 8;
        // :1011 reads both rows in stored order; :1074-1075 reads the first stored row into inliers[1]
        int32_t* first = reverse ? row1 : row0;
        int32_t* second = reverse ? row0 : row1;
        if (first && mf.read_at(at, first, (size_t)take * 4) != (ssize_t)take * 4) return 0;
        if (second && mf.read_at(at + (off_t)n * 4, second, (size_t)take * 4) != (ssize_t)take * 4) return 0;
    }
    if (nf) *nf = n > 0 ? n : 0;
    return nm;
}

extern "C" int vsm_mat_get_imatch(const char* image, const char* other, int fc_other, float* F, int* nf,
                                  int32_t* row0, int32_t* row1, int cap)
{
    return mat_get_inliers(image, other, fc_other, false, F, nf, row0, row1, cap);
}

extern "C" int vsm_mat_get_imatch_r(const char* image, const char* other, int fc_other, float* F, int* nf,
                                    int32_t* row0, int32_t* row1, int cap)
{
    return mat_get_inliers(image, other, fc_other, true, F, nf, row0, row1, cap);
}
