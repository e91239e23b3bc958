// Queries against a map database sliced over the GPUs of one machine, in one process: the database rows are cut
// into contiguous slices, every device searches its slices (vsm_knn2_device), the per-slice 2-NN lists meet in an
// RCCL all-gather over the devices, and a kernel merges them (vsm_merge_top2_device).  Declarations: include/vsm.h.
//
// The reference's only brute-force nearest-neighbour search is NearestEmbedFunc.forward
// (Graph_Idea/model/model_2_0.py:43-78: every query against every code, argmin); this is that search with the
// codes spread over several devices' memories.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the six entry points used here), so the library itself loads
// without it and a process that has loaded a copy already (PyTorch ships one) shares it.
#include "../../include/vsm.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;

    bool load()
    {
        if (handle) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) {
            error = std::string("cannot load librccl: ") + dlerror();
            return false;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(handle, n);
            if (!p && error.empty()) error = std::string("librccl lacks ") + n;
            return p;
        };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!error.empty()) {
            dlclose(handle);
            handle = nullptr;
            return false;
        }
        return true;
    }
};

thread_local std::string g_dbq_create_error;

}  // namespace

struct vsm_dbq {
    std::vector<int> devices;
    std::vector<vsm_ctx*> ctx;           // one per device
    std::vector<ncclComm_t> comm;        // one per device (ncclCommInitAll)
    std::vector<void*> send, recv;       // per device: [slices of the device][4][nq] and [nslices][4][nq] words
    size_t buf_queries = 0;              // queries the buffers above were sized for
    int nslices = 0, per_dev = 0;
    std::vector<int> slice_set;          // resident set of every slice, on ctx[slice / per_dev]
    std::vector<int64_t> base;           // slice s owns database rows [base[s], base[s + 1])
    int dim = 0;
    Rccl rccl;
    std::string err;
};

namespace {

int dbq_fail(vsm_dbq* q, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (q) q->err = buf;
    else g_dbq_create_error = buf;
    return code;
}

void dbq_free_buffers(vsm_dbq* q)
{
    for (size_t d = 0; d < q->devices.size(); ++d) {
        if (hipSetDevice(q->devices[d]) != hipSuccess) continue;
        if (d < q->send.size() && q->send[d]) (void)hipFree(q->send[d]);
        if (d < q->recv.size() && q->recv[d]) (void)hipFree(q->recv[d]);
    }
    q->send.assign(q->devices.size(), nullptr);
    q->recv.assign(q->devices.size(), nullptr);
    q->buf_queries = 0;
}

void dbq_release_slices(vsm_dbq* q)
{
    for (int s = 0; s < (int)q->slice_set.size(); ++s)
        if (q->slice_set[s] >= 0) vsm_release_set(q->ctx[s / q->per_dev], q->slice_set[s]);
    q->slice_set.assign(q->nslices, -1);
}

}  // namespace

extern "C" vsm_dbq* vsm_dbq_create(const int* devices, int ndev, int nslices)
{
    if (!devices || ndev < 1 || nslices < ndev || nslices % ndev != 0) {
        dbq_fail(nullptr, VSM_EINVAL, "vsm_dbq_create wants ndev >= 1 devices and a multiple of ndev slices (got %d, %d)", ndev, nslices);
        return nullptr;
    }
    for (int a = 0; a < ndev; ++a)
        for (int b = a + 1; b < ndev; ++b)
            if (devices[a] == devices[b]) {
                dbq_fail(nullptr, VSM_EINVAL, "device %d is named twice: a collective takes one rank per device "
                                              "(give one device several slices instead)", devices[a]);
                return nullptr;
            }
    vsm_dbq* q = new vsm_dbq();
    q->devices.assign(devices, devices + ndev);
    q->nslices = nslices;
    q->per_dev = nslices / ndev;
    q->slice_set.assign(nslices, -1);
    q->base.assign(nslices + 1, 0);
    q->send.assign(ndev, nullptr);
    q->recv.assign(ndev, nullptr);
    for (int d = 0; d < ndev; ++d) {
        vsm_ctx* c = vsm_create(devices[d]);
        if (!c) {
            dbq_fail(nullptr, VSM_ENODEV, "vsm_create(%d): %s", devices[d], vsm_last_error(nullptr));
            vsm_dbq_destroy(q);
            return nullptr;
        }
        q->ctx.push_back(c);
    }
    if (!q->rccl.load()) {
        dbq_fail(nullptr, VSM_ENODEV, "%s", q->rccl.error.c_str());
        vsm_dbq_destroy(q);
        return nullptr;
    }
    q->comm.assign(ndev, nullptr);
    const ncclResult_t r = q->rccl.CommInitAll(q->comm.data(), ndev, q->devices.data());
    if (r != ncclSuccess) {
        dbq_fail(nullptr, VSM_ENODEV, "ncclCommInitAll over %d devices: %s", ndev, q->rccl.GetErrorString(r));
        q->comm.clear();
        vsm_dbq_destroy(q);
        return nullptr;
    }
    return q;
}

extern "C" void vsm_dbq_destroy(vsm_dbq* q)
{
    if (!q) return;
    dbq_release_slices(q);
    dbq_free_buffers(q);
    for (ncclComm_t c : q->comm)
        if (c) (void)q->rccl.CommDestroy(c);
    for (vsm_ctx* c : q->ctx) vsm_destroy(c);
    delete q;
}

extern "C" const char* vsm_dbq_last_error(const vsm_dbq* q) { return q ? q->err.c_str() : g_dbq_create_error.c_str(); }

extern "C" int vsm_dbq_slices(const vsm_dbq* q, int64_t* base)
{
    if (!q) return VSM_EINVAL;
    if (base) memcpy(base, q->base.data(), sizeof(int64_t) * (q->nslices + 1));
    return q->nslices;
}

extern "C" int vsm_dbq_load(vsm_dbq* q, const float* rows, long long n, int dim, int layout)
{
    if (!q) return VSM_EINVAL;
    if (n < 0 || (n > 0 && !rows) || dim < 1 || dim > 256) return dbq_fail(q, VSM_EINVAL, "bad database (n = %lld, dim = %d)", n, dim);
    if (layout != VSM_LAYOUT_ROWS && layout != VSM_LAYOUT_DIM_MAJOR) return dbq_fail(q, VSM_EINVAL, "unknown layout %d", layout);
    dbq_release_slices(q);
    q->dim = dim;
    for (int s = 0; s <= q->nslices; ++s) q->base[s] = (int64_t)((__int128)n * s / q->nslices);
    const int ndev = (int)q->devices.size();
    std::vector<std::string> errs(ndev);
    std::vector<int> codes(ndev, VSM_OK);
    auto load = [&](int d) {
        std::vector<float> cut;   // a slice of a [dim][n] database is not contiguous: it is gathered first
        for (int s = d * q->per_dev; s < (d + 1) * q->per_dev; ++s) {
            const int64_t lo = q->base[s], m = q->base[s + 1] - lo;
            if (m > INT32_MAX) {
                codes[d] = VSM_EINVAL;
                errs[d] = "a slice holds more than 2^31 rows: use more slices";
                return;
            }
            const float* src = rows + (size_t)lo * dim;
            if (layout == VSM_LAYOUT_DIM_MAJOR) {
                cut.resize((size_t)m * dim);
                for (int k = 0; k < dim; ++k) memcpy(cut.data() + (size_t)k * m, rows + (size_t)k * n + lo, sizeof(float) * (size_t)m);
                src = cut.data();
            }
            const int id = vsm_upload_f32(q->ctx[d], src, (int)m, dim, layout);
            if (id < 0) {
                codes[d] = id;
                errs[d] = vsm_last_error(q->ctx[d]);
                return;
            }
            q->slice_set[s] = id;
        }
    };
    std::vector<std::thread> th;
    for (int d = 1; d < ndev; ++d) th.emplace_back(load, d);
    load(0);
    for (auto& t : th) t.join();
    for (int d = 0; d < ndev; ++d)
        if (codes[d] != VSM_OK) return dbq_fail(q, codes[d], "slice upload on device %d: %s", q->devices[d], errs[d].c_str());
    return VSM_OK;
}

extern "C" int vsm_dbq_query(vsm_dbq* q, const float* queries, int nq, int layout, int metric,
                             float* k1, float* k2, int64_t* i1, int64_t* i2)
{
    if (!q) return VSM_EINVAL;
    if (nq < 0 || (nq > 0 && !queries) || !k1 || !k2 || !i1 || !i2) return dbq_fail(q, VSM_EINVAL, "bad argument");
    if (q->dim == 0) return dbq_fail(q, VSM_EINVAL, "no database loaded");
    if (nq == 0) return 0;
    const int ndev = (int)q->devices.size();
    const size_t slice_words = (size_t)4 * nq;
    if ((size_t)nq > q->buf_queries) {
        dbq_free_buffers(q);
        for (int d = 0; d < ndev; ++d) {
            hipError_t e = hipSetDevice(q->devices[d]);
            if (e == hipSuccess) e = hipMalloc(&q->send[d], slice_words * 4 * q->per_dev);
            if (e == hipSuccess) e = hipMalloc(&q->recv[d], slice_words * 4 * q->nslices);
            if (e != hipSuccess) return dbq_fail(q, VSM_ENOMEM, "result buffers on device %d: %s", q->devices[d], hipGetErrorString(e));
        }
        q->buf_queries = (size_t)nq;
    }
    // every device: the queries (each gets its own copy over its own link), then its slices one after the other
    std::vector<std::string> errs(ndev);
    std::vector<int> codes(ndev, VSM_OK);
    auto search = [&](int d) {
        vsm_ctx* c = q->ctx[d];
        const int qs = vsm_upload_f32(c, queries, nq, q->dim, layout);
        if (qs < 0) {
            codes[d] = qs;
            errs[d] = vsm_last_error(c);
            return;
        }
        for (int k = 0; k < q->per_dev && codes[d] == VSM_OK; ++k) {
            const int rc = vsm_knn2_device(c, qs, q->slice_set[d * q->per_dev + k], metric,
                                           static_cast<char*>(q->send[d]) + (size_t)k * slice_words * 4);
            if (rc < 0) {
                codes[d] = rc;
                errs[d] = vsm_last_error(c);
            }
        }
        vsm_release_set(c, qs);
    };
    std::vector<std::thread> th;
    for (int d = 1; d < ndev; ++d) th.emplace_back(search, d);
    search(0);
    for (auto& t : th) t.join();
    for (int d = 0; d < ndev; ++d)
        if (codes[d] != VSM_OK) return dbq_fail(q, codes[d], "search on device %d: %s", q->devices[d], errs[d].c_str());
    // the per-slice lists of all devices, in slice order, on every device
    ncclResult_t r = q->rccl.GroupStart();
    for (int d = 0; d < ndev && r == ncclSuccess; ++d)
        r = q->rccl.AllGather(q->send[d], q->recv[d], slice_words * q->per_dev, ncclInt32, q->comm[d],
                              static_cast<hipStream_t>(vsm_stream(q->ctx[d])));
    const ncclResult_t r2 = q->rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return dbq_fail(q, VSM_ENODEV, "ncclAllGather: %s", q->rccl.GetErrorString(r));
    for (int d = 0; d < ndev; ++d) {
        const int rc = vsm_synchronize(q->ctx[d]);
        if (rc != VSM_OK) return dbq_fail(q, rc, "all-gather on device %d: %s", q->devices[d], vsm_last_error(q->ctx[d]));
    }
    // any device could answer now; the first one merges
    const int rc = vsm_merge_top2_device(q->ctx[0], q->recv[0], q->base.data(), q->nslices, nq, metric, k1, k2, i1, i2);
    if (rc != VSM_OK) return dbq_fail(q, rc, "merge: %s", vsm_last_error(q->ctx[0]));
    return nq;
}
