// C-ABI layer: contexts, descriptor sets, submission of pair batches, result hand-back.
// Declarations and the reference code each entry point stands in for: include/vsm.h.
#include "../../include/vsm.h"
#include "vsm_internal.h"
#include "f32_kernels.h"

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

using namespace vsm;

// Scratch of vsm_match_pair_noalloc: the device memory its kernels need beyond the caller's inputs and
// outputs (pair record, work items, per-chunk partial results, candidate lists).  One copy per GPU,
// held until the process ends.  Host code never passes this name to a kernel: g_scratch_ptr[device]
// holds the array's address on that device (hipGetSymbolAddress, taken once per device in vsm_create).
// File scope on purpose: the runtime cannot resolve a device variable declared inside an unnamed
// namespace (hipGetSymbolAddress aborts with "Cannot create GlobalVar Obj").
constexpr size_t kNoAllocScratchBytes = (size_t)4 << 20;
constexpr int kNoAllocMaxRows = 16384;     // most rows per set of a vsm_match_pair_noalloc call
constexpr int kMaxDevices = 256;
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[kMaxDevices] = {};
// recorded behind every vsm_match_pair_noalloc call on the device and waited for by the next one: calls that share
// the array run one after the other even when they come from different contexts
static hipEvent_t g_scratch_done[kMaxDevices] = {};

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        // hipFree waits for the device to go idle, so growing a buffer that a queued submission
        // still uses is safe (it costs a stall, and only while the workspace is still growing)
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8;  // a little slack so near-equal batches reuse it
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            e = hipMalloc(&p, bytes);
            want = bytes;
        }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct HostBuf {  // pinned
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

enum SetKind { SET_FREE = 0, SET_U8 = 1, SET_F32 = 2, SET_XY = 3 };

struct DescSet {
    SetKind kind = SET_FREE;
    int n = 0, n_pad = 0, dim = 0;
    void* frag = nullptr;  // u8: fragment-major int8; f32: fragment-major float; xy: n x (x, y) floats
    void* nrm = nullptr;   // u8: int32 nrm8; f32: float squared norms
    void* shadow = nullptr;  // f32 only: bf16 operand-order copy + constants of the coarse pass
    Bf16Shadow shadow_view() const { return bf16_shadow_view(shadow, n_pad, dim); }
};

}  // namespace

struct vsm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    unsigned char* noalloc_scratch = nullptr;   // this device's copy of `scratch`
    hipEvent_t noalloc_done = nullptr;          // this device's g_scratch_done
    // Workgroups a float search is cut into when its database is long enough: two per CU, i.e. the coarse grid
    // is resident all at once (512 on a whole MI355X; twice as many ran 2.5 % slower on the 16384 x 1M query:
    // half as long survivor sublists, whose thresholds are young for a larger share of their tiles)
    int coarse_wgs = 512;
    std::string err;
    std::vector<DescSet> sets;
    // workspace (grow-only)
    DevBuf d_pairs, d_rr1, d_rrj, d_rr2, d_rr3, d_colpart, d_cand;
    DevBuf d_slot0, d_slot1, d_slotd, d_counts;
    DevBuf d_zero_tile;   // kTileBytes of zeros (see SiftCandidates::zero_tile)
    DevBuf d_dense0, d_dense1, d_knnj, d_knn1, d_knn2, d_stage, d_coarse, d_probs, d_geom, d_short;
    // float sets: 0 = bf16 coarse pass + exact re-rank (exact kernel when a survivor list overflows), 1 = exact kernel only
    int f32_mode = 0, last_f32_path = -1;
    int f32_reverse = VSM_F32_REVERSE_AUTO, last_f32_reverse = -1;
    bool f32_copied_out = false;   // the accept kernel of the last float submission copied the caller's words to pinned memory itself
    bool f32_watch_word = false;   // ... and writes a sequence number behind them when they are all on their way
    uint32_t f32_seq = 0;
    // What the last one-pair float call left zeroed on the device (overflow flag at the head of d_probs, the words the
    // splits of its searches shared in d_coarse); anything else that uses those buffers voids it.
    struct {
        const void* flag = nullptr;
        const void* thr = nullptr;
        int thr_words = 0;
        size_t probs_cap = 0, coarse_cap = 0;
    } f32_clean;
    // u8 sets: 0 = the tile-minimum query pass for ratio-test submissions (packed form otherwise), 1 = packed form only
    int u8_mode = 0, last_u8_path = -1;
    // VSM_U8_AUTO only: after a tile-minimum submission most of whose rows needed the rescan pass (data full of
    // exact duplicates), this many following submissions run the packed form before the tile-minimum form is tried again
    int u8_packed_left = 0;
    int last_u8_rescanned = -1;   // candidates of the last finished u8 submission that took the rescan pass
    int* flag_dev = nullptr;   // survivor-overflow flag of the last coarse submission (inside d_probs)
    HostBuf h_small, h_probs, h_gather;
    // Two sets of per-submission staging and result buffers, so that vsm_match_pairs can queue
    // submission k + 1 while the match lists of submission k travel to the host on copy_stream.
    struct ResultSlot {
        DevBuf d_offsets, d_tight0, d_tight1;
        HostBuf h_pairs, h_off;
        // A small u8 submission brings offsets and both lists to this pinned block itself (one kernel behind the
        // pack kernel, in front of the `packed` event): collecting it then costs no trip to the device at all.
        // Layout: [offsets and the two rescan counters: 2 * (npairs + 2) words][idx0: rows0 words][idx1: rows0 words]
        HostBuf h_res;
        bool res_on_host = false;
        size_t res_rows0 = 0;   // width of the two list segments in h_res
        int res_pairs = 0;
        hipEvent_t packed = nullptr;
        hipEvent_t staged = nullptr;    // the slot's pinned staging buffers have been copied to the device
        bool staged_recorded = false;
        // timing brackets of the submission that last used this slot
        hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
        bool ev_valid = false, ev_verify_valid = false;
        int query_pairs = 0;   // pairs in the launch the ev0..ev1 bracket belongs to
        bool u8_stats = false;    // a u8 submission used this slot last: two counters of its rescan pass lie behind the offsets
        bool tile_path = false;   // ... and it ran the tile-minimum query pass
        size_t rows0 = 0;         // its image-0 rows over all pairs
    } slots[2];
    int rs = 0;   // slot the next submission uses / the last one used
    // submissions queued by vsm_submit_pairs and not yet collected (at most two, oldest first)
    int inflight = 0, inflight_head = 0;
    int inflight_pairs[2] = {0, 0};
    bool inflight_on_host[2] = {false, false};   // float fall-back: lists already complete in the slot
    size_t scratch_budget = (size_t)8 << 30;   // device scratch one submission may use
    ResultSlot& R() { return slots[rs]; }
    hipStream_t copy_stream = nullptr;
    // last device-resident result
    int last_npairs = 0;
    long long last_total = 0;
    int last_slot = 0;     // result slot that holds those lists
    // timing
    int timing = 0;
    int timing_slot = 0;   // slot whose brackets vsm_last_kernel_ms / vsm_last_verify_ms read
};

namespace {

int fail(vsm_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((c), e_ == hipErrorOutOfMemory ? VSM_ENOMEM : VSM_ENODEV,           \
                        "%s failed: %s", #call, hipGetErrorString(e_));                     \
    } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

int new_set_slot(vsm_ctx* c)
{
    for (size_t i = 0; i < c->sets.size(); ++i)
        if (c->sets[i].kind == SET_FREE) return (int)i;
    c->sets.emplace_back();
    return (int)c->sets.size() - 1;
}

const DescSet* get_set(const vsm_ctx* c, int id)
{
    if (!c || id < 0 || id >= (int)c->sets.size() || c->sets[id].kind == SET_FREE) return nullptr;
    return &c->sets[id];
}

int check_params(vsm_ctx* c, const vsm_params* p)
{
    if (!p) return fail(c, VSM_EINVAL, "params is NULL");
    if (p->ratio_num > 0 && (p->ratio_den <= 0 || p->ratio_num > 16384 || p->ratio_den > 16384))
        return fail(c, VSM_EINVAL, "ratio %d/%d out of range (0 < num, den <= 16384)", p->ratio_num, p->ratio_den);
    return VSM_OK;
}

// One submission may use this many bytes of per-row scratch (partial results 12 per chunk, candidate records and
// the verify pass's row list 36, output slots and packed lists 16: about 64 bytes per image-0 row).
constexpr size_t kDefaultScratchBudget = (size_t)8 << 30;
constexpr size_t kScratchPerRow = 64;

struct Plan {
    std::vector<SiftPair> pairs;
    std::vector<SiftItem> items;   // query pass: (pair, strip, chunk)
    int max_vitems = 0;            // verify pass: blocks of kVerifyBlock candidates, at most
    size_t rows = 0, out0 = 0, out1 = 0;
};

// Train rows per work item of the query pass.  A submission of few pairs (one pair per call is the
// common case of a per-pair loop) is cut into narrower chunks of the train set, so that its query pass
// still has a few hundred workgroups.  strips_total: query strips of the whole submission.
int chunk_cols_for(long long strips_total, int nB_pad)
{
    const long long kWantItems = 256;
    if (strips_total <= 0 || strips_total * ((nB_pad + kChunkCols - 1) / kChunkCols) >= kWantItems) return kChunkCols;
    const long long want_chunks = (kWantItems + strips_total - 1) / strips_total;
    const long long cols = ((long long)nB_pad + want_chunks - 1) / want_chunks;
    return (int)std::min<long long>(kChunkCols, std::max<long long>(2 * kStepRows, (cols + kStepRows - 1) / kStepRows * kStepRows));
}

int plan_u8(vsm_ctx* c, const int32_t* pairs, int npairs, Plan& pl)
{
    pl.pairs.resize(npairs);
    pl.items.clear();
    pl.max_vitems = 0;
    long long strips_total = 0;
    for (int p = 0; p < npairs; ++p) {
        const DescSet* A = get_set(c, pairs[2 * p]);
        if (A && A->kind == SET_U8) strips_total += A->n_pad / kStripRows;
    }
    for (int p = 0; p < npairs; ++p) {
        const DescSet* A = get_set(c, pairs[2 * p]);
        const DescSet* B = get_set(c, pairs[2 * p + 1]);
        if (!A || !B) return fail(c, VSM_EINVAL, "pair %d names an unknown set", p);
        if (A->kind != SET_U8 || B->kind != SET_U8) return fail(c, VSM_EINVAL, "pair %d mixes set kinds", p);
        SiftPair& sp = pl.pairs[p];
        sp.fragA = static_cast<const int8_t*>(A->frag);
        sp.nrmA = static_cast<const int32_t*>(A->nrm);
        sp.fragB = static_cast<const int8_t*>(B->frag);
        sp.nrmB = static_cast<const int32_t*>(B->nrm);
        sp.nA = A->n; sp.nB = B->n; sp.nA_pad = A->n_pad; sp.nB_pad = B->n_pad;
        const bool empty = A->n == 0 || B->n == 0;
        sp.nstrips = empty ? 0 : A->n_pad / kStripRows;
        // (the forced tail form wants every train set in one chunk, also where a lone small pair would otherwise be
        // cut into several to fill the chip)
        const int chunk_cols = empty || c->u8_mode == VSM_U8_TILE_FUSED ? kChunkCols : chunk_cols_for(strips_total, B->n_pad);
        sp.chunk_cols = chunk_cols;
        sp.pad0 = 0;
        sp.nchunks = empty ? 0 : (B->n_pad + chunk_cols - 1) / chunk_cols;
        sp.row_off = (int64_t)pl.rows;
        sp.out0_off = (int64_t)pl.out0;
        sp.out1_off = (int64_t)pl.out1;
        pl.rows += (size_t)sp.nchunks * sp.nA_pad;
        pl.out0 += (size_t)sp.nA;
        pl.out1 += (size_t)sp.nB;
        for (int s = 0; s < sp.nstrips; ++s)
            for (int ch = 0; ch < sp.nchunks; ++ch) pl.items.push_back(SiftItem{p, (int16_t)s, (int16_t)ch});
        if (!empty)
            pl.max_vitems += (A->n + kVerifyBlock - 1) / kVerifyBlock;
    }
    if ((pl.rows + pl.out0) * kScratchPerRow > std::max(c->scratch_budget, kDefaultScratchBudget))
        return fail(c, VSM_EINVAL, "submission needs about %zu bytes of scratch (limit %zu); submit fewer pairs per call",
                    (pl.rows + pl.out0) * kScratchPerRow, std::max(c->scratch_budget, kDefaultScratchBudget));
    return VSM_OK;
}

struct RunOpts {
    bool want_dense = false, want_knn = false, want_dist = false, want_pack = true;
    // float submissions of one pair: the device records travel as kernel arguments instead of a copy, and the accept
    // kernel copies these words (device addresses; filled in by the caller once the buffers exist) to copy_dst
    bool by_args = false;
    uint32_t done_at = 0, done_seq = 0;   // with copy_dst: the word that reports the copy as done, and what it will hold
    std::function<void(WordSegments*)> copy_words;
    uint32_t* copy_dst = nullptr;
    bool packed_only = false;   // keep to the packed query pass whatever the rule
};

// Queues the whole u8 pipeline for one planned submission.
int run_u8(vsm_ctx* c, Plan& pl, const MatchRule& rule, const RunOpts& o)
{
    const int npairs = (int)pl.pairs.size();
    const int nitems = (int)pl.items.size();
    c->R().query_pairs = npairs;
    const int nvitems = rule.mutual ? pl.max_vitems : 0;
    // pair records and work items share one pinned block and one device block, [pairs | items | the verify
    // pass's items and their counter (written on the device)]: one copy command per submission
    const size_t pairs_bytes = (sizeof(SiftPair) * std::max(npairs, 1) + 255) & ~(size_t)255;
    HIPCHK(c, c->R().h_pairs.ensure(pairs_bytes + sizeof(SiftItem) * std::max(nitems, 1)));
    // (after them, the rescan pass's items, at most as many as the query pass has)
    HIPCHK(c, c->d_pairs.ensure(pairs_bytes + sizeof(SiftItem) * std::max(2 * nitems + nvitems, 1) + 64));
    const size_t rows = std::max<size_t>(pl.rows, 1);
    const size_t out0 = std::max<size_t>(pl.out0, 1), out1 = std::max<size_t>(pl.out1, 1);
    HIPCHK(c, c->d_rr1.ensure(rows * 4));
    HIPCHK(c, c->d_rrj.ensure(rows * 4));
    HIPCHK(c, c->d_rr2.ensure(rows * 4));
    // vbest (8) + i0, j1, d1, aux (4 each) + rescan list (8) + row list of the verify pass (4 + 4) per row, then
    // per-pair counts, largest accepted distances and row-list lengths, and the rescan counter
    // and the per-slice counts of the select and row-list passes (pairs of many rows are cut into slices)
    int max_rows = 0, max_chunks = 0;
    for (const SiftPair& sp : pl.pairs) {
        max_rows = std::max(max_rows, sp.nA);
        max_chunks = std::max(max_chunks, sp.nchunks);
    }
    const int slice_ints = std::max(std::max(sift_select_slices(max_rows), sift_verify_rows_slices(max_rows)),
                                    kChunkCols / kStripRows);
    HIPCHK(c, c->d_cand.ensure(out0 * 40 + sizeof(int32_t) * ((4 + (size_t)slice_ints) * std::max(npairs, 1) + 16)));
    if (!c->d_zero_tile.p) {
        HIPCHK(c, c->d_zero_tile.ensure(kTileBytes));
        HIPCHK(c, hipMemsetAsync(c->d_zero_tile.p, 0, kTileBytes, c->stream));
    }
    // The tile-minimum query pass leaves row and exact distances to a resolve pass over the candidates (every row
    // when the rule has no test that thins them out: nearest row only, e.g. the word of every descriptor); it does
    // not produce the raw 2-NN outputs, whose runner-up distance it knows only up to a tie.
    bool tile_path = c->u8_mode != VSM_U8_PACKED && !o.packed_only && !o.want_knn;
    // a bound on the image-0 rows the mutual check must look at exists only when a test keeps the candidates near
    const bool row_lists = rule.num2 > 0 || rule.max_dist >= 0;
    if (tile_path && c->u8_mode == VSM_U8_AUTO && c->u8_packed_left > 0 && nitems > 0) {
        // the last tile-minimum submission sent most of its rows to the rescan pass: the packed form is cheaper on such data
        --c->u8_packed_left;
        tile_path = false;
    }
    if (tile_path && c->u8_mode == VSM_U8_AUTO && npairs == 1 && nitems > 0 &&
        (double)pl.pairs[0].nA * (double)pl.pairs[0].nB <= (double)(1 << 29)) {
        // One pair of modest size per call is bound by its launches, and the packed form queues eight where the
        // tile-minimum form queues twelve (83 against 117 us on the stream for 8192 x 8192); the faster query pass
        // pays from some 23 000 x 23 000 rows on.
        tile_path = false;
    }
    if (tile_path) HIPCHK(c, c->d_rr3.ensure(rows * 4));   // third array of the rescan pass's per-(chunk, slot) results
    c->R().u8_stats = true;
    c->R().tile_path = tile_path && nitems > 0;
    c->R().rows0 = pl.out0;
    // select and resolve passes in the tail of the query pass: every pair at most one chunk of rows on either side
    // (a call with a pair or two is bound by its launches, and this form has one more of them)
    bool fused = tile_path && nitems > 0 && c->u8_mode != VSM_U8_TILE_SEPARATE &&
                 (npairs >= 8 || c->u8_mode == VSM_U8_TILE_FUSED);
    for (const SiftPair& sp : pl.pairs) fused = fused && sp.nchunks <= 1 && sp.nstrips <= kChunkCols / kStripRows;
    c->last_u8_path = nitems ? (tile_path ? (fused ? 2 : 0) : 1) : c->last_u8_path;

    HIPCHK(c, c->d_slot0.ensure(out0 * 4));
    HIPCHK(c, c->d_slot1.ensure(out0 * 4));
    if (o.want_dist) HIPCHK(c, c->d_slotd.ensure(out0 * 4));
    HIPCHK(c, c->d_counts.ensure(sizeof(int32_t) * std::max(npairs, 1)));
    // (two more words behind the offsets: work items and candidates of the rescan pass, fetched with them)
    HIPCHK(c, c->R().d_offsets.ensure(sizeof(int64_t) * (npairs + 2)));
    int32_t* const d_stats = reinterpret_cast<int32_t*>(c->R().d_offsets.as<int64_t>() + npairs + 1);
    if (o.want_pack) {
        HIPCHK(c, c->R().d_tight0.ensure(out0 * 4));
        HIPCHK(c, c->R().d_tight1.ensure(out0 * 4));
    }
    if (o.want_dense) {
        HIPCHK(c, c->d_dense0.ensure(out0 * 4));
        HIPCHK(c, c->d_dense1.ensure(out1 * 4));
    }
    if (o.want_knn) {
        HIPCHK(c, c->d_knnj.ensure(out0 * 4));
        HIPCHK(c, c->d_knn1.ensure(out0 * 4));
        HIPCHK(c, c->d_knn2.ensure(out0 * 4));
    }
    // The slot's pinned staging buffers may still feed the copy of the submission that last used them: wait for
    // that copy, not for the stream (a submission queued behind another one would otherwise leave the chip idle
    // between the two while the host catches up).
    if (c->R().staged_recorded) HIPCHK(c, hipEventSynchronize(c->R().staged));
    SiftItem* const d_items = reinterpret_cast<SiftItem*>(c->d_pairs.as<char>() + pairs_bytes);
    if (npairs) memcpy(c->R().h_pairs.p, pl.pairs.data(), sizeof(SiftPair) * npairs);
    if (nitems) memcpy(c->R().h_pairs.as<char>() + pairs_bytes, pl.items.data(), sizeof(SiftItem) * nitems);
    if (npairs == 1)
        // one pair: its record travels as a kernel argument and its work items are written on the device (a copy
        // command costs such a call more than ten microseconds on the stream, a small kernel three)
        launch_sift_plan_single(pl.pairs[0], c->d_pairs.as<SiftPair>(), d_items, c->stream,
                                !(tile_path && nitems > 0) ? d_stats : nullptr);   // (the packed form's rescan counters: zero)
    else if (npairs)
        HIPCHK(c, hipMemcpyAsync(c->d_pairs.p, c->R().h_pairs.p, nitems ? pairs_bytes + sizeof(SiftItem) * nitems : sizeof(SiftPair) * npairs,
                                 hipMemcpyHostToDevice, c->stream));
    if (npairs == 1) {
        c->R().staged_recorded = false;   // nothing was staged: no event on the stream either
    } else {
        HIPCHK(c, hipEventRecord(c->R().staged, c->stream));
        c->R().staged_recorded = true;
    }

    c->R().ev_valid = false;
    c->timing_slot = c->rs;
    c->R().ev_verify_valid = false;
    if (nitems) {
        if (c->timing && !fused) HIPCHK(c, hipEventRecord(c->R().ev0, c->stream));
        if (fused) {
            // launched below, once the candidate arrays are laid out
        } else if (tile_path)
            launch_sift_query_tile(c->d_pairs.as<SiftPair>(), d_items, nitems,
                                   c->d_rr1.as<int32_t>(), c->d_rrj.as<int32_t>(), c->d_rr2.as<int32_t>(), c->stream);
        else
            launch_sift_query(c->d_pairs.as<SiftPair>(), d_items, nitems,
                              c->d_rr1.as<int32_t>(), c->d_rrj.as<int32_t>(), c->d_rr2.as<int32_t>(),
                              rule.num2 > 0 || o.want_knn, c->stream);
        if (c->timing && !fused) {
            HIPCHK(c, hipEventRecord(c->R().ev1, c->stream));
            c->R().ev_valid = true;
        }
    }
    if (npairs) {
        SiftCandidates cand{};
        cand.vbest = c->d_cand.as<unsigned long long>();
        cand.i0 = reinterpret_cast<int32_t*>(cand.vbest + out0);
        cand.j1 = cand.i0 + out0;
        cand.d1 = cand.j1 + out0;
        cand.aux = cand.d1 + out0;
        cand.rescan_list = cand.aux + out0;
        int32_t* row_list = cand.rescan_list + 2 * out0;
        int32_t* row_codes = row_list + out0;
        cand.count = row_codes + out0;
        int32_t* dmax = cand.count + std::max(npairs, 1);
        int32_t* nvrows = dmax + std::max(npairs, 1);
        cand.rescan_cnt = nvrows + std::max(npairs, 1);
        int32_t* slice_counts = cand.rescan_cnt + std::max(npairs, 1) + 16;
        cand.rescan_items = d_items + nitems + nvitems + 8;   // behind the verify pass's items and their two counters
        cand.rescan_nitems = d_stats;
        cand.rs_d1 = c->d_rrj.as<int32_t>();   // the query pass's group and runner-up records are spent by then
        cand.rs_j1 = c->d_rr2.as<int32_t>();
        cand.rs_d2 = c->d_rr3.as<int32_t>();
        if (tile_path && row_lists) {   // otherwise nothing bounds the rows that matter: the verify pass streams them all
            cand.dmax = dmax;
            cand.vrows = row_list;
            cand.vcodes = row_codes;
            cand.nvrows = nvrows;
            cand.zero_tile = c->d_zero_tile.as<int8_t>();
        }
        SiftOutputs out{};
        out.slot_i0 = c->d_slot0.as<int32_t>();
        out.slot_i1 = c->d_slot1.as<int32_t>();
        out.slot_d = o.want_dist ? c->d_slotd.as<int32_t>() : nullptr;
        out.counts = c->d_counts.as<int32_t>();
        if (o.want_dense) {
            out.dense0 = c->d_dense0.as<int32_t>();
            out.dense1 = c->d_dense1.as<int32_t>();
        }
        if (o.want_knn) {
            out.knn_j1 = c->d_knnj.as<int32_t>();
            out.knn_d1 = c->d_knn1.as<int32_t>();
            out.knn_d2 = c->d_knn2.as<int32_t>();
        }
        if (fused) {
            if (c->timing) HIPCHK(c, hipEventRecord(c->R().ev0, c->stream));
            launch_sift_query_tile_fused(c->d_pairs.as<SiftPair>(), npairs, d_items, nitems, c->d_rr1.as<int32_t>(), rule,
                                         cand, slice_counts, c->stream);
            if (c->timing) {
                HIPCHK(c, hipEventRecord(c->R().ev1, c->stream));
                c->R().ev_valid = true;
            }
            launch_sift_fused_finish(c->d_pairs.as<SiftPair>(), npairs, nitems, rule, cand, slice_counts, c->stream);
            if (nvitems && row_lists)
                launch_sift_verify_rows(c->d_pairs.as<SiftPair>(), npairs, max_rows, c->d_rr1.as<int32_t>(), cand,
                                        slice_counts, c->stream);
        } else if (tile_path) {
            launch_sift_select_tile(c->d_pairs.as<SiftPair>(), npairs, max_rows, max_chunks, c->d_rr1.as<int32_t>(),
                                    c->d_rrj.as<int32_t>(), c->d_rr2.as<int32_t>(), rule, cand, slice_counts, c->stream);
            launch_sift_resolve(c->d_pairs.as<SiftPair>(), npairs, max_rows, nitems, rule, cand, c->stream);
            if (nvitems && row_lists)
                launch_sift_verify_rows(c->d_pairs.as<SiftPair>(), npairs, max_rows, c->d_rr1.as<int32_t>(), cand,
                                        slice_counts, c->stream);
        } else {
            launch_sift_select(c->d_pairs.as<SiftPair>(), npairs, c->d_rr1.as<int32_t>(),
                               c->d_rrj.as<int32_t>(), c->d_rr2.as<int32_t>(), rule, cand, out, c->stream);
        }
        if ((!tile_path || nitems == 0) && npairs != 1) HIPCHK(c, hipMemsetAsync(d_stats, 0, 2 * sizeof(int32_t), c->stream));
        if (nvitems) {
            if (c->timing) HIPCHK(c, hipEventRecord(c->R().ev2, c->stream));
            launch_sift_verify(c->d_pairs.as<SiftPair>(), npairs, d_items + nitems, nvitems,
                               reinterpret_cast<int*>(d_items + nitems + nvitems), cand, c->stream);
            if (c->timing) {
                HIPCHK(c, hipEventRecord(c->R().ev3, c->stream));
                c->R().ev_verify_valid = true;
            }
        }
        launch_sift_confirm(c->d_pairs.as<SiftPair>(), npairs, cand, nvitems ? 1 : 0, out, c->stream);
        if (o.want_pack)
            launch_pack(c->d_pairs.as<SiftPair>(), npairs, c->d_counts.as<int32_t>(),
                        c->R().d_offsets.as<int64_t>(), c->d_slot0.as<int32_t>(), c->d_slot1.as<int32_t>(),
                        c->R().d_tight0.as<int32_t>(), c->R().d_tight1.as<int32_t>(), c->stream);
    }
    else
        HIPCHK(c, hipMemsetAsync(d_stats, 0, 2 * sizeof(int32_t), c->stream));
    HIPCHK(c, hipGetLastError());
    return VSM_OK;
}

// The rescan counters of the u8 submission that last used the current result slot, as fetched from behind its
// offsets: kept for vsm_last_u8_rescanned, and (VSM_U8_AUTO) the next submissions go to the packed form when more than
// one row in eight needed the rescan pass.
void note_u8_stats(vsm_ctx* c, const int32_t* stats)
{
    c->last_u8_rescanned = stats[1];
    if (c->R().tile_path && c->u8_mode == VSM_U8_AUTO && (size_t)stats[1] * 8 > c->R().rows0) c->u8_packed_left = 31;
}

MatchRule make_rule(const vsm_params* p)
{
    MatchRule r;
    r.num2 = p->ratio_num > 0 ? (int64_t)p->ratio_num * p->ratio_num : 0;
    r.den2 = (int64_t)p->ratio_den * p->ratio_den;
    r.max_dist = p->max_dist_sq;
    r.mutual = p->mutual ? 1 : 0;
    return r;
}

int set_kind_of_pair(vsm_ctx* c, int set0, int set1, SetKind* kind)
{
    const DescSet* A = get_set(c, set0);
    const DescSet* B = get_set(c, set1);
    if (!A || !B) return fail(c, VSM_EINVAL, "unknown set id (%d, %d)", set0, set1);
    if (A->kind != B->kind) return fail(c, VSM_EINVAL, "sets %d and %d hold different descriptor kinds", set0, set1);
    if (A->kind != SET_U8 && A->kind != SET_F32) return fail(c, VSM_EINVAL, "sets %d and %d hold no descriptors", set0, set1);
    if (A->kind == SET_F32 && A->dim != B->dim) return fail(c, VSM_EINVAL, "sets %d and %d differ in dimension", set0, set1);
    *kind = A->kind;
    return VSM_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------
extern "C" vsm_ctx* vsm_create(int device)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        fail(nullptr, VSM_ENODEV, "no HIP device available (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        fail(nullptr, VSM_EINVAL, "device %d out of range (0..%d)", device, ndev - 1);
        return nullptr;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        fail(nullptr, VSM_ENODEV, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
        return nullptr;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        fail(nullptr, VSM_ENODEV, "hipGetDeviceProperties: %s", hipGetErrorString(e));
        return nullptr;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fail(nullptr, VSM_ENODEV, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return nullptr;
    }
    // this device's copy of the scratch array of vsm_match_pair_noalloc (the device is current here)
    unsigned char* scratch_dev = nullptr;
    if (device < kMaxDevices) {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        if (!g_scratch_ptr[device]) {
            void* ptr = nullptr;
            if ((e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch))) != hipSuccess || !ptr) {
                fail(nullptr, VSM_ENODEV, "hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e));
                return nullptr;
            }
            hipEvent_t ev = nullptr;
            if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) {
                fail(nullptr, VSM_ENODEV, "hipEventCreate on device %d: %s", device, hipGetErrorString(e));
                return nullptr;
            }
            g_scratch_ptr[device] = ptr;
            g_scratch_done[device] = ev;
        }
        scratch_dev = static_cast<unsigned char*>(g_scratch_ptr[device]);
    }
    vsm_ctx* c = new vsm_ctx();
    c->device = device;
    c->noalloc_scratch = scratch_dev;
    c->noalloc_done = device < kMaxDevices ? g_scratch_done[device] : nullptr;
    c->coarse_wgs = std::max(2 * prop.multiProcessorCount, 64);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->slots[0].ev0)) != hipSuccess || (e = hipEventCreate(&c->slots[0].ev1)) != hipSuccess ||
        (e = hipEventCreate(&c->slots[0].ev2)) != hipSuccess || (e = hipEventCreate(&c->slots[0].ev3)) != hipSuccess ||
        (e = hipEventCreate(&c->slots[1].ev0)) != hipSuccess || (e = hipEventCreate(&c->slots[1].ev1)) != hipSuccess ||
        (e = hipEventCreate(&c->slots[1].ev2)) != hipSuccess || (e = hipEventCreate(&c->slots[1].ev3)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->slots[0].packed, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->slots[1].packed, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->slots[0].staged, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->slots[1].staged, hipEventDisableTiming)) != hipSuccess) {
        fail(nullptr, VSM_ENODEV, "stream/event creation: %s", hipGetErrorString(e));
        vsm_destroy(c);   // releases whatever was created
        return nullptr;
    }
    return c;
}

extern "C" void vsm_destroy(vsm_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (auto& s : c->sets) {
        if (s.frag) (void)hipFree(s.frag);
        if (s.nrm) (void)hipFree(s.nrm);
        if (s.shadow) (void)hipFree(s.shadow);
    }
    DevBuf* bufs[] = {&c->d_zero_tile, &c->d_pairs, &c->d_rr1, &c->d_rrj, &c->d_rr2, &c->d_rr3, &c->d_colpart, &c->d_cand,
                      &c->d_slot0, &c->d_slot1, &c->d_slotd, &c->d_counts, &c->d_dense0, &c->d_dense1,
                      &c->d_knnj, &c->d_knn1, &c->d_knn2, &c->d_stage, &c->d_coarse,
                      &c->d_probs, &c->d_geom, &c->d_short};
    for (DevBuf* b : bufs) b->release();
    for (auto& r : c->slots) {
        r.d_offsets.release();
        r.d_tight0.release();
        r.d_tight1.release();
        r.h_pairs.release();
        r.h_off.release();
        r.h_res.release();
        if (r.packed) (void)hipEventDestroy(r.packed);
        if (r.staged) (void)hipEventDestroy(r.staged);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    c->h_small.release();
    c->h_probs.release();
    c->h_gather.release();
    for (auto& r : c->slots)
        for (hipEvent_t ev : {r.ev0, r.ev1, r.ev2, r.ev3})
            if (ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* vsm_last_error(const vsm_ctx* c)
{
    return c ? c->err.c_str() : g_create_error.c_str();
}

extern "C" int vsm_synchronize(vsm_ctx* c)
{
    if (!c) return VSM_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VSM_OK;
}

extern "C" void* vsm_stream(vsm_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int vsm_set_timing(vsm_ctx* c, int on)
{
    if (!c) return VSM_EINVAL;
    c->timing = on ? 1 : 0;
    return VSM_OK;
}

extern "C" float vsm_last_verify_ms(vsm_ctx* c)
{
    if (!c || !c->slots[c->timing_slot].ev_verify_valid) return -1.0f;
    if (hipEventSynchronize(c->slots[c->timing_slot].ev3) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, c->slots[c->timing_slot].ev2, c->slots[c->timing_slot].ev3) != hipSuccess) return -1.0f;
    return ms;
}

extern "C" float vsm_last_kernel_ms(vsm_ctx* c)
{
    if (!c || !c->slots[c->timing_slot].ev_valid) return -1.0f;
    if (hipEventSynchronize(c->slots[c->timing_slot].ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, c->slots[c->timing_slot].ev0, c->slots[c->timing_slot].ev1) != hipSuccess) return -1.0f;
    return ms;
}

// ---------------------------------------------------------------------------------
// descriptor sets
// ---------------------------------------------------------------------------------
static int upload_u8_common(vsm_ctx* c, const void* src, int n, bool src_on_device)
{
    if (!c) return VSM_EINVAL;
    if (n < 0 || (n > 0 && !src)) return fail(c, VSM_EINVAL, "bad descriptor buffer (n=%d)", n);
    // work items name blocks of kVerifyBlock rows in 16 bits
    if (n > 32767 * kVerifyBlock) return fail(c, VSM_EINVAL, "a u8 set holds at most %d rows (got %d)", 32767 * kVerifyBlock, n);
    // the transform kernel reads the rows with 16-byte loads (host buffers go through an aligned staging copy)
    if (src_on_device && n > 0 && (reinterpret_cast<uintptr_t>(src) & 15) != 0)
        return fail(c, VSM_EINVAL, "device descriptor buffer %p is not 16-byte aligned", src);
    HIPCHK(c, hipSetDevice(c->device));
    DescSet s;
    s.kind = SET_U8;
    s.n = n;
    s.n_pad = round_up(n, kRowPad);
    s.dim = kDescDim;
    if (s.n_pad > 0) {
        HIPCHK(c, hipMalloc(&s.frag, (size_t)s.n_pad * kDescDim));
        // nrm8c[n_pad] followed by the half-distance constants cq[n_pad] of the tile-minimum query pass
        hipError_t e = hipMalloc(&s.nrm, (size_t)s.n_pad * 2 * sizeof(int32_t));
        if (e != hipSuccess) {
            (void)hipFree(s.frag);
            return fail(c, VSM_ENOMEM, "hipMalloc(norms): %s", hipGetErrorString(e));
        }
        const uint8_t* dsrc = static_cast<const uint8_t*>(src);
        if (!src_on_device) {
            e = c->d_stage.ensure((size_t)n * kDescDim);
            if (e == hipSuccess)
                e = hipMemcpyAsync(c->d_stage.p, src, (size_t)n * kDescDim, hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess) {
                (void)hipFree(s.frag);
                (void)hipFree(s.nrm);
                return fail(c, VSM_ENODEV, "staging copy: %s", hipGetErrorString(e));
            }
            dsrc = c->d_stage.as<uint8_t>();
        }
        launch_sift_transform(dsrc, n, s.n_pad, static_cast<int8_t*>(s.frag), static_cast<int32_t*>(s.nrm), c->stream);
        // the staging buffer (or the caller's host buffer) may be reused right after return
        e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            (void)hipFree(s.frag);
            (void)hipFree(s.nrm);
            return fail(c, VSM_ENODEV, "descriptor transform: %s", hipGetErrorString(e));
        }
    }
    const int id = new_set_slot(c);
    c->sets[id] = s;
    return id;
}

extern "C" int vsm_upload_u8(vsm_ctx* c, const uint8_t* des, int n) { return upload_u8_common(c, des, n, false); }
extern "C" int vsm_upload_u8_device(vsm_ctx* c, const void* des_dev, int n) { return upload_u8_common(c, des_dev, n, true); }

static int upload_f32_common(vsm_ctx* c, const void* src, int n, int dim, int layout, bool src_on_device)
{
    if (!c) return VSM_EINVAL;
    if (n < 0 || (n > 0 && !src)) return fail(c, VSM_EINVAL, "bad descriptor buffer (n=%d)", n);
    if (dim <= 0 || dim > 256) return fail(c, VSM_EINVAL, "dim %d must be in 1..256", dim);
    if (layout != VSM_LAYOUT_ROWS && layout != VSM_LAYOUT_DIM_MAJOR) return fail(c, VSM_EINVAL, "unknown layout %d", layout);
    if (src_on_device && n > 0 && (reinterpret_cast<uintptr_t>(src) & 3) != 0)
        return fail(c, VSM_EINVAL, "device descriptor buffer %p is not 4-byte aligned", src);
    HIPCHK(c, hipSetDevice(c->device));
    DescSet s;
    s.kind = SET_F32;
    s.n = n;
    s.n_pad = round_up(n, kF32RowPad);
    s.dim = dim;
    if (s.n_pad > 0) {
        HIPCHK(c, hipMalloc(&s.frag, (size_t)s.n_pad * f32_dimp(dim) * sizeof(float)));
        hipError_t e = hipMalloc(&s.nrm, ((size_t)s.n_pad + 64) * sizeof(float));
        if (e != hipSuccess) {
            (void)hipFree(s.frag);
            return fail(c, VSM_ENOMEM, "hipMalloc(norms): %s", hipGetErrorString(e));
        }
        e = hipMalloc(&s.shadow, bf16_shadow_bytes(s.n_pad, dim));
        if (e != hipSuccess) {
            (void)hipFree(s.frag);
            (void)hipFree(s.nrm);
            return fail(c, VSM_ENOMEM, "hipMalloc(bf16 shadow): %s", hipGetErrorString(e));
        }
        const float* dsrc = static_cast<const float*>(src);
        if (!src_on_device) {
            e = c->d_stage.ensure((size_t)n * dim * sizeof(float));
            if (e == hipSuccess)
                e = hipMemcpyAsync(c->d_stage.p, src, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess) {
                (void)hipFree(s.frag);
                (void)hipFree(s.nrm);
                (void)hipFree(s.shadow);
                return fail(c, VSM_ENODEV, "staging copy: %s", hipGetErrorString(e));
            }
            dsrc = c->d_stage.as<float>();
        }
        launch_f32_transform(dsrc, n, s.n_pad, dim, layout == VSM_LAYOUT_DIM_MAJOR,
                             static_cast<float*>(s.frag), static_cast<float*>(s.nrm), c->stream);
        launch_bf16_transform(dsrc, n, s.n_pad, dim, layout == VSM_LAYOUT_DIM_MAJOR, s.shadow,
                              static_cast<const float*>(s.nrm), c->stream);
        e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            (void)hipFree(s.frag);
            (void)hipFree(s.nrm);
            (void)hipFree(s.shadow);
            return fail(c, VSM_ENODEV, "descriptor transform: %s", hipGetErrorString(e));
        }
    }
    const int id = new_set_slot(c);
    c->sets[id] = s;
    return id;
}

extern "C" int vsm_upload_f32(vsm_ctx* c, const float* des, int n, int dim, int layout)
{
    return upload_f32_common(c, des, n, dim, layout, false);
}
extern "C" int vsm_upload_f32_device(vsm_ctx* c, const void* des_dev, int n, int dim, int layout)
{
    return upload_f32_common(c, des_dev, n, dim, layout, true);
}

extern "C" int vsm_set_u8_mode(vsm_ctx* c, int mode)
{
    if (!c) return VSM_EINVAL;
    if (mode < VSM_U8_AUTO || mode > VSM_U8_TILE_FUSED) return fail(c, VSM_EINVAL, "unknown u8 mode %d", mode);
    c->u8_mode = mode;
    c->u8_packed_left = 0;   // what VSM_U8_AUTO learnt from earlier submissions is forgotten
    return VSM_OK;
}

extern "C" int vsm_last_u8_path(const vsm_ctx* c) { return c ? c->last_u8_path : VSM_EINVAL; }
extern "C" int vsm_last_u8_rescanned(const vsm_ctx* c) { return c ? c->last_u8_rescanned : VSM_EINVAL; }

extern "C" int vsm_release_set(vsm_ctx* c, int set)
{
    if (!c) return VSM_EINVAL;
    if (!get_set(c, set)) return fail(c, VSM_EINVAL, "unknown set id %d", set);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    DescSet& s = c->sets[set];
    if (s.frag) (void)hipFree(s.frag);
    if (s.nrm) (void)hipFree(s.nrm);
    if (s.shadow) (void)hipFree(s.shadow);
    s = DescSet();
    return VSM_OK;
}

extern "C" int vsm_set_count(const vsm_ctx* c, int set)
{
    const DescSet* s = get_set(c, set);
    return s ? s->n : VSM_EINVAL;
}

// ---------------------------------------------------------------------------------
// f32 helpers (one pair per submission)
// ---------------------------------------------------------------------------------
namespace {

struct F32Result {  // device pointers valid until the next submission
    int32_t *j1, *j2, *cidx;
    float *k1, *k2, *ck;
};

CoarseProblem make_problem(const DescSet* A, const DescSet* B, int metric)
{
    CoarseProblem p{};
    p.pr.fragA = static_cast<const float*>(A->frag);
    p.pr.nrmA = static_cast<const float*>(A->nrm);
    p.pr.fragB = static_cast<const float*>(B->frag);
    p.pr.nrmB = static_cast<const float*>(B->nrm);
    p.pr.nA = A->n; p.pr.nB = B->n; p.pr.nA_pad = A->n_pad; p.pr.nB_pad = B->n_pad;
    p.pr.dim = A->dim;
    p.pr.metric = metric;
    const Bf16Shadow sa = A->shadow_view(), sb = B->shadow_view();
    p.bfA = sa.rows;
    p.bfB = sb.rows;
    p.initB = metric == VSM_METRIC_L2 ? sb.init_l2 : sb.init_dot;
    p.maxnormB_sq = sb.maxnorm_sq;
    p.rowsA = sa.rows_f32;
    p.rowsB = sb.rows_f32;
    p.top1 = 0;
    p.key_floor = -INFINITY;
    return p;
}

// Plans `probs` (non-empty searches only), uploads them and queues the coarse + re-rank kernels.
// Results land in d_knnj / d_rrj (best and runner-up row) and d_knn1 / d_knn2 (their keys) at each
// problem's out_off; the overflow flag is cleared first and must be read back by the caller.
// One host-to-device copy carries the cleared flag, the problems and `extra_bytes` more of the caller's
// (filled through *extra_host before this returns is too late: pass a filler).
int queue_coarse(vsm_ctx* c, std::vector<CoarseProblem>& probs, int dim, size_t min_rows, size_t extra_bytes = 0,
                 const std::function<void(char*)>& fill_extra = nullptr, char** extra_dev = nullptr)
{
    const CoarseTotals t = plan_coarse(probs.data(), (int)probs.size(), c->coarse_wgs);
    const size_t rows = std::max<size_t>(std::max<size_t>((size_t)t.out_rows, min_rows), 1);
    HIPCHK(c, c->d_knnj.ensure(rows * 4));
    HIPCHK(c, c->d_rrj.ensure(rows * 4));
    HIPCHK(c, c->d_knn1.ensure(rows * 4));
    HIPCHK(c, c->d_knn2.ensure(rows * 4));
    c->f32_clean.flag = nullptr;
    HIPCHK(c, c->d_coarse.ensure(coarse_scratch_bytes(t)));
    HIPCHK(c, c->h_small.ensure(64));
    // block layout: [flag, 256 bytes][problems][extra]
    const size_t pbytes = (sizeof(CoarseProblem) * probs.size() + 255) & ~(size_t)255;
    const size_t bytes = 256 + pbytes + extra_bytes;
    HIPCHK(c, c->h_probs.ensure(bytes));
    HIPCHK(c, c->d_probs.ensure(bytes));
    char* h = c->h_probs.as<char>();
    memset(h, 0, 256);
    if (!probs.empty()) memcpy(h + 256, probs.data(), sizeof(CoarseProblem) * probs.size());
    if (extra_bytes && fill_extra) fill_extra(h + 256 + pbytes);
    HIPCHK(c, hipMemcpyAsync(c->d_probs.p, h, bytes, hipMemcpyHostToDevice, c->stream));
    c->flag_dev = c->d_probs.as<int>();
    if (extra_dev) *extra_dev = c->d_probs.as<char>() + 256 + pbytes;
    if (probs.empty()) return VSM_OK;
    const CoarseProblem* d_probs = reinterpret_cast<const CoarseProblem*>(c->d_probs.as<char>() + 256);
    const bool finished = launch_bf16_batch(d_probs, (int)probs.size(), dim, t,
                      c->d_knnj.as<int32_t>(), c->d_rrj.as<int32_t>(), c->d_knn1.as<float>(), c->d_knn2.as<float>(),
                      c->d_coarse.p, c->flag_dev, c->stream);
    if (probs[0].pr.metric == VSM_METRIC_L2 && !finished) {   // one metric per submission
        int max_rows = 0;
        for (const CoarseProblem& p : probs) max_rows = std::max(max_rows, p.pr.nA);
        launch_f32_l2_finish(d_probs, (int)probs.size(), max_rows, nullptr, c->d_knnj.as<int32_t>(), c->d_rrj.as<int32_t>(),
                             c->d_knn1.as<float>(), c->d_knn2.as<float>(), c->stream);
    }
    return VSM_OK;
}

// reads the overflow flag of the last queue_coarse (synchronises the stream)
// submissions up to this size hand their result to pinned host memory themselves (see ResultSlot::h_res)
constexpr int kSmallResultPairs = 4096;
constexpr size_t kSmallResultRows = 262144;
constexpr size_t kSmallResultRowsF32 = 1 << 21;   // float lists are thin (a score floor, a mutual check): more rows

// Queues, behind the pack kernel of the submission in the current slot, the kernel that copies [offsets and the two
// counters behind them][idx0][idx1][flag word] to the slot's pinned block; only as many list words as there are
// matches (a count the device has) cross the bus.  flag (may be NULL): a device word to bring along, e.g. the
// survivor-overflow flag of a float submission.  No-op for submissions above the limits.
int queue_small_result(vsm_ctx* c, int npairs, size_t out0, size_t row_limit, const int* flag)
{
    vsm_ctx::ResultSlot& R = c->R();
    R.res_on_host = false;
    if (npairs <= 0 || npairs > kSmallResultPairs || out0 > row_limit) return VSM_OK;
    const size_t off_words = 2 * ((size_t)npairs + 2), rows0 = std::max<size_t>(out0, 1);
    HIPCHK(c, R.h_res.ensure((off_words + 2 * rows0 + 1) * 4));
    WordSegments seg{};
    seg.src[0] = R.d_offsets.as<uint32_t>(); seg.count[0] = (int)off_words;
    seg.src[1] = R.d_tight0.as<uint32_t>(); seg.count[1] = (int)rows0;
    seg.src[2] = R.d_tight1.as<uint32_t>(); seg.count[2] = (int)rows0;
    seg.n = 3;
    seg.lists = 6;
    if (flag) {
        seg.src[3] = reinterpret_cast<const uint32_t*>(flag); seg.count[3] = 1;
        seg.n = 4;
    }
    launch_gather_words(seg, R.h_res.as<uint32_t>(), c->stream, R.d_offsets.as<int64_t>() + npairs);
    R.res_on_host = true;
    R.res_rows0 = rows0;
    R.res_pairs = npairs;
    return VSM_OK;
}

int coarse_overflowed(vsm_ctx* c, bool* overflow)
{
    int* flag = c->h_small.as<int>() + 8;
    HIPCHK(c, hipMemcpyAsync(flag, c->flag_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    *overflow = *flag != 0;
    return VSM_OK;
}

// One search (optionally also the reverse one for a mutual check); A and B are non-empty.
int run_f32_knn(vsm_ctx* c, const DescSet* A, const DescSet* B, int metric, bool want_col, F32Result* r)
{
    const size_t rows = (size_t)A->n_pad + B->n_pad;
    c->R().ev_valid = false;
    c->timing_slot = c->rs;
    int rc;
    if (c->f32_mode == 0) {
        // bf16 coarse pass + exact re-rank; the train side of a mutual check is the same search
        // with the roles swapped
        std::vector<CoarseProblem> probs;
        probs.push_back(make_problem(A, B, metric));
        if (want_col) probs.push_back(make_problem(B, A, metric));
        if (c->timing) HIPCHK(c, hipEventRecord(c->R().ev0, c->stream));
        if ((rc = queue_coarse(c, probs, A->dim, rows)) != VSM_OK) return rc;
        if (c->timing) HIPCHK(c, hipEventRecord(c->R().ev1, c->stream));
        bool overflow = false;
        if ((rc = coarse_overflowed(c, &overflow)) != VSM_OK) return rc;
        if (!overflow) {
            r->j1 = c->d_knnj.as<int32_t>();
            r->j2 = c->d_rrj.as<int32_t>();
            r->k1 = c->d_knn1.as<float>();
            r->k2 = c->d_knn2.as<float>();
            r->cidx = r->j1 + A->n_pad;
            r->ck = r->k1 + A->n_pad;
            c->R().ev_valid = c->timing != 0;
            c->last_f32_path = 0;
            return VSM_OK;
        }
        // a survivor list overflowed (heavily tied or adversarially ordered scores): exact kernel
    }
    HIPCHK(c, c->d_knnj.ensure(rows * 4));
    HIPCHK(c, c->d_rrj.ensure(rows * 4));
    HIPCHK(c, c->d_knn1.ensure(rows * 4));
    HIPCHK(c, c->d_knn2.ensure(rows * 4));
    r->j1 = c->d_knnj.as<int32_t>();
    r->j2 = c->d_rrj.as<int32_t>();
    r->k1 = c->d_knn1.as<float>();
    r->k2 = c->d_knn2.as<float>();
    r->cidx = r->j1 + A->n_pad;
    r->ck = r->k1 + A->n_pad;
    CoarseProblem p = make_problem(A, B, metric);
    p.out_off = 0;
    if (c->timing) HIPCHK(c, hipEventRecord(c->R().ev0, c->stream));
    if (metric == VSM_METRIC_L2) {
        // the train side of a mutual check is the same search with the roles exchanged, as on the fast path:
        // both directions get their two best rows ordered in difference form
        HIPCHK(c, c->d_colpart.ensure(std::max(f32_scratch_bytes(p.pr, false), f32_scratch_bytes(make_problem(B, A, metric).pr, false))));
        launch_f32_knn(p.pr, r->j1, r->j2, r->k1, r->k2, nullptr, nullptr, c->d_colpart.p, c->stream);
        launch_f32_l2_finish(nullptr, 1, A->n, &p, r->j1, r->j2, r->k1, r->k2, c->stream);
        if (want_col) {
            CoarseProblem q = make_problem(B, A, metric);
            q.out_off = A->n_pad;
            launch_f32_knn(q.pr, r->cidx, r->j2 + A->n_pad, r->ck, r->k2 + A->n_pad, nullptr, nullptr, c->d_colpart.p, c->stream);
            launch_f32_l2_finish(nullptr, 1, B->n, &q, r->j1, r->j2, r->k1, r->k2, c->stream);
        }
    } else {
        HIPCHK(c, c->d_colpart.ensure(f32_scratch_bytes(p.pr, want_col)));
        launch_f32_knn(p.pr, r->j1, r->j2, r->k1, r->k2, want_col ? r->cidx : nullptr, want_col ? r->ck : nullptr,
                       c->d_colpart.p, c->stream);
    }
    if (c->timing) {
        HIPCHK(c, hipEventRecord(c->R().ev1, c->stream));
        c->R().ev_valid = true;
    }
    HIPCHK(c, hipGetLastError());
    c->last_f32_path = 1;
    return VSM_OK;
}

int match_pair_f32(vsm_ctx* c, const DescSet* A, const DescSet* B, const vsm_params* p, int metric,
                   int32_t* matches0, int32_t* matches1, float* best0)
{
    // dense result on the host; the accept tests are O(n) and run here
    for (int i = 0; i < A->n; ++i) matches0[i] = -1;
    if (matches1) for (int j = 0; j < B->n; ++j) matches1[j] = -1;
    if (A->n == 0 || B->n == 0) return 0;
    F32Result r;
    int rc = run_f32_knn(c, A, B, metric, true, &r);
    if (rc) return rc;
    std::vector<int32_t> j1(A->n), ci(B->n);
    std::vector<float> k1(A->n), k2(A->n);
    HIPCHK(c, hipMemcpyAsync(j1.data(), r.j1, sizeof(int32_t) * A->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(k1.data(), r.k1, sizeof(float) * A->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(k2.data(), r.k2, sizeof(float) * A->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(ci.data(), r.cidx, sizeof(int32_t) * B->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int nm = 0;
    for (int i = 0; i < A->n; ++i) {
        if (best0) best0[i] = k1[i];
        const int j = j1[i];
        if (j < 0) continue;
        bool ok = true;
        if (metric == VSM_METRIC_DOT) {
            ok = k1[i] > p->min_score;
        } else {
            if (p->max_dist_sq_f >= 0.0f && k1[i] > p->max_dist_sq_f) ok = false;
            if (ok && p->ratio_num > 0 && B->n >= 2) {
                const double lhs = (double)k1[i] * (double)p->ratio_den * (double)p->ratio_den;
                const double rhs = (double)k2[i] * (double)p->ratio_num * (double)p->ratio_num;
                ok = lhs < rhs;
            }
        }
        if (ok && p->mutual && ci[j] != i) ok = false;
        if (!ok) continue;
        matches0[i] = j;
        if (matches1 && matches1[j] < 0) matches1[j] = i;
        ++nm;
    }
    return nm;
}

// ---- submissions of float pairs: coarse + re-rank for every search, accept tests on the device ----
struct F32Plan {
    std::vector<CoarseProblem> probs;   // image 0 -> image 1 searches, then (mutual check) the reverse ones
    int nfwd = 0;                       // how many of them are forward searches
    std::vector<SiftPair> pairs;     // only the counts and output offsets are used
    std::vector<F32PairRef> refs;    // problem indices while planning, row offsets once planned
    size_t out0 = 0, out1 = 0;
    double work = 0;                 // sum of rows x rows over the forward searches
    int dim = 0;
};

int plan_f32(vsm_ctx* c, const int32_t* pairs, int npairs, int metric, bool mutual, F32Plan& pl)
{
    pl.pairs.resize(npairs);
    pl.refs.resize(npairs);
    for (int p = 0; p < npairs; ++p) {
        const DescSet* A = get_set(c, pairs[2 * p]);
        const DescSet* B = get_set(c, pairs[2 * p + 1]);
        if (!A || !B) return fail(c, VSM_EINVAL, "pair %d names an unknown set", p);
        if (A->kind != SET_F32 || B->kind != SET_F32) return fail(c, VSM_EINVAL, "pair %d mixes set kinds", p);
        if (A->dim != B->dim) return fail(c, VSM_EINVAL, "pair %d: sets differ in dimension", p);
        if (pl.dim == 0) pl.dim = A->dim;
        if (f32_dimp(A->dim) != f32_dimp(pl.dim))
            return fail(c, VSM_EINVAL, "pair %d: one submission takes one padded dimension (64, 128 or 256)", p);
        SiftPair sp{};
        sp.nA = A->n; sp.nB = B->n; sp.nA_pad = A->n_pad; sp.nB_pad = B->n_pad;
        sp.out0_off = (int64_t)pl.out0;
        sp.out1_off = (int64_t)pl.out1;
        pl.out0 += (size_t)A->n;
        pl.out1 += (size_t)B->n;
        pl.pairs[p] = sp;
        F32PairRef rf{-1, -1};
        if (A->n > 0 && B->n > 0) {
            rf.fwd_off = (int64_t)pl.probs.size();
            pl.probs.push_back(make_problem(A, B, metric));
            pl.work += (double)A->n * (double)B->n;
        }
        pl.refs[p] = rf;
    }
    pl.nfwd = (int)pl.probs.size();
    if (mutual)
        for (int p = 0; p < npairs; ++p) {
            if (pl.refs[p].fwd_off < 0) continue;
            pl.refs[p].bwd_off = (int64_t)pl.probs.size();
            pl.probs.push_back(make_problem(get_set(c, pairs[2 * p + 1]), get_set(c, pairs[2 * p]), metric));
        }
    if (pl.dim == 0) pl.dim = 256;
    if (pl.probs.size() > 65535)   // one grid row of the re-rank kernel per search
        return fail(c, VSM_EINVAL, "%zu searches in one submission (limit 65535); submit fewer pairs per call", pl.probs.size());
    return VSM_OK;
}

F32Rule make_f32_rule(const vsm_params* p, int metric)
{
    F32Rule r;
    r.metric = metric;
    r.mutual = p->mutual ? 1 : 0;
    r.ratio_num = p->ratio_num;
    r.ratio_den = p->ratio_den;
    r.min_score = p->min_score;
    r.max_dist = p->max_dist_sq_f;
    return r;
}

// Queues the whole float pipeline for one planned submission; the caller reads the overflow flag.
//
// Dot ranking: only the best row of a query is ever asked for here, and no row at or below the score
// threshold can be accepted, so the searches run in their top-1 form with that floor.
// Mutual check of a large submission whose tests reject most rows: the forward searches and the accept tests
// run first, and the reverse searches then ask only about the train rows the surviving candidates name (the
// candidate list of a pair, as it lies in the output slots, is the reverse search's query list).  Small
// submissions, and rules that pass nearly every row, run both directions in one launch instead.
int run_f32_batch(vsm_ctx* c, F32Plan& pl, const F32Rule& rule, const RunOpts& o)
{
    const int npairs = (int)pl.pairs.size();
    const int nprobs = (int)pl.probs.size(), nfwd = pl.nfwd, nrev = nprobs - nfwd;
    const bool dot = rule.metric == VSM_METRIC_DOT;
    const bool selective = dot ? rule.min_score > -INFINITY : (rule.ratio_num > 0 || rule.max_dist >= 0.0f);
    const bool two_phase = nrev > 0 && (c->f32_reverse == VSM_F32_REVERSE_CANDIDATES ||
                                        (c->f32_reverse == VSM_F32_REVERSE_AUTO && selective && pl.work >= (double)(1 << 28)));
    // Dot ranking, candidates only: the accept test is a threshold on the very score the searches rank by, so a row
    // that scores at least what a candidate scores in the candidate's column passes it with its own best score
    // and is a candidate itself; the reverse searches then look at the candidates' rows only (their list, as it
    // lies in the output slots of image 0, is ascending).  Not for L2: its results are re-ordered by the
    // difference form afterwards, and the two orders need not agree on rows that were left out.
    const bool db_lists = two_phase && dot;
    c->R().ev_valid = false;
    c->R().u8_stats = false;
    c->timing_slot = c->rs;
    c->last_f32_reverse = nrev == 0 ? -1 : two_phase ? VSM_F32_REVERSE_CANDIDATES : VSM_F32_REVERSE_ALL;
    c->f32_copied_out = false;
    c->f32_watch_word = false;

    // output slots first: the candidate lists in them are the query lists of the reverse searches
    const size_t n0 = std::max<size_t>(pl.out0, 1), n1 = std::max<size_t>(pl.out1, 1);
    HIPCHK(c, c->d_slot0.ensure(n0 * 4));
    HIPCHK(c, c->d_slot1.ensure(n0 * 4));
    if (o.want_dist) HIPCHK(c, c->d_slotd.ensure(n0 * 4));
    HIPCHK(c, c->d_counts.ensure(sizeof(int32_t) * std::max(npairs, 1)));
    HIPCHK(c, c->R().d_offsets.ensure(sizeof(int64_t) * (npairs + 1)));
    if (o.want_pack) {
        HIPCHK(c, c->R().d_tight0.ensure(n0 * 4));
        HIPCHK(c, c->R().d_tight1.ensure(n0 * 4));
    }
    if (o.want_dense) {
        HIPCHK(c, c->d_dense0.ensure(n0 * 4));
        HIPCHK(c, c->d_dense1.ensure(n1 * 4));
    }

    for (CoarseProblem& p : pl.probs) {
        p.top1 = dot ? 1 : 0;
        p.key_floor = dot ? rule.min_score : -INFINITY;
    }
    if (two_phase)
        for (int p = 0; p < npairs; ++p) {
            const F32PairRef& rf = pl.refs[p];
            if (rf.bwd_off < 0) continue;
            CoarseProblem& q = pl.probs[rf.bwd_off];
            // capacity of the list = the forward search's query rows
            q.pr.nA = pl.pairs[p].nA;
            q.pr.nA_pad = pl.pairs[p].nA_pad;
            q.qlist = c->d_slot1.as<int32_t>() + pl.pairs[p].out0_off;
            q.qcount = c->d_counts.as<int32_t>() + p;
            if (db_lists) {
                q.dblist = c->d_slot0.as<int32_t>() + pl.pairs[p].out0_off;
                q.dbcount = q.qcount;
            }
        }
    // geometry and offsets: one group when both directions go in one launch, else two
    CoarseTotals t1, t2;
    if (two_phase) {
        t1 = plan_coarse(pl.probs.data(), nfwd, c->coarse_wgs);
        t2 = plan_coarse(pl.probs.data() + nfwd, nrev, c->coarse_wgs, t1.out_rows);
    } else {
        t1 = plan_coarse(pl.probs.data(), nprobs, c->coarse_wgs);
        t2 = t1;
    }
    const size_t rows = std::max<size_t>((size_t)t2.out_rows, 1);
    HIPCHK(c, c->d_knnj.ensure(rows * 4));
    HIPCHK(c, c->d_rrj.ensure(rows * 4));
    HIPCHK(c, c->d_knn1.ensure(rows * 4));
    HIPCHK(c, c->d_knn2.ensure(rows * 4));
    HIPCHK(c, c->d_coarse.ensure(std::max(coarse_scratch_bytes(t1), coarse_scratch_bytes(t2))));
    HIPCHK(c, c->h_small.ensure(64));
    for (F32PairRef& rf : pl.refs) {
        if (rf.fwd_off >= 0) rf.fwd_off = pl.probs[rf.fwd_off].out_off;
        if (rf.bwd_off >= 0) rf.bwd_off = pl.probs[rf.bwd_off].out_off;
    }
    // one host-to-device copy: [flag, 256 bytes][problems][pair records][references to the searches]
    const size_t pbytes = (sizeof(CoarseProblem) * (size_t)nprobs + 255) & ~(size_t)255;
    const size_t pair_bytes = (sizeof(SiftPair) * (size_t)npairs + 255) & ~(size_t)255;
    const size_t ref_bytes = sizeof(F32PairRef) * (size_t)npairs;
    const size_t bytes = 256 + pbytes + pair_bytes + ref_bytes;
    HIPCHK(c, c->h_probs.ensure(bytes));
    HIPCHK(c, c->d_probs.ensure(bytes));
    char* h = c->h_probs.as<char>();
    memset(h, 0, 256);
    if (nprobs) memcpy(h + 256, pl.probs.data(), sizeof(CoarseProblem) * (size_t)nprobs);
    if (npairs) {
        memcpy(h + 256 + pbytes, pl.pairs.data(), sizeof(SiftPair) * (size_t)npairs);
        memcpy(h + 256 + pbytes + pair_bytes, pl.refs.data(), ref_bytes);
    }
    if (c->timing) HIPCHK(c, hipEventRecord(c->R().ev0, c->stream));
    // One pair: its records travel as arguments of the coarse kernel, and the previous such call has left the flag
    // and the shared words zeroed (its accept kernel does), so the searches are the first thing on the stream.
    const bool by_args = o.by_args && npairs == 1 && nprobs >= 1 && nprobs <= 2 && !two_phase && t1.coarse_blocks > 0;
    F32PlanArgs args{};
    F32LeaveClean leave{};
    if (by_args) {
        for (int k = 0; k < nprobs; ++k) args.probs[k] = pl.probs[k];
        args.pair = pl.pairs[0];
        args.ref = pl.refs[0];
        args.nprobs = nprobs;
        args.pbytes = (int32_t)pbytes;
        args.pair_bytes = (int32_t)pair_bytes;
        leave.thr = coarse_shared_thr(c->d_coarse.p, t1, &leave.thr_words);
        leave.flag = c->d_probs.as<int>();
        const bool clean = c->f32_clean.flag == c->d_probs.p && c->f32_clean.probs_cap == c->d_probs.cap &&
                           (leave.thr_words == 0 || (c->f32_clean.thr == leave.thr && c->f32_clean.thr_words >= leave.thr_words &&
                                                     c->f32_clean.coarse_cap == c->d_coarse.cap));
        if (!clean) {
            HIPCHK(c, hipMemsetAsync(c->d_probs.p, 0, 256, c->stream));
            if (leave.thr_words) HIPCHK(c, hipMemsetAsync(leave.thr, 0, (size_t)leave.thr_words * sizeof(unsigned), c->stream));
        }
    } else {
        HIPCHK(c, hipMemcpyAsync(c->d_probs.p, h, bytes, hipMemcpyHostToDevice, c->stream));
    }
    c->f32_clean.flag = nullptr;
    c->flag_dev = c->d_probs.as<int>();
    const CoarseProblem* d_probs = reinterpret_cast<const CoarseProblem*>(c->d_probs.as<char>() + 256);
    const SiftPair* d_pairs = reinterpret_cast<const SiftPair*>(c->d_probs.as<char>() + 256 + pbytes);
    const F32PairRef* d_refs = reinterpret_cast<const F32PairRef*>(c->d_probs.as<char>() + 256 + pbytes + pair_bytes);
    if (npairs == 0) {
        HIPCHK(c, hipMemsetAsync(c->R().d_offsets.p, 0, sizeof(int64_t), c->stream));
        return VSM_OK;
    }
    int32_t* j1 = c->d_knnj.as<int32_t>();
    int32_t* j2 = c->d_rrj.as<int32_t>();
    float* k1 = c->d_knn1.as<float>();
    float* k2 = c->d_knn2.as<float>();
    auto search = [&](const CoarseProblem* dp, const CoarseProblem* hp, int n, const CoarseTotals& t, bool lists) {
        if (n <= 0) return;
        const bool finished = launch_bf16_batch(dp, n, pl.dim, t, j1, j2, k1, k2, c->d_coarse.p, c->flag_dev, c->stream, dot, lists,
                                                by_args, by_args ? &args : nullptr, c->d_probs.as<char>());
        if (!dot && !finished) {
            int max_rows = 0;
            for (int i = 0; i < n; ++i) max_rows = std::max(max_rows, hp[i].pr.nA);
            launch_f32_l2_finish(dp, n, max_rows, nullptr, j1, j2, k1, k2, c->stream);
        }
    };
    SiftOutputs out{};
    out.slot_i0 = c->d_slot0.as<int32_t>();
    out.slot_i1 = c->d_slot1.as<int32_t>();
    out.slot_d = o.want_dist ? c->d_slotd.as<int32_t>() : nullptr;
    out.counts = c->d_counts.as<int32_t>();
    out.dense0 = o.want_dense ? c->d_dense0.as<int32_t>() : nullptr;
    out.dense1 = o.want_dense ? c->d_dense1.as<int32_t>() : nullptr;
    if (two_phase) {
        search(d_probs, pl.probs.data(), nfwd, t1, false);
        F32Rule first = rule;
        first.mutual = 0;
        SiftOutputs cand = out;
        cand.dense0 = nullptr;
        cand.dense1 = nullptr;
        launch_f32_accept(d_pairs, d_refs, npairs, first, j1, k1, k2, cand, c->stream);
        search(d_probs + nfwd, pl.probs.data() + nfwd, nrev, t2, db_lists);
        launch_f32_confirm(d_pairs, d_refs, npairs, j1, out, c->stream);
    } else {
        search(d_probs, pl.probs.data(), nprobs, t1, false);
        WordSegments seg{};
        if (o.copy_words && o.copy_dst && npairs == 1) o.copy_words(&seg);
        // the flag may only be zeroed by the kernel that also hands it to the caller
        const bool leaves = by_args && seg.n > 0;
        F32PairArgs one{};
        if (by_args) {
            one.pair = args.pair;
            one.ref = args.ref;
            one.given = 1;
            one.done_at = o.done_at;
            one.done_seq = seg.n > 0 && !c->timing ? o.done_seq : 0;
        }
        launch_f32_accept(d_pairs, d_refs, npairs, rule, j1, k1, k2, out, c->stream, seg.n ? &seg : nullptr, o.copy_dst,
                          leaves ? &leave : nullptr, by_args ? &one : nullptr);
        // (with kernel timing switched on the caller asks for the bracket's events next, and waiting on an event the
        // device has not quite reached is the slowest way to find out that the kernel is done)
        c->f32_watch_word = by_args && seg.n > 0 && o.done_seq != 0 && !c->timing;
        c->f32_copied_out = seg.n > 0;
        if (leaves) {
            c->f32_clean.flag = c->d_probs.p;
            c->f32_clean.probs_cap = c->d_probs.cap;
            c->f32_clean.thr = leave.thr;
            c->f32_clean.thr_words = leave.thr_words;
            c->f32_clean.coarse_cap = c->d_coarse.cap;
        }
    }
    if (c->timing) {
        HIPCHK(c, hipEventRecord(c->R().ev1, c->stream));
        c->R().ev_valid = true;
    }
    if (o.want_pack)
        launch_pack(d_pairs, npairs, out.counts, c->R().d_offsets.as<int64_t>(), out.slot_i0, out.slot_i1,
                    c->R().d_tight0.as<int32_t>(), c->R().d_tight1.as<int32_t>(), c->stream);
    HIPCHK(c, hipGetLastError());
    return VSM_OK;
}

// Fast path for a submission of float pairs.  *done = false when the context is in f32-only mode or a
// survivor list overflowed: the caller then takes the per-pair path on the f32 kernel.
int f32_pairs_fast(vsm_ctx* c, const int32_t* pairs, int npairs, const vsm_params* p, int metric,
                   const RunOpts& o, bool* done)
{
    *done = false;
    if (c->f32_mode != 0) return VSM_OK;
    F32Plan pl;
    int rc = plan_f32(c, pairs, npairs, metric, p->mutual != 0, pl);
    if (rc) return rc;
    if ((rc = run_f32_batch(c, pl, make_f32_rule(p, metric), o)) != VSM_OK) return rc;
    bool overflow = false;
    // a small submission brings offsets, lists and the overflow flag to pinned memory in one kernel: one wait, and
    // nothing left to copy when the lists are asked for
    if (o.want_pack && npairs > 0 && (rc = queue_small_result(c, npairs, pl.out0, kSmallResultRowsF32, c->flag_dev)) != VSM_OK) return rc;
    if (c->R().res_on_host) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        overflow = c->R().h_res.as<uint32_t>()[2 * ((size_t)npairs + 2) + 2 * c->R().res_rows0] != 0;
        if (overflow) c->R().res_on_host = false;
    } else if ((rc = coarse_overflowed(c, &overflow)) != VSM_OK) {
        return rc;
    }
    if (overflow) {
        c->R().ev_valid = false;
        return VSM_OK;
    }
    c->last_f32_path = 0;
    *done = true;
    return VSM_OK;
}

// One float pair, lists or dense forms straight into the caller's buffers with a single wait: the
// coarse / re-rank / accept kernels, the overflow flag and the result copies are queued together, and
// the flag is looked at after the one synchronisation.  Returns the match count, a VSM_E* code, or
// kNotDone when the context is in f32-only mode or a survivor list overflowed.
constexpr int kNotDone = -1000;
int f32_single_pair_fast(vsm_ctx* c, const DescSet* A, const DescSet* B, int set0, int set1, const vsm_params* p,
                         int metric, int32_t* idx0, int32_t* idx1, float* dist, int cap, int32_t* dense0,
                         int32_t* dense1)
{
    if (c->f32_mode != 0) return kNotDone;
    const int32_t one[2] = {set0, set1};
    F32Plan pl;
    int rc = plan_f32(c, one, 1, metric, p->mutual != 0, pl);
    if (rc) return rc;
    // Everything the caller gets is copied, end to end, into one block of pinned host memory (it is mapped into the
    // device's address space) by the accept kernel itself; one wait, no copy command and no kernel behind it.  The
    // device records of the pair travel as kernel arguments.
    const int take = std::min(std::max(cap, 0), A->n);   // a pair has at most nA matches
    WordSegments seg{};
    void* dst_host[8];
    int flag_host = 0, cnt_host = 0;
    auto list_words = [&](WordSegments* sg) {   // run once the device buffers exist
        auto add = [&](const void* src, int words, void* host, bool list = false) {
            if (words <= 0) return;
            sg->src[sg->n] = static_cast<const uint32_t*>(src);
            sg->count[sg->n] = words;
            if (list) sg->lists |= 1u << sg->n;
            dst_host[sg->n] = host;
            ++sg->n;
        };
        sg->n = 0;
        sg->lists = 0;
        add(c->flag_dev, 1, &flag_host);
        add(c->d_counts.p, 1, &cnt_host);
        if (idx0) add(c->d_slot0.p, take, idx0, true);
        if (idx1) add(c->d_slot1.p, take, idx1, true);
        if (dist) add(c->d_slotd.p, take, dist, true);
        if (dense0) add(c->d_dense0.p, A->n, dense0);
        if (dense1) add(c->d_dense1.p, B->n, dense1);
        seg = *sg;
    };
    const size_t words = 2 + (size_t)take * ((idx0 ? 1 : 0) + (idx1 ? 1 : 0) + (dist ? 1 : 0)) + (dense0 ? (size_t)A->n : 0) +
                         (dense1 ? (size_t)B->n : 0);
    HIPCHK(c, c->h_gather.ensure((words + 1) * 4));
    if (++c->f32_seq == 0) c->f32_seq = 1;
    volatile uint32_t* done_word = c->h_gather.as<uint32_t>() + words;
    *done_word = 0;
    RunOpts o;
    o.done_at = (uint32_t)words;
    o.done_seq = c->f32_seq;
    o.want_pack = false;
    o.want_dist = dist != nullptr;
    o.want_dense = dense0 != nullptr;
    o.by_args = true;
    o.copy_words = list_words;
    o.copy_dst = c->h_gather.as<uint32_t>();
    if ((rc = run_f32_batch(c, pl, make_f32_rule(p, metric), o)) != VSM_OK) return rc;
    if (!c->f32_copied_out) {   // (a pair without searches, or the two-stage mutual check: a kernel of its own copies)
        WordSegments sg{};
        list_words(&sg);
        launch_gather_words(sg, c->h_gather.as<uint32_t>(), c->stream);
    }
    // The accept kernel's last store is the sequence number behind the copied words: watching that word sees the
    // result a few microseconds before the stream reports the kernel as finished.  Bounded: after 2 ms (a fault, a
    // device lost) the stream is waited for, which reports what happened.
    bool seen = false;
    if (c->f32_watch_word) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(const_cast<const uint32_t*>(done_word), __ATOMIC_ACQUIRE) == c->f32_seq) {
                seen = true;
                break;
            }
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    if (!seen) HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    const uint32_t* got = c->h_gather.as<uint32_t>();
    for (int k = 0; k < seg.n; ++k) {   // flag and count come first
        const int words = (seg.lists >> k) & 1u ? std::min(seg.count[k], std::max(cnt_host, 0)) : seg.count[k];
        memcpy(dst_host[k], got, (size_t)words * 4);
        got += seg.count[k];
    }
    if (flag_host != 0) {
        c->R().ev_valid = false;
        return kNotDone;
    }
    c->last_f32_path = 0;
    return cnt_host;
}

// Per-pair path on the f32 kernel; leaves offsets and tight lists on the device like a submission.
long long f32_pairs_slow(vsm_ctx* c, const int32_t* pairs, int npairs, const vsm_params* p, int metric)
{
    std::vector<int64_t> off(npairs + 1, 0);
    std::vector<int32_t> i0, i1, m0;
    const int saved = c->f32_mode;
    c->f32_mode = 1;
    for (int k = 0; k < npairs; ++k) {
        const DescSet* A = get_set(c, pairs[2 * k]);
        const DescSet* B = get_set(c, pairs[2 * k + 1]);
        if (!A || !B || A->kind != SET_F32 || B->kind != SET_F32 || A->dim != B->dim) {
            c->f32_mode = saved;
            return fail(c, VSM_EINVAL, "pair %d is not a pair of float sets of one dimension", k);
        }
        m0.assign(std::max(A->n, 1), -1);
        const int nm = match_pair_f32(c, A, B, p, metric, m0.data(), nullptr, nullptr);
        if (nm < 0) {
            c->f32_mode = saved;
            return nm;
        }
        for (int i = 0; i < A->n; ++i)
            if (m0[i] >= 0) {
                i0.push_back(i);
                i1.push_back(m0[i]);
            }
        off[k + 1] = (int64_t)i0.size();
    }
    c->f32_mode = saved;
    const size_t tot = i0.size();
    c->R().u8_stats = false;
    HIPCHK(c, c->R().d_offsets.ensure(sizeof(int64_t) * (npairs + 1)));
    HIPCHK(c, c->R().d_tight0.ensure(std::max<size_t>(tot, 1) * 4));
    HIPCHK(c, c->R().d_tight1.ensure(std::max<size_t>(tot, 1) * 4));
    HIPCHK(c, hipMemcpy(c->R().d_offsets.p, off.data(), sizeof(int64_t) * (npairs + 1), hipMemcpyHostToDevice));
    if (tot) {
        HIPCHK(c, hipMemcpy(c->R().d_tight0.p, i0.data(), tot * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->R().d_tight1.p, i1.data(), tot * 4, hipMemcpyHostToDevice));
    }
    return (long long)tot;
}

}  // namespace

// ---------------------------------------------------------------------------------
// matching
// ---------------------------------------------------------------------------------
extern "C" long long vsm_match_pairs_device(vsm_ctx* c, const int32_t* pairs, int npairs,
                                            const vsm_params* p, int metric)
{
    if (!c) return VSM_EINVAL;
    if (npairs < 0 || (npairs > 0 && !pairs)) return fail(c, VSM_EINVAL, "bad pair list");
    int rc = check_params(c, p);
    if (rc) return rc;
    if (c->inflight) return fail(c, VSM_EINVAL, "collect the submissions in flight first");   // it would reuse their result slot
    HIPCHK(c, hipSetDevice(c->device));
    const DescSet* first = npairs > 0 ? get_set(c, pairs[0]) : nullptr;
    const bool float_sets = first && first->kind == SET_F32;
    RunOpts o;
    c->R().res_on_host = false;
    if (float_sets) {
        bool done = false;
        if ((rc = f32_pairs_fast(c, pairs, npairs, p, metric, o, &done)) != VSM_OK) return rc;
        if (!done) {
            const long long got = f32_pairs_slow(c, pairs, npairs, p, metric);
            if (got < 0) return got;
            c->last_npairs = npairs;
            c->last_total = got;
            c->last_slot = c->rs;
            return got;
        }
    } else {
        Plan pl;
        if ((rc = plan_u8(c, pairs, npairs, pl)) != VSM_OK) return rc;
        if ((rc = run_u8(c, pl, make_rule(p), o)) != VSM_OK) return rc;
        if ((rc = queue_small_result(c, npairs, pl.out0, kSmallResultRows, nullptr)) != VSM_OK) return rc;
    }
    if (c->R().res_on_host) {   // total (and the rescan counters) come with the lists
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        const int64_t* off = c->R().h_res.as<int64_t>();
        if (!float_sets && c->R().u8_stats) note_u8_stats(c, reinterpret_cast<const int32_t*>(off + npairs + 1));
        c->last_npairs = npairs;
        c->last_total = off[npairs];
        c->last_slot = c->rs;
        return off[npairs];
    }
    HIPCHK(c, c->h_small.ensure(64));
    long long* tot = c->h_small.as<long long>();
    tot[0] = tot[1] = 0;
    const bool stats = !float_sets && c->R().u8_stats;
    if (npairs)
        HIPCHK(c, hipMemcpyAsync(tot, c->R().d_offsets.as<int64_t>() + npairs, sizeof(int64_t) * (stats ? 2 : 1),
                                 hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (stats) note_u8_stats(c, reinterpret_cast<const int32_t*>(tot + 1));
    c->last_npairs = npairs;
    c->last_total = *tot;
    c->last_slot = c->rs;
    return *tot;
}

extern "C" long long vsm_fetch_matches(vsm_ctx* c, int64_t* offsets, int32_t* idx0, int32_t* idx1, long long cap)
{
    if (!c) return VSM_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    const long long tot = c->last_total;
    if (c->R().res_on_host && c->R().res_pairs == c->last_npairs && c->last_slot == c->rs) {
        // the submission brought its lists to pinned memory itself, and whoever set last_total has waited for it
        if (tot > cap) return fail(c, VSM_ECAP, "%lld matches do not fit a buffer of %lld", tot, cap);
        const size_t off_words = 2 * ((size_t)c->last_npairs + 2);
        const uint32_t* words = c->R().h_res.as<uint32_t>();
        if (offsets) memcpy(offsets, words, sizeof(int64_t) * (c->last_npairs + 1));
        if (tot > 0 && idx0) memcpy(idx0, words + off_words, sizeof(int32_t) * tot);
        if (tot > 0 && idx1) memcpy(idx1, words + off_words + c->R().res_rows0, sizeof(int32_t) * tot);
        return tot;
    }
    if (offsets) {
        if (c->last_npairs)
            HIPCHK(c, hipMemcpyAsync(offsets, c->R().d_offsets.p, sizeof(int64_t) * (c->last_npairs + 1), hipMemcpyDeviceToHost, c->stream));
        else
            offsets[0] = 0;
    }
    if (tot > cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return fail(c, VSM_ECAP, "%lld matches do not fit a buffer of %lld", tot, cap);
    }
    if (tot > 0) {
        if (idx0) HIPCHK(c, hipMemcpyAsync(idx0, c->R().d_tight0.p, sizeof(int32_t) * tot, hipMemcpyDeviceToHost, c->stream));
        if (idx1) HIPCHK(c, hipMemcpyAsync(idx1, c->R().d_tight1.p, sizeof(int32_t) * tot, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return tot;
}

extern "C" long long vsm_match_pairs(vsm_ctx* c, const int32_t* pairs, int npairs, const vsm_params* p,
                                     int metric, int64_t* offsets, int32_t* idx0, int32_t* idx1, long long cap)
{
    if (!c) return VSM_EINVAL;
    if (npairs < 0 || (npairs > 0 && !pairs) || !offsets) return fail(c, VSM_EINVAL, "bad pair list or offsets buffer");
    int rc = check_params(c, p);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    // cut the list into submissions that fit the scratch budget, of about equal scratch need
    std::vector<size_t> need(npairs);
    size_t need_total = 0;
    for (int k = 0; k < npairs; ++k) {
        const DescSet* A = get_set(c, pairs[2 * k]);
        const DescSet* B = get_set(c, pairs[2 * k + 1]);
        if (!A || !B) return fail(c, VSM_EINVAL, "pair %d names an unknown set", k);
        const size_t chunks = ((size_t)B->n_pad + kChunkCols - 1) / kChunkCols;
        need[k] = A->kind == SET_F32 ? ((size_t)A->n_pad + B->n_pad) * 2 * (kCoarseCap * 8 + 16 + 64)
                                     : (chunks * A->n_pad + A->n) * kScratchPerRow;
        need_total += need[k];
    }
    const size_t per_sub = std::max<size_t>(c->scratch_budget / 2, 1);
    const size_t want_subs = std::max<size_t>((need_total + per_sub - 1) / per_sub, ((size_t)npairs + 8191) / 8192);
    const size_t target = want_subs ? (need_total + want_subs - 1) / want_subs : 0;
    std::vector<int> cuts{0};
    while (cuts.back() < npairs) {
        const int done = cuts.back();
        size_t scratch = 0;
        int take = 0;
        while (done + take < npairs) {
            if (take > 0 && (scratch + need[done + take] > per_sub || scratch >= target || take >= 8192)) break;
            scratch += need[done + take];
            ++take;
        }
        cuts.push_back(done + take);
    }
    const int nsub = (int)cuts.size() - 1;
    offsets[0] = 0;
    long long total = 0;
    // submission k + 1 is queued before the lists of submission k are fetched, and the fetch runs on
    // the copy stream, so the compute stream never waits for the host
    if (c->inflight) return fail(c, VSM_EINVAL, "collect the submissions in flight first");
    auto submit = [&](int k) { return vsm_submit_pairs(c, pairs + 2 * (size_t)cuts[k], cuts[k + 1] - cuts[k], p, metric); };
    auto drain = [&]() {   // after a failure: nothing of this call may still be queued
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->copy_stream);
        c->inflight = 0;
    };
    // Float submissions are searched inside submit (the survivor-overflow check needs the result) and
    // cannot queue behind one another: they run one after the other, submit(k) then collect(k).
    const DescSet* first = npairs > 0 ? get_set(c, pairs[0]) : nullptr;
    const bool pipelined = !(first && first->kind == SET_F32);
    if (pipelined && nsub > 0 && (rc = submit(0)) != VSM_OK) return rc;
    for (int k = 0; k < nsub; ++k) {
        if ((pipelined ? k + 1 < nsub && (rc = submit(k + 1)) != VSM_OK : (rc = submit(k)) != VSM_OK)) {
            drain();
            return rc;
        }
        const int done = cuts[k], take = cuts[k + 1] - cuts[k];
        const long long got = vsm_collect_matches(c, offsets + done, idx0 ? idx0 + total : nullptr,
                                                  idx1 ? idx1 + total : nullptr, cap - total);
        if (got < 0) {
            drain();
            return got;
        }
        for (int i = 0; i <= take; ++i) offsets[done + i] += total;
        total += got;
    }
    return total;
}

extern "C" int vsm_submit_pairs(vsm_ctx* c, const int32_t* pairs, int npairs, const vsm_params* p, int metric)
{
    if (!c) return VSM_EINVAL;
    if (npairs < 0 || (npairs > 0 && !pairs)) return fail(c, VSM_EINVAL, "bad pair list");
    int rc = check_params(c, p);
    if (rc) return rc;
    if (c->inflight >= 2) return fail(c, VSM_EINVAL, "two submissions are in flight already: collect one first");
    HIPCHK(c, hipSetDevice(c->device));
    const int slot = (c->inflight_head + c->inflight) & 1;
    const int timing_slot_before = c->timing_slot;
    c->rs = slot;
    c->R().res_on_host = false;
    if (slot == c->last_slot) c->last_npairs = -1;   // the lists of the last collected submission are about to be overwritten
    const DescSet* first = npairs > 0 ? get_set(c, pairs[0]) : nullptr;
    bool on_host = false;
    if (first && first->kind == SET_F32) {
        // float pairs are searched before this call returns (the survivor-overflow check needs the
        // result); only the copy of the lists is left to vsm_collect_matches
        if (c->inflight) return fail(c, VSM_EINVAL, "float submissions are not queued behind others: collect first");
        RunOpts o;
        bool done = false;
        if ((rc = f32_pairs_fast(c, pairs, npairs, p, metric, o, &done)) != VSM_OK) return rc;
        if (!done) {
            const long long got = f32_pairs_slow(c, pairs, npairs, p, metric);
            if (got < 0) return (int)got;
            on_host = true;
        }
    } else {
        Plan pl;
        if ((rc = plan_u8(c, pairs, npairs, pl)) != VSM_OK) return rc;
        RunOpts o;
        if ((rc = run_u8(c, pl, make_rule(p), o)) != VSM_OK) return rc;
        if ((rc = queue_small_result(c, npairs, pl.out0, kSmallResultRows, nullptr)) != VSM_OK) return rc;
    }
    HIPCHK(c, hipEventRecord(c->R().packed, c->stream));
    c->timing_slot = timing_slot_before;   // the brackets become readable once this submission is collected
    c->inflight_pairs[slot] = npairs;
    c->inflight_on_host[slot] = on_host;
    ++c->inflight;
    return VSM_OK;
}

extern "C" long long vsm_collect_matches(vsm_ctx* c, int64_t* offsets, int32_t* idx0, int32_t* idx1, long long cap)
{
    if (!c) return VSM_EINVAL;
    if (!offsets) return fail(c, VSM_EINVAL, "offsets buffer is NULL");
    if (c->inflight <= 0) return fail(c, VSM_EINVAL, "no submission in flight");
    HIPCHK(c, hipSetDevice(c->device));
    const int slot = c->inflight_head;
    const int npairs = c->inflight_pairs[slot];
    c->inflight_head ^= 1;
    --c->inflight;
    c->rs = slot;
    c->timing_slot = slot;
    vsm_ctx::ResultSlot& R = c->R();
    HIPCHK(c, hipEventSynchronize(R.packed));
    const bool stats = R.u8_stats && !c->inflight_on_host[slot];
    if (R.res_on_host) {   // everything is in pinned memory already
        const int64_t* off = R.h_res.as<int64_t>();
        if (stats) note_u8_stats(c, reinterpret_cast<const int32_t*>(off + npairs + 1));
        const long long got = off[npairs];
        c->last_npairs = npairs;
        c->last_total = got;
        c->last_slot = slot;
        if (got > cap) return fail(c, VSM_ECAP, "%lld matches do not fit a buffer of %lld", got, cap);
        const size_t off_words = 2 * ((size_t)npairs + 2), rows0 = R.res_rows0;
        const uint32_t* words = R.h_res.as<uint32_t>();
        if (got > 0 && idx0) memcpy(idx0, words + off_words, sizeof(int32_t) * got);
        if (got > 0 && idx1) memcpy(idx1, words + off_words + rows0, sizeof(int32_t) * got);
        memcpy(offsets, off, sizeof(int64_t) * (npairs + 1));
        return got;
    }
    HIPCHK(c, R.h_off.ensure(sizeof(int64_t) * (npairs + 2)));
    HIPCHK(c, hipMemcpyAsync(R.h_off.p, R.d_offsets.p, sizeof(int64_t) * (npairs + (stats ? 2 : 1)), hipMemcpyDeviceToHost, c->copy_stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    const int64_t* off = R.h_off.as<int64_t>();
    if (stats) note_u8_stats(c, reinterpret_cast<const int32_t*>(off + npairs + 1));
    const long long got = off[npairs];
    c->last_npairs = npairs;
    c->last_total = got;
    c->last_slot = slot;
    if (got > cap) return fail(c, VSM_ECAP, "%lld matches do not fit a buffer of %lld", got, cap);
    if (got > 0) {
        if (idx0) HIPCHK(c, hipMemcpyAsync(idx0, R.d_tight0.p, sizeof(int32_t) * got, hipMemcpyDeviceToHost, c->copy_stream));
        if (idx1) HIPCHK(c, hipMemcpyAsync(idx1, R.d_tight1.p, sizeof(int32_t) * got, hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    }
    memcpy(offsets, off, sizeof(int64_t) * (npairs + 1));
    return got;
}

extern "C" int vsm_match_pair(vsm_ctx* c, int set0, int set1, const vsm_params* p, int metric,
                              int32_t* idx0, int32_t* idx1, void* dist, int cap)
{
    if (!c) return VSM_EINVAL;
    int rc = check_params(c, p);
    if (rc) return rc;
    SetKind kind;
    if ((rc = set_kind_of_pair(c, set0, set1, &kind)) != VSM_OK) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if (kind == SET_F32) {
        const DescSet* A = get_set(c, set0);
        const DescSet* B = get_set(c, set1);
        const int fast = f32_single_pair_fast(c, A, B, set0, set1, p, metric, idx0, idx1, static_cast<float*>(dist), cap,
                                              nullptr, nullptr);
        if (fast != kNotDone) {
            if (fast > cap) return fail(c, VSM_ECAP, "%d matches do not fit a buffer of %d", fast, cap);
            return fast;
        }
        std::vector<int32_t> m0(std::max(A->n, 1));
        std::vector<float> b0(std::max(A->n, 1));
        const int nm = match_pair_f32(c, A, B, p, metric, m0.data(), nullptr, b0.data());
        if (nm < 0) return nm;
        if (nm > cap) return fail(c, VSM_ECAP, "%d matches do not fit a buffer of %d", nm, cap);
        int k = 0;
        for (int i = 0; i < A->n; ++i) {
            if (m0[i] < 0) continue;
            if (idx0) idx0[k] = i;
            if (idx1) idx1[k] = m0[i];
            if (dist) static_cast<float*>(dist)[k] = b0[i];
            ++k;
        }
        return nm;
    }
    const int32_t pr[2] = {set0, set1};
    Plan pl;
    if ((rc = plan_u8(c, pr, 1, pl)) != VSM_OK) return rc;
    RunOpts o;
    o.want_dist = dist != nullptr;
    o.want_pack = false;
    if ((rc = run_u8(c, pl, make_rule(p), o)) != VSM_OK) return rc;
    // count and lists in one block, one copy, one wait (a pair has at most nA matches)
    const DescSet* A = get_set(c, set0);
    const int take = std::min(std::max(cap, 0), A->n);
    WordSegments seg{};
    void* dst_host[8];
    auto add = [&](const void* src, int words, void* host) {
        if (words <= 0) return;
        seg.src[seg.n] = static_cast<const uint32_t*>(src);
        seg.count[seg.n] = words;
        dst_host[seg.n] = host;
        ++seg.n;
    };
    int nm = 0;
    int32_t stats[2] = {0, 0};
    add(c->d_counts.p, 1, &nm);
    if (idx0) add(c->d_slot0.p, take, idx0);
    if (idx1) add(c->d_slot1.p, take, idx1);
    if (dist) add(c->d_slotd.p, take, dist);
    const int stats_seg = seg.n;
    add(c->R().d_offsets.as<int64_t>() + 2, 2, stats);   // the rescan pass's counters (one pair: behind two offsets)
    size_t words = 0;
    for (int k = 0; k < seg.n; ++k) words += (size_t)seg.count[k];
    // the gather kernel writes straight into pinned host memory (it is mapped into the device's address space):
    // no staging block and no copy command between the last kernel and the wait
    HIPCHK(c, c->h_gather.ensure(words * 4));
    launch_gather_words(seg, c->h_gather.as<uint32_t>(), c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    const uint32_t* got = c->h_gather.as<uint32_t>();
    for (int k = 0; k < seg.n; ++k) {
        // the first segment is the count; list segments hold `take` slots of which nm are filled
        const size_t fill = k == 0 ? 1 : k == stats_seg ? 2 : (size_t)std::min(nm, take);
        memcpy(dst_host[k], got, fill * 4);
        got += seg.count[k];
    }
    note_u8_stats(c, stats);
    if (nm > cap) return fail(c, VSM_ECAP, "%d matches do not fit a buffer of %d", nm, cap);
    return nm;
}

// One u8 pair through the same kernels as vsm_match_pair (the tile-minimum form with its passes one after the other),
// with every intermediate in the per-device static scratch array and the results written straight into the caller's
// device buffers.  Calls no hipMalloc / hipFree / allocating routine: the pair record travels as a kernel argument
// and the work items are written by a kernel.  Calls on one GPU are ordered behind one another by an event per
// device (created in vsm_create), whatever context or stream they come from: they share the array.
extern "C" int vsm_match_pair_noalloc(vsm_ctx* c, int set0, int set1, const vsm_params* p,
                                      int32_t* idx0_dev, int32_t* idx1_dev, int32_t* count_dev, int cap)
{
    if (!c) return VSM_EINVAL;
    int rc = check_params(c, p);
    if (rc) return rc;
    const DescSet* A = get_set(c, set0);
    const DescSet* B = get_set(c, set1);
    if (!A || !B) return fail(c, VSM_EINVAL, "unknown set id (%d, %d)", set0, set1);
    if (A->kind != SET_U8 || B->kind != SET_U8)
        return fail(c, VSM_EUNSUPPORTED, "vsm_match_pair_noalloc takes u8 (SIFT) sets only");
    if (!idx0_dev || !idx1_dev || !count_dev) return fail(c, VSM_EINVAL, "idx0_dev, idx1_dev and count_dev must be device buffers");
    if (A->n > kNoAllocMaxRows || B->n > kNoAllocMaxRows)
        return fail(c, VSM_ECAP, "vsm_match_pair_noalloc takes sets of at most %d rows (got %d and %d): its scratch is a fixed "
                                 "%zu-byte array", kNoAllocMaxRows, A->n, B->n, kNoAllocScratchBytes);
    if (cap < A->n) return fail(c, VSM_ECAP, "idx0_dev / idx1_dev must hold %d entries (one per image-0 row), cap is %d", A->n, cap);
    if (!c->noalloc_scratch || !c->noalloc_done) return fail(c, VSM_EUNSUPPORTED, "no scratch array on device %d", c->device);
    HIPCHK(c, hipSetDevice(c->device));
    if (A->n == 0 || B->n == 0) {
        HIPCHK(c, hipMemsetAsync(count_dev, 0, sizeof(int32_t), c->stream));
        return VSM_OK;
    }
    SiftPair sp{};
    sp.fragA = static_cast<const int8_t*>(A->frag);
    sp.nrmA = static_cast<const int32_t*>(A->nrm);
    sp.fragB = static_cast<const int8_t*>(B->frag);
    sp.nrmB = static_cast<const int32_t*>(B->nrm);
    sp.nA = A->n; sp.nB = B->n; sp.nA_pad = A->n_pad; sp.nB_pad = B->n_pad;
    sp.nstrips = A->n_pad / kStripRows;
    sp.chunk_cols = chunk_cols_for(sp.nstrips, B->n_pad);
    sp.nchunks = (B->n_pad + sp.chunk_cols - 1) / sp.chunk_cols;
    const int nitems = sp.nstrips * sp.nchunks;
    const int nvitems = p->mutual ? (A->n + kVerifyBlock - 1) / kVerifyBlock : 0;
    const size_t rows = (size_t)sp.nchunks * sp.nA_pad, out0 = (size_t)A->n;
    const MatchRule rule = make_rule(p);
    const bool row_lists = rule.num2 > 0 || rule.max_dist >= 0;
    // carve the scratch array (every part 256-byte aligned)
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return c->noalloc_scratch + at;
    };
    int8_t* zero_tile = reinterpret_cast<int8_t*>(carve(kTileBytes));
    SiftPair* d_pair = reinterpret_cast<SiftPair*>(carve(sizeof(SiftPair)));
    // query items, the verify pass's items and their two counters, then the rescan pass's items
    SiftItem* d_items = reinterpret_cast<SiftItem*>(carve(sizeof(SiftItem) * (size_t)(2 * nitems + nvitems + 8)));
    int32_t* rr1 = reinterpret_cast<int32_t*>(carve(rows * 4));
    int32_t* rrj = reinterpret_cast<int32_t*>(carve(rows * 4));
    int32_t* rr2 = reinterpret_cast<int32_t*>(carve(rows * 4));
    SiftCandidates cand{};
    cand.rs_d1 = rrj;   // the group and runner-up records are spent when the rescan pass runs
    cand.rs_j1 = rr2;
    cand.rs_d2 = reinterpret_cast<int32_t*>(carve(rows * 4));
    cand.vbest = reinterpret_cast<unsigned long long*>(carve(out0 * 8));
    cand.i0 = reinterpret_cast<int32_t*>(carve(out0 * 4));
    cand.j1 = reinterpret_cast<int32_t*>(carve(out0 * 4));
    cand.d1 = reinterpret_cast<int32_t*>(carve(out0 * 4));
    cand.aux = reinterpret_cast<int32_t*>(carve(out0 * 4));
    cand.rescan_list = reinterpret_cast<int32_t*>(carve(out0 * 4));
    int32_t* row_list = reinterpret_cast<int32_t*>(carve(out0 * 4));
    int32_t* row_codes = reinterpret_cast<int32_t*>(carve(out0 * 4));
    // count, largest accepted distance, row-list length, rescan count, the rescan pass's two counters, slice counts
    int32_t* small = reinterpret_cast<int32_t*>(carve(sizeof(int32_t) * (8 + 2 * kMaxRowSlices)));
    cand.count = small;
    cand.rescan_cnt = small + 3;
    cand.rescan_nitems = small + 4;
    int32_t* slice_counts = small + 8;
    cand.rescan_items = d_items + nitems + nvitems + 8;
    if (row_lists) {
        cand.dmax = small + 1;
        cand.nvrows = small + 2;
        cand.vrows = row_list;
        cand.vcodes = row_codes;
        cand.zero_tile = zero_tile;
    }
    if (off > kNoAllocScratchBytes)
        return fail(c, VSM_ECAP, "this pair needs %zu bytes of scratch, the fixed array holds %zu", off, kNoAllocScratchBytes);
    // Everything a kernel of this call reads from the scratch array is written earlier in the same call (pair record
    // and items by the plan kernel, partial results by the query pass for every padded row, candidate records, counts
    // and the largest accepted distance by the select and resolve passes, the rescan and verify item lists by their
    // plan kernels), except the tile of zeros, which is cleared here: what an earlier call left behind is never read.
    std::lock_guard<std::mutex> lock(g_scratch_mu);   // the wait and the record below are one step for other threads
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->noalloc_done, 0));
    HIPCHK(c, hipMemsetAsync(zero_tile, 0, kTileBytes, c->stream));
    launch_sift_plan_single(sp, d_pair, d_items, c->stream);
    launch_sift_query_tile(d_pair, d_items, nitems, rr1, rrj, rr2, c->stream);
    launch_sift_select_tile(d_pair, 1, A->n, sp.nchunks, rr1, rrj, rr2, rule, cand, slice_counts, c->stream);
    launch_sift_resolve(d_pair, 1, A->n, nitems, rule, cand, c->stream);
    SiftOutputs out{};
    out.slot_i0 = idx0_dev;
    out.slot_i1 = idx1_dev;
    out.counts = count_dev;
    if (nvitems) {
        if (row_lists) launch_sift_verify_rows(d_pair, 1, A->n, rr1, cand, slice_counts, c->stream);
        launch_sift_verify(d_pair, 1, d_items + nitems, nvitems, reinterpret_cast<int*>(d_items + nitems + nvitems), cand, c->stream);
    }
    launch_sift_confirm(d_pair, 1, cand, nvitems ? 1 : 0, out, c->stream);
    HIPCHK(c, hipEventRecord(c->noalloc_done, c->stream));
    HIPCHK(c, hipGetLastError());
    return VSM_OK;
}

extern "C" int vsm_match_pair_dense(vsm_ctx* c, int set0, int set1, const vsm_params* p, int metric,
                                    int32_t* matches0, int32_t* matches1)
{
    if (!c) return VSM_EINVAL;
    if (!matches0 || !matches1) return fail(c, VSM_EINVAL, "matches0/matches1 must not be NULL");
    int rc = check_params(c, p);
    if (rc) return rc;
    SetKind kind;
    if ((rc = set_kind_of_pair(c, set0, set1, &kind)) != VSM_OK) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const DescSet* A = get_set(c, set0);
    const DescSet* B = get_set(c, set1);
    if (kind == SET_F32) {
        const int fast = f32_single_pair_fast(c, A, B, set0, set1, p, metric, nullptr, nullptr, nullptr, 0, matches0, matches1);
        if (fast != kNotDone) return fast;
        return match_pair_f32(c, A, B, p, metric, matches0, matches1, nullptr);
    }
    const int32_t pr[2] = {set0, set1};
    Plan pl;
    if ((rc = plan_u8(c, pr, 1, pl)) != VSM_OK) return rc;
    RunOpts o;
    o.want_dense = true;
    o.want_pack = false;
    if ((rc = run_u8(c, pl, make_rule(p), o)) != VSM_OK) return rc;
    HIPCHK(c, c->h_small.ensure(64));
    int32_t* cnt = c->h_small.as<int32_t>();
    HIPCHK(c, hipMemcpyAsync(cnt, c->d_counts.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cnt + 2, c->R().d_offsets.as<int64_t>() + 2, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (A->n) HIPCHK(c, hipMemcpyAsync(matches0, c->d_dense0.p, sizeof(int32_t) * A->n, hipMemcpyDeviceToHost, c->stream));
    if (B->n) HIPCHK(c, hipMemcpyAsync(matches1, c->d_dense1.p, sizeof(int32_t) * B->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    note_u8_stats(c, cnt + 2);
    return *cnt;
}

extern "C" int vsm_knn2(vsm_ctx* c, int set_q, int set_db, int metric,
                        int32_t* idx1, int32_t* idx2, void* d1, void* d2)
{
    if (!c) return VSM_EINVAL;
    SetKind kind;
    int rc = set_kind_of_pair(c, set_q, set_db, &kind);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const DescSet* A = get_set(c, set_q);
    const DescSet* B = get_set(c, set_db);
    if (A->n == 0) return 0;
    if (B->n == 0) {
        for (int i = 0; i < A->n; ++i) {
            if (idx1) idx1[i] = -1;
            if (idx2) idx2[i] = -1;
        }
        return A->n;
    }
    if (kind == SET_F32) {
        F32Result r;
        if ((rc = run_f32_knn(c, A, B, metric, false, &r)) != VSM_OK) return rc;
        if ((size_t)A->n <= kSmallResultRows) {
            // the four arrays go to pinned memory end to end in one kernel: one wait instead of four copy commands into
            // the caller's pageable arrays
            WordSegments seg{};
            void* host[4];
            auto add = [&](const void* src, void* dst) {
                if (!dst) return;
                seg.src[seg.n] = static_cast<const uint32_t*>(src);
                seg.count[seg.n] = A->n;
                host[seg.n++] = dst;
            };
            add(r.j1, idx1); add(r.j2, idx2); add(r.k1, d1); add(r.k2, d2);
            if (seg.n) {
                HIPCHK(c, c->h_gather.ensure((size_t)seg.n * A->n * 4));
                launch_gather_words(seg, c->h_gather.as<uint32_t>(), c->stream);
                HIPCHK(c, hipStreamSynchronize(c->stream));
                for (int k = 0; k < seg.n; ++k) memcpy(host[k], c->h_gather.as<uint32_t>() + (size_t)k * A->n, (size_t)A->n * 4);
            }
            return A->n;
        }
        if (idx1) HIPCHK(c, hipMemcpyAsync(idx1, r.j1, sizeof(int32_t) * A->n, hipMemcpyDeviceToHost, c->stream));
        if (idx2) HIPCHK(c, hipMemcpyAsync(idx2, r.j2, sizeof(int32_t) * A->n, hipMemcpyDeviceToHost, c->stream));
        if (d1) HIPCHK(c, hipMemcpyAsync(d1, r.k1, sizeof(float) * A->n, hipMemcpyDeviceToHost, c->stream));
        if (d2) HIPCHK(c, hipMemcpyAsync(d2, r.k2, sizeof(float) * A->n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return A->n;
    }
    if (idx2) return fail(c, VSM_EUNSUPPORTED, "the runner-up index is not tracked for u8 sets (pass idx2 = NULL)");
    const int32_t pr[2] = {set_q, set_db};
    Plan pl;
    if ((rc = plan_u8(c, pr, 1, pl)) != VSM_OK) return rc;
    RunOpts o;
    o.want_knn = true;
    o.want_pack = false;
    MatchRule rule{};
    rule.num2 = 0; rule.den2 = 1; rule.max_dist = -1; rule.mutual = 0;
    if ((rc = run_u8(c, pl, rule, o)) != VSM_OK) return rc;
    if (idx1) HIPCHK(c, hipMemcpyAsync(idx1, c->d_knnj.p, sizeof(int32_t) * A->n, hipMemcpyDeviceToHost, c->stream));
    if (d1) HIPCHK(c, hipMemcpyAsync(d1, c->d_knn1.p, sizeof(int32_t) * A->n, hipMemcpyDeviceToHost, c->stream));
    if (d2) HIPCHK(c, hipMemcpyAsync(d2, c->d_knn2.p, sizeof(int32_t) * A->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return A->n;
}

extern "C" int vsm_recover_pose(vsm_ctx* c, const int32_t* xy_pairs, int npairs, const float* cameras, const float* F,
                                const int64_t* in_offsets, const int32_t* in_idx0, const int32_t* in_idx1,
                                float* R, float* T, int32_t* ne)
{
    if (!c) return VSM_EINVAL;
    if (npairs < 0 || (npairs > 0 && (!xy_pairs || !cameras || !F || !in_offsets || !R || !T || !ne)))
        return fail(c, VSM_EINVAL, "bad argument");
    if (npairs == 0) return 0;
    const int64_t total = in_offsets[npairs];
    if (in_offsets[0] != 0 || total < 0 || (total > 0 && (!in_idx0 || !in_idx1)))
        return fail(c, VSM_EINVAL, "in_offsets must start at 0 and ascend, with the lists present");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<GeomPair> gp(npairs);
    for (int p = 0; p < npairs; ++p) {
        const DescSet* A = get_set(c, xy_pairs[2 * p]);
        const DescSet* B = get_set(c, xy_pairs[2 * p + 1]);
        if (!A || !B || A->kind != SET_XY || B->kind != SET_XY)
            return fail(c, VSM_EINVAL, "pair %d does not name two keypoint sets", p);
        const int64_t n = in_offsets[p + 1] - in_offsets[p];
        if (n < 0 || n > INT32_MAX) return fail(c, VSM_EINVAL, "in_offsets must ascend (pair %d)", p);
        for (int k = 0; k < 8; ++k)
            if (k != 2 && k != 3 && k != 6 && k != 7 && !(cameras[8 * (size_t)p + k] > 0.0f))
                return fail(c, VSM_EINVAL, "pair %d: focal lengths must be positive", p);
        GeomPair& q = gp[p];
        q.xy0 = static_cast<const float*>(A->frag);
        q.xy1 = static_cast<const float*>(B->frag);
        q.n0 = A->n; q.n1 = B->n;
        q.n = (int32_t)n;
        q.pad = 0;
        q.off = in_offsets[p];
    }
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t tot = (size_t)std::max<int64_t>(total, 1);
    const size_t b_gp = up(sizeof(GeomPair) * npairs), b_cam = up((size_t)npairs * 32), b_F = up((size_t)npairs * 36);
    const size_t b_idx = up(tot * 4), b_T = up((size_t)npairs * 12), b_ne = up((size_t)npairs * 4);
    HIPCHK(c, c->d_geom.ensure(b_gp + b_cam + 2 * b_F + 2 * b_idx + b_T + b_ne));
    char* w = c->d_geom.as<char>();
    GeomPair* d_gp = reinterpret_cast<GeomPair*>(w); w += b_gp;
    float* d_cam = reinterpret_cast<float*>(w); w += b_cam;
    float* d_F = reinterpret_cast<float*>(w); w += b_F;
    float* d_R = reinterpret_cast<float*>(w); w += b_F;
    int32_t* d_i0 = reinterpret_cast<int32_t*>(w); w += b_idx;
    int32_t* d_i1 = reinterpret_cast<int32_t*>(w); w += b_idx;
    float* d_T = reinterpret_cast<float*>(w); w += b_T;
    int32_t* d_ne = reinterpret_cast<int32_t*>(w); w += b_ne;
    // d_geom may still serve a queued verification: that work lies on this same stream, before these copies
    HIPCHK(c, hipMemcpyAsync(d_gp, gp.data(), sizeof(GeomPair) * npairs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_cam, cameras, (size_t)npairs * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_F, F, (size_t)npairs * 36, hipMemcpyHostToDevice, c->stream));
    if (total > 0) {
        HIPCHK(c, hipMemcpyAsync(d_i0, in_idx0, (size_t)total * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_i1, in_idx1, (size_t)total * 4, hipMemcpyHostToDevice, c->stream));
    }
    launch_geom_pose(d_gp, npairs, d_cam, d_F, d_i0, d_i1, d_R, d_T, d_ne, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(R, d_R, (size_t)npairs * 36, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(T, d_T, (size_t)npairs * 12, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(ne, d_ne, (size_t)npairs * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // also: gp and the caller's arrays must outlive the copies
    int with_pose = 0;
    for (int p = 0; p < npairs; ++p) with_pose += ne[p] > 0;
    return with_pose;
}


// ---------------------------------------------------------------------------------
// geometric verification
// ---------------------------------------------------------------------------------
extern "C" int vsm_upload_xy(vsm_ctx* c, const float* xy, int n)
{
    if (!c) return VSM_EINVAL;
    if (n < 0 || (n > 0 && !xy)) return fail(c, VSM_EINVAL, "bad keypoint buffer (n=%d)", n);
    HIPCHK(c, hipSetDevice(c->device));
    DescSet s;
    s.kind = SET_XY;
    s.n = n;
    s.n_pad = n;
    s.dim = 2;
    if (n > 0) {
        HIPCHK(c, hipMalloc(&s.frag, (size_t)n * 2 * sizeof(float)));
        hipError_t e = hipMemcpy(s.frag, xy, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(s.frag);
            return fail(c, VSM_ENODEV, "keypoint copy: %s", hipGetErrorString(e));
        }
    }
    const int id = new_set_slot(c);
    c->sets[id] = s;
    return id;
}

extern "C" long long vsm_verify_matches(vsm_ctx* c, const int32_t* xy_pairs, int npairs, const int64_t* offsets,
                                        const int32_t* idx0, const int32_t* idx1, const vsm_geom_params* g,
                                        float* F, int32_t* nf, int64_t* in_offsets, int32_t* in_idx0,
                                        int32_t* in_idx1, long long cap)
{
    if (!c) return VSM_EINVAL;
    if (npairs < 0 || (npairs > 0 && !xy_pairs) || !g || !in_offsets) return fail(c, VSM_EINVAL, "bad argument");
    if (npairs > 65535) return fail(c, VSM_EINVAL, "%d pairs in one call (limit 65535)", npairs);
    if (!(g->max_sampson > 0.0f) || g->hypotheses < 0 || g->hypotheses > 65536)
        return fail(c, VSM_EINVAL, "bad geometry parameters (threshold %g, %d samples)", (double)g->max_sampson, g->hypotheses);
    const bool resident = !offsets && !idx0 && !idx1;   // verify the lists of the last submission where they lie
    if (!resident && (!offsets || (offsets[npairs] > 0 && (!idx0 || !idx1))))
        return fail(c, VSM_EINVAL, "pass offsets, idx0 and idx1, or none of them");
    // (one more submission may be in flight: it uses the other result slot, and this call queues behind it)
    if (resident && c->last_npairs != npairs)
        return fail(c, VSM_EINVAL, "no collected submission of %d pairs whose lists are still on the device", npairs);
    HIPCHK(c, hipSetDevice(c->device));
    in_offsets[0] = 0;
    if (npairs == 0) return 0;
    std::vector<int64_t> off_host;
    if (resident) {
        off_host.resize((size_t)npairs + 1);
        HIPCHK(c, hipMemcpyAsync(off_host.data(), c->slots[c->last_slot].d_offsets.p, sizeof(int64_t) * (npairs + 1), hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
        offsets = off_host.data();
    }
    const int64_t total = offsets[npairs];
    if (offsets[0] != 0 || total < 0) return fail(c, VSM_EINVAL, "offsets must start at 0 and ascend");
    std::vector<GeomPair> gp(npairs);
    int max_matches = 0;
    for (int p = 0; p < npairs; ++p) {
        const DescSet* A = get_set(c, xy_pairs[2 * p]);
        const DescSet* B = get_set(c, xy_pairs[2 * p + 1]);
        if (!A || !B || A->kind != SET_XY || B->kind != SET_XY)
            return fail(c, VSM_EINVAL, "pair %d does not name two keypoint sets", p);
        const int64_t n = offsets[p + 1] - offsets[p];
        if (n < 0 || n > INT32_MAX) return fail(c, VSM_EINVAL, "offsets must ascend (pair %d)", p);
        GeomPair& q = gp[p];
        q.xy0 = static_cast<const float*>(A->frag);
        q.xy1 = static_cast<const float*>(B->frag);
        q.n0 = A->n; q.n1 = B->n;
        q.n = (int32_t)n;
        q.pad = 0;
        q.off = offsets[p];
        max_matches = std::max(max_matches, (int)n);
    }
    // one arena: pair records, match lists (unless resident), coordinates, per-pair results, slots, tight lists
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t tot = (size_t)std::max<int64_t>(total, 1);
    const size_t b_gp = up(sizeof(GeomPair) * npairs), b_sp = up(sizeof(SiftPair) * npairs);
    const size_t b_idx = resident ? 0 : up(tot * 4), b_co = up(tot * 16), b_F = up((size_t)npairs * 36);
    const size_t b_cnt = up((size_t)npairs * 4), b_off = up(((size_t)npairs + 1) * 8), b_slot = up(tot * 4);
    const int nhyp = g->hypotheses > 0 ? g->hypotheses : 256;
    const size_t b_Fh = up((size_t)npairs * nhyp * 36);
    HIPCHK(c, c->d_geom.ensure(b_gp + b_sp + 2 * b_idx + b_co + b_Fh + 2 * b_F + 2 * b_cnt + b_off + 4 * b_slot));
    char* w = c->d_geom.as<char>();
    GeomPair* d_gp = reinterpret_cast<GeomPair*>(w); w += b_gp;
    SiftPair* d_sp = reinterpret_cast<SiftPair*>(w); w += b_sp;
    const int32_t* d_i0 = resident ? c->slots[c->last_slot].d_tight0.as<int32_t>() : reinterpret_cast<int32_t*>(w); w += b_idx;
    const int32_t* d_i1 = resident ? c->slots[c->last_slot].d_tight1.as<int32_t>() : reinterpret_cast<int32_t*>(w); w += b_idx;
    float4* d_co = reinterpret_cast<float4*>(w); w += b_co;
    float* d_Fh = reinterpret_cast<float*>(w); w += b_Fh;
    float* d_Fb = reinterpret_cast<float*>(w); w += b_F;
    float* d_Fo = reinterpret_cast<float*>(w); w += b_F;
    int32_t* d_bc = reinterpret_cast<int32_t*>(w); w += b_cnt;
    int32_t* d_cnt = reinterpret_cast<int32_t*>(w); w += b_cnt;
    int64_t* d_off = reinterpret_cast<int64_t*>(w); w += b_off;
    int32_t* d_s0 = reinterpret_cast<int32_t*>(w); w += b_slot;
    int32_t* d_s1 = reinterpret_cast<int32_t*>(w); w += b_slot;
    int32_t* d_t0 = reinterpret_cast<int32_t*>(w); w += b_slot;
    int32_t* d_t1 = reinterpret_cast<int32_t*>(w); w += b_slot;
    HIPCHK(c, hipMemcpyAsync(d_gp, gp.data(), sizeof(GeomPair) * npairs, hipMemcpyHostToDevice, c->stream));
    if (!resident && total > 0) {
        HIPCHK(c, hipMemcpyAsync(const_cast<int32_t*>(d_i0), idx0, (size_t)total * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(const_cast<int32_t*>(d_i1), idx1, (size_t)total * 4, hipMemcpyHostToDevice, c->stream));
    }
    GeomRule rule;
    rule.max_sampson = g->max_sampson;
    rule.hypotheses = nhyp;
    rule.seed = g->seed;
    rule.min_inliers = g->min_inliers;
    rule.first_pair = (unsigned long long)g->first_pair;
    const bool bracket = c->timing && c->inflight == 0;   // the slot's events belong to a submission in flight otherwise
    if (bracket) {
        c->R().ev_valid = false;
        c->timing_slot = c->rs;
        HIPCHK(c, hipEventRecord(c->R().ev0, c->stream));
    }
    launch_geom_verify(d_gp, npairs, max_matches, d_i0, d_i1, rule, d_co, d_Fh, d_Fb, d_bc, d_Fo, d_cnt, d_s0, d_s1, d_sp, c->stream);
    if (bracket) {
        HIPCHK(c, hipEventRecord(c->R().ev1, c->stream));
        c->R().ev_valid = true;
    }
    launch_pack(d_sp, npairs, d_cnt, d_off, d_s0, d_s1, d_t0, d_t1, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(in_offsets, d_off, sizeof(int64_t) * (npairs + 1), hipMemcpyDeviceToHost, c->stream));
    if (F) HIPCHK(c, hipMemcpyAsync(F, d_Fo, (size_t)npairs * 36, hipMemcpyDeviceToHost, c->stream));
    if (nf) HIPCHK(c, hipMemcpyAsync(nf, d_cnt, (size_t)npairs * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // the pair records and lists above are host memory of this call
    const long long got = in_offsets[npairs];
    if (got > cap) return fail(c, VSM_ECAP, "%lld inlier matches do not fit a buffer of %lld", got, cap);
    if (got > 0) {
        if (in_idx0) HIPCHK(c, hipMemcpyAsync(in_idx0, d_t0, (size_t)got * 4, hipMemcpyDeviceToHost, c->stream));
        if (in_idx1) HIPCHK(c, hipMemcpyAsync(in_idx1, d_t1, (size_t)got * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return got;
}

extern "C" int vsm_last_submission_pairs(const vsm_ctx* c) { return c ? c->slots[c->timing_slot].query_pairs : VSM_EINVAL; }

extern "C" int vsm_set_scratch_budget(vsm_ctx* c, long long bytes)
{
    if (!c) return VSM_EINVAL;
    if (bytes < (1ll << 20)) return fail(c, VSM_EINVAL, "scratch budget %lld is below 1 MiB", bytes);
    c->scratch_budget = (size_t)bytes;
    return VSM_OK;
}

extern "C" int vsm_set_f32_mode(vsm_ctx* c, int mode)
{
    if (!c) return VSM_EINVAL;
    if (mode != VSM_F32_FAST_EXACT && mode != VSM_F32_MFMA_F32) return fail(c, VSM_EINVAL, "unknown f32 mode %d", mode);
    c->f32_mode = mode;
    return VSM_OK;
}

extern "C" int vsm_set_f32_reverse(vsm_ctx* c, int mode)
{
    if (!c) return VSM_EINVAL;
    if (mode < VSM_F32_REVERSE_AUTO || mode > VSM_F32_REVERSE_ALL) return fail(c, VSM_EINVAL, "unknown reverse mode %d", mode);
    c->f32_reverse = mode;
    return VSM_OK;
}

extern "C" int vsm_last_f32_path(const vsm_ctx* c) { return c ? c->last_f32_path : VSM_EINVAL; }
extern "C" int vsm_last_f32_reverse(const vsm_ctx* c) { return c ? c->last_f32_reverse : VSM_EINVAL; }

extern "C" int vsm_knn2_device(vsm_ctx* c, int set_q, int set_db, int metric, void* packed_dev)
{
    if (!c) return VSM_EINVAL;
    if (!packed_dev) return fail(c, VSM_EINVAL, "output buffer is NULL");
    SetKind kind;
    int rc = set_kind_of_pair(c, set_q, set_db, &kind);
    if (rc) return rc;
    if (kind != SET_F32) return fail(c, VSM_EUNSUPPORTED, "vsm_knn2_device takes float sets");
    HIPCHK(c, hipSetDevice(c->device));
    const DescSet* A = get_set(c, set_q);
    const DescSet* B = get_set(c, set_db);
    const size_t nq = (size_t)A->n;
    if (nq == 0) return 0;
    int32_t* out = static_cast<int32_t*>(packed_dev);
    if (B->n == 0) {
        HIPCHK(c, hipMemsetAsync(out, 0, nq * 8, c->stream));
        HIPCHK(c, hipMemsetAsync(out + 2 * nq, 0xFF, nq * 8, c->stream));   // rows = -1
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return A->n;
    }
    F32Result r;
    if ((rc = run_f32_knn(c, A, B, metric, false, &r)) != VSM_OK) return rc;
    // the four arrays side by side in the caller's buffer: one kernel instead of four copy commands
    WordSegments seg{};
    const void* src[4] = {r.k1, r.k2, r.j1, r.j2};
    for (int k = 0; k < 4; ++k) {
        seg.src[k] = static_cast<const uint32_t*>(src[k]);
        seg.count[k] = (int)nq;
    }
    seg.n = 4;
    launch_gather_words(seg, reinterpret_cast<uint32_t*>(out), c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return A->n;
}

extern "C" int vsm_merge_top2_device(vsm_ctx* c, const void* packed_dev, const int64_t* base, int nslices, int nq,
                                     int metric, float* ok1, float* ok2, int64_t* oi1, int64_t* oi2)
{
    if (!c) return VSM_EINVAL;
    if (!packed_dev || !base || nslices <= 0 || nq < 0 || !ok1 || !ok2 || !oi1 || !oi2) return fail(c, VSM_EINVAL, "bad argument");
    if (nq == 0) return VSM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    // workspace: base[nslices + 1], then the four outputs
    const size_t base_bytes = ((size_t)(nslices + 1) * 8 + 255) & ~(size_t)255;
    HIPCHK(c, c->d_stage.ensure(base_bytes + (size_t)nq * 24));
    char* ws = c->d_stage.as<char>();
    int64_t* d_base = reinterpret_cast<int64_t*>(ws);
    int64_t* d_i1 = reinterpret_cast<int64_t*>(ws + base_bytes);
    int64_t* d_i2 = d_i1 + nq;
    float* d_k1 = reinterpret_cast<float*>(d_i2 + nq);
    float* d_k2 = d_k1 + nq;
    HIPCHK(c, hipMemcpyAsync(d_base, base, (size_t)(nslices + 1) * 8, hipMemcpyHostToDevice, c->stream));
    launch_merge_top2(static_cast<const int32_t*>(packed_dev), d_base, nslices, nq, metric, d_k1, d_k2, d_i1, d_i2, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(ok1, d_k1, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(ok2, d_k2, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(oi1, d_i1, (size_t)nq * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(oi2, d_i2, (size_t)nq * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VSM_OK;
}

extern "C" int vsm_merge_top2_f32(const float* k1, const float* k2, const int32_t* i1, const int32_t* i2,
                                  const int64_t* base, int nslices, int nq, int metric,
                                  float* ok1, float* ok2, int64_t* oi1, int64_t* oi2)
{
    if (!k1 || !k2 || !i1 || !i2 || !base || nslices <= 0 || nq < 0 || !ok1 || !ok2 || !oi1 || !oi2) return VSM_EINVAL;
    const bool dot = metric == VSM_METRIC_DOT;
    for (int q = 0; q < nq; ++q) {
        float b1 = 0, b2 = 0;
        int64_t x1 = -1, x2 = -1;
        for (int s = 0; s < nslices; ++s) {
            for (int r = 0; r < 2; ++r) {
                const size_t o = (size_t)s * nq + q;
                const int32_t li = r ? i2[o] : i1[o];
                if (li < 0) continue;
                const float k = r ? k2[o] : k1[o];
                const int64_t id = base[s] + li;
                auto better = [&](float a, float b) { return dot ? a > b : a < b; };
                if (x1 < 0 || better(k, b1) || (k == b1 && id < x1)) {
                    b2 = b1; x2 = x1;
                    b1 = k; x1 = id;
                } else if (x2 < 0 || better(k, b2) || (k == b2 && id < x2)) {
                    b2 = k; x2 = id;
                }
            }
        }
        ok1[q] = b1; ok2[q] = b2; oi1[q] = x1; oi2[q] = x2;
    }
    return VSM_OK;
}

extern "C" int vsm_nearest_embed(vsm_ctx* c, const float* x, int batch, int dim, int npts,
                                 const float* emb, int k, int64_t* argmin, float* result)
{
    if (!c) return VSM_EINVAL;
    if (!x || !emb || !argmin || batch < 0 || npts < 0 || k <= 0) return fail(c, VSM_EINVAL, "bad argument");
    // queries: every (batch, point) column of x; codes: the columns of emb
    const int code_set = vsm_upload_f32(c, emb, k, dim, VSM_LAYOUT_DIM_MAJOR);
    if (code_set < 0) return code_set;
    std::vector<int32_t> idx(std::max(npts, 1));
    std::vector<float> E;
    if (result) {
        E.resize((size_t)k * dim);
        for (int cc = 0; cc < k; ++cc)
            for (int d = 0; d < dim; ++d) E[(size_t)cc * dim + d] = emb[(size_t)d * k + cc];
    }
    int rc = VSM_OK;
    for (int b = 0; b < batch && rc >= 0; ++b) {
        const int qs = vsm_upload_f32(c, x + (size_t)b * dim * npts, npts, dim, VSM_LAYOUT_DIM_MAJOR);
        if (qs < 0) { rc = qs; break; }
        rc = vsm_knn2(c, qs, code_set, VSM_METRIC_L2, idx.data(), nullptr, nullptr, nullptr);
        vsm_release_set(c, qs);
        if (rc < 0) break;
        for (int pt = 0; pt < npts; ++pt) {
            argmin[(size_t)b * npts + pt] = idx[pt];
            if (result)
                for (int d = 0; d < dim; ++d)
                    result[((size_t)b * dim + d) * npts + pt] = E[(size_t)idx[pt] * dim + d];
        }
    }
    vsm_release_set(c, code_set);
    return rc < 0 ? rc : VSM_OK;
}

// ---------------------------------------------------------------------------------
// image-pair shortlist from a visual vocabulary
// ---------------------------------------------------------------------------------

namespace {

// Nearest vocabulary row of every descriptor row of sets[0..nimg): submissions of (image, vocabulary)
// pairs with every test off; after each one `use(first image, images, result slot)` queues the caller's
// kernels on the lists (pair p of the slot = image first + p, d_tight1 = the word of every row in order).
template <class Use>
int assign_words(vsm_ctx* c, const int32_t* sets, int nimg, int vocab_set, Use use)
{
    vsm_params prm{};
    prm.ratio_num = 0; prm.ratio_den = 1; prm.mutual = 0;
    prm.max_dist_sq = -1; prm.min_score = -INFINITY; prm.max_dist_sq_f = -1.0f;
    const int64_t rows_per_group = (int64_t)std::max<size_t>(c->scratch_budget / (4 * kScratchPerRow), 1);
    std::vector<int32_t> pl;
    for (int g0 = 0; g0 < nimg;) {
        pl.clear();
        int64_t rows = 0;
        int g1 = g0;
        while (g1 < nimg && g1 - g0 < 4096) {
            const int n = get_set(c, sets[g1])->n;
            if (g1 > g0 && rows + n > rows_per_group) break;
            rows += n;
            pl.push_back(sets[g1]);
            pl.push_back(vocab_set);
            ++g1;
        }
        const long long got = vsm_match_pairs_device(c, pl.data(), g1 - g0, &prm, VSM_METRIC_L2);
        if (got < 0) return (int)got;
        if (got != rows) return fail(c, VSM_ENODEV, "internal: %lld word assignments for %lld rows", got, (long long)rows);
        use(g0, g1 - g0, c->slots[c->last_slot]);
        g0 = g1;
    }
    c->last_npairs = 0;   // the resident lists were an intermediate
    c->last_total = 0;
    return VSM_OK;
}

}  // namespace

extern "C" long long vsm_shortlist_pairs(vsm_ctx* c, const int32_t* sets, int nimg, int vocab_set, int topk,
                                         int32_t* pairs, long long cap, int32_t* scores)
{
    if (!c) return VSM_EINVAL;
    if (nimg < 0 || (nimg > 0 && !sets) || topk < 1 || cap < 0 || (cap > 0 && !pairs)) return fail(c, VSM_EINVAL, "bad argument");
    if (nimg > 16384) return fail(c, VSM_EINVAL, "%d images in one shortlist (limit 16384)", nimg);
    if (c->inflight) return fail(c, VSM_EINVAL, "collect the submissions in flight first");
    const DescSet* V = get_set(c, vocab_set);
    if (!V || (V->kind != SET_U8 && V->kind != SET_F32) || V->n < 1)
        return fail(c, VSM_EINVAL, "set %d is not a vocabulary (a descriptor set of at least one row)", vocab_set);
    const int K = V->n;
    constexpr int kMaxRows = 1 << 26;   // 15 (largest weight for 16384 images) x rows stays below 2^31
    for (int i = 0; i < nimg; ++i) {
        const DescSet* A = get_set(c, sets[i]);
        if (!A || A->kind != V->kind || A->dim != V->dim)
            return fail(c, VSM_EINVAL, "image %d (set %d) is not a descriptor set of the vocabulary's kind and dimension", i, sets[i]);
        if (A->n > kMaxRows) return fail(c, VSM_EINVAL, "image %d has %d rows (limit %d)", i, A->n, kMaxRows);
    }
    if (nimg == 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t n2 = (size_t)nimg * nimg;
    const size_t max_pairs = std::min<size_t>((size_t)nimg * (size_t)std::min(topk, nimg), n2 / 2 + 1);
    const size_t b_hist = up((size_t)nimg * K * 4), b_w = up((size_t)K * 4), b_sc = up(n2 * 4), b_mark = up(n2);
    const size_t b_cnt = up((size_t)nimg * 4), b_off = up(((size_t)nimg + 1) * 8), b_pairs = up(max_pairs * 8);
    const size_t need = b_hist + b_w + b_sc + b_mark + b_cnt + b_off + b_pairs;
    if (need > c->scratch_budget)
        return fail(c, VSM_EINVAL, "shortlist of %d images over %d words needs %zu bytes of scratch (limit %zu)", nimg, K, need, c->scratch_budget);
    HIPCHK(c, c->d_short.ensure(need));
    char* w = c->d_short.as<char>();
    uint32_t* d_hist = reinterpret_cast<uint32_t*>(w); w += b_hist;
    uint32_t* d_w = reinterpret_cast<uint32_t*>(w); w += b_w;
    int32_t* d_sc = reinterpret_cast<int32_t*>(w); w += b_sc;
    uint8_t* d_mark = reinterpret_cast<uint8_t*>(w); w += b_mark;
    int32_t* d_cnt = reinterpret_cast<int32_t*>(w); w += b_cnt;
    int64_t* d_off = reinterpret_cast<int64_t*>(w); w += b_off;
    int32_t* d_pairs = reinterpret_cast<int32_t*>(w); w += b_pairs;
    HIPCHK(c, hipMemsetAsync(d_hist, 0, b_hist, c->stream));
    HIPCHK(c, hipMemsetAsync(d_mark, 0, b_mark, c->stream));
    int rc = assign_words(c, sets, nimg, vocab_set, [&](int g0, int m, const vsm_ctx::ResultSlot& R) {
        launch_bow_count(R.d_offsets.as<int64_t>(), R.d_tight1.as<int32_t>(), g0, m, K, d_hist, c->stream);
    });
    if (rc != VSM_OK) return rc;
    launch_bow_shortlist(d_hist, nimg, K, std::min(topk, nimg), d_w, d_sc, d_mark, d_cnt, d_off, d_pairs, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, c->h_small.ensure(64));
    long long* tot = c->h_small.as<long long>();
    HIPCHK(c, hipMemcpyAsync(tot, d_off + nimg, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    if (scores) HIPCHK(c, hipMemcpyAsync(scores, d_sc, n2 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const long long npairs = *tot;
    if (npairs > cap) return fail(c, VSM_ECAP, "%lld pairs do not fit a buffer of %lld", npairs, cap);
    if (npairs > 0) {
        HIPCHK(c, hipMemcpyAsync(pairs, d_pairs, (size_t)npairs * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return npairs;
}

extern "C" int vsm_train_vocabulary(vsm_ctx* c, const int32_t* sets, int nimg, int nwords, int iterations,
                                    uint8_t* words)
{
    if (!c) return VSM_EINVAL;
    if (nimg < 0 || (nimg > 0 && !sets) || nwords < 1 || iterations < 0 || !words) return fail(c, VSM_EINVAL, "bad argument");
    if (c->inflight) return fail(c, VSM_EINVAL, "collect the submissions in flight first");
    int64_t total = 0;
    for (int i = 0; i < nimg; ++i) {
        const DescSet* A = get_set(c, sets[i]);
        if (!A || A->kind != SET_U8) return fail(c, VSM_EINVAL, "image %d (set %d) is not a u8 descriptor set", i, sets[i]);
        total += A->n;
    }
    if (total == 0) return 0;
    if (total > (1 << 24)) return fail(c, VSM_EINVAL, "%lld descriptor rows (limit %d: sums of bytes are kept in 32 bits)", (long long)total, 1 << 24);
    const int K = (int)std::min<int64_t>(nwords, total);
    HIPCHK(c, hipSetDevice(c->device));
    // first values: rows floor(k * total / K) of the concatenation of the sets, k = 0 .. K-1
    std::vector<const int8_t*> frags(K);
    std::vector<int32_t> rows(K);
    {
        int img = 0;
        int64_t first = 0;   // global index of row 0 of image img
        for (int k = 0; k < K; ++k) {
            const int64_t g = (int64_t)(((__int128)k * total) / K);
            while (g >= first + get_set(c, sets[img])->n) first += get_set(c, sets[img++])->n;
            frags[k] = static_cast<const int8_t*>(get_set(c, sets[img])->frag);
            rows[k] = (int32_t)(g - first);
        }
    }
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_words = up((size_t)K * kDescDim), b_sums = up((size_t)K * kDescDim * 4), b_cnt = up((size_t)K * 4);
    const size_t b_frags = up((size_t)K * sizeof(void*)), b_rows = up((size_t)K * 4);
    HIPCHK(c, c->d_short.ensure(b_words + b_sums + b_cnt + b_frags + b_rows));
    char* w = c->d_short.as<char>();
    uint8_t* d_words = reinterpret_cast<uint8_t*>(w); w += b_words;
    unsigned* d_sums = reinterpret_cast<unsigned*>(w); w += b_sums;
    unsigned* d_cnt = reinterpret_cast<unsigned*>(w); w += b_cnt;
    const int8_t** d_frags = reinterpret_cast<const int8_t**>(w); w += b_frags;
    int32_t* d_rows = reinterpret_cast<int32_t*>(w); w += b_rows;
    HIPCHK(c, hipMemcpyAsync(d_frags, frags.data(), (size_t)K * sizeof(void*), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_rows, rows.data(), (size_t)K * 4, hipMemcpyHostToDevice, c->stream));
    launch_kmeans_seed(d_frags, d_rows, K, d_words, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));   // frags and rows are host memory of this call
    for (int it = 0; it < iterations; ++it) {
        const int vocab = vsm_upload_u8_device(c, d_words, K);
        if (vocab < 0) return vocab;
        HIPCHK(c, hipMemsetAsync(d_sums, 0, b_sums + b_cnt, c->stream));
        const int rc = assign_words(c, sets, nimg, vocab, [&](int, int m, const vsm_ctx::ResultSlot& R) {
            launch_kmeans_accumulate(c->d_pairs.as<SiftPair>(), m, R.d_offsets.as<int64_t>(), R.d_tight1.as<int32_t>(), K,
                                     d_sums, d_cnt, c->stream);
        });
        if (rc != VSM_OK) {
            vsm_release_set(c, vocab);
            return rc;
        }
        launch_kmeans_update(d_sums, d_cnt, K, d_words, c->stream);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        vsm_release_set(c, vocab);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(words, d_words, (size_t)K * kDescDim, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return K;
}

extern "C" int vsm_sample_rows_u8(vsm_ctx* c, const int32_t* sets, int nimg, int stride)
{
    if (!c) return VSM_EINVAL;
    if (nimg < 0 || (nimg > 0 && !sets) || stride < 1) return fail(c, VSM_EINVAL, "bad argument");
    int64_t total = 0;
    for (int i = 0; i < nimg; ++i) {
        const DescSet* A = get_set(c, sets[i]);
        if (!A || A->kind != SET_U8) return fail(c, VSM_EINVAL, "image %d (set %d) is not a u8 descriptor set", i, sets[i]);
        total += A->n;
    }
    const int64_t m64 = (total + stride - 1) / stride;
    if (m64 > 32767ll * kVerifyBlock) return fail(c, VSM_EINVAL, "%lld sampled rows exceed the size of one set", (long long)m64);
    const int m = (int)m64;
    if (m == 0) return vsm_upload_u8(c, nullptr, 0);
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<const int8_t*> frags(m);
    std::vector<int32_t> rows(m);
    {
        int img = 0;
        int64_t first = 0;
        for (int k = 0; k < m; ++k) {
            const int64_t g = (int64_t)k * stride;
            while (g >= first + get_set(c, sets[img])->n) first += get_set(c, sets[img++])->n;
            frags[k] = static_cast<const int8_t*>(get_set(c, sets[img])->frag);
            rows[k] = (int32_t)(g - first);
        }
    }
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_out = up((size_t)m * kDescDim), b_frags = up((size_t)m * sizeof(void*)), b_rows = up((size_t)m * 4);
    HIPCHK(c, c->d_short.ensure(b_out + b_frags + b_rows));
    char* w = c->d_short.as<char>();
    uint8_t* d_out = reinterpret_cast<uint8_t*>(w); w += b_out;
    const int8_t** d_frags = reinterpret_cast<const int8_t**>(w); w += b_frags;
    int32_t* d_rows = reinterpret_cast<int32_t*>(w);
    HIPCHK(c, hipMemcpyAsync(d_frags, frags.data(), (size_t)m * sizeof(void*), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_rows, rows.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
    launch_kmeans_seed(d_frags, d_rows, m, d_out, c->stream);
    HIPCHK(c, hipStreamSynchronize(c->stream));   // frags and rows are host memory of this call
    HIPCHK(c, hipGetLastError());
    return vsm_upload_u8_device(c, d_out, m);
}
