// hvs.hip -- host side of libhvs.so: context, HBM residency, launch plan, C ABI (include/hvs.h).
//
// Build (see __graft_entry__.build):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared hvs.hip -o libhvs.so
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <sched.h>

#include <rocprim/rocprim.hpp>

#include "../../include/hvs.h"
#include "hvs_kernels.h"
#include "hvs_filter.h"

namespace {

thread_local std::string g_global_err;

}  // namespace

// Workspace of ONE query batch and the stream it runs on -- a "lane".  A context owns two (round 4): the last level of batch b
// ends with its re-scoring (HBM-bound gathers) and the final merge (latency-bound), both of which leave the matrix pipes idle,
// and batch b+1 begins with preparation, the exact seed and two small filter levels that cannot fill the chip; with batch
// b+1 on the other lane's stream the two ends run side by side.  hvs_ctx IS its main lane (base class: every `c->fb`,
// `c->stream` ... below means "the lane this batch runs on"); run_queries swaps the spare lane in for every other batch.
struct HvsLane {
    hipStream_t stream = nullptr;
    // sort of the batch's queries
    uint64_t *d_keys = nullptr, *d_keys_sorted = nullptr;
    uint32_t *d_qidx = nullptr, *d_qorder = nullptr;
    uint32_t *d_qra = nullptr, *d_qrb = nullptr;  // position range of each query of the batch, by batch-local index
    void* d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    uint32_t batch_cap = 0;
    // exact engine: per-(query, chunk) candidate lists
    uint64_t* d_cand = nullptr;
    uint32_t* d_cand_cnt = nullptr;
    size_t cand_lists = 0;
    // filter engines: per-slot / per-group state
    HvsBatch fb{};
    uint32_t fb_slots_cap = 0;
    size_t fb_cand_entries = 0, fb_pair_entries = 0;  // capacity of fb.cand / fb.pairs in entries
    uint32_t* d_layout = nullptr;
    // work-item lists of the current batch (HvsItems): per-quad block ranges, per-segment counts / offsets, the list
    uint32_t *d_qlo = nullptr, *d_qhi = nullptr, *d_segcnt = nullptr, *d_segoff = nullptr, *d_lvloff = nullptr, *d_cursor = nullptr;
    uint32_t* d_items = nullptr;
    size_t items_cap = 0;
    uint32_t quads_cap = 0, segs_cap = 0;
    HvsSegs segs{};  // of the current batch
    uint32_t class_counts[5] = {0, 0, 0, 0, 0};  // queries per predicate class in the current batch
};

struct hvs_ctx : HvsLane {
    int device = 0;
    HvsLane spare;                 // the second lane (its buffers are allocated by the first call that has two batches)
    bool lanes_failed = false;     // no room for a second workspace: calls stay on one lane
    hipEvent_t ev_lane = nullptr;  // end of the spare lane's latest batch
    // end of the LAST level's filter launch of the latest batch on the main / the spare lane: the next batch (other lane) starts
    // behind it -- its head then runs beside this batch's last re-scoring and final merge, not beside its filter launches (two
    // crews of filter workgroups streaming different tiles through the same L2s: measured 12 % slower than one lane)
    hipEvent_t ev_fdone[2] = {nullptr, nullptr};
    hipEvent_t ev_fdone_cur = nullptr;  // the one the batch being enqueued records (nullptr: none)
    // likewise the end of the filter launch of the level BEFORE the last: the next batch's preparation (query sort, slot layout,
    // fragments, work-item lists -- memory- and latency-bound) starts behind it and runs beside this batch's re-scoring of
    // that level; its compute-heavy part (exact seed, low filter levels) waits for ev_fdone (`gate_heavy`)
    hipEvent_t ev_pdone[2] = {nullptr, nullptr};
    hipEvent_t ev_pdone_cur = nullptr;
    hipEvent_t gate_heavy = nullptr;    // what the batch being enqueued waits for between its preparation and its seed
    int engine = HVS_ENGINE_AUTO;
    bool scalar_order = false;  // baseline engine's summation order (exact engine only)
    uint32_t k = HVS_KNN;       // neighbours per query (hvs_set_k; the reference's KNN_LIMIT, optimized_impl.h:26)
    int cap = 256;              // candidate-list capacity the kernels run with: 256 (k <= 128) or 512 (k <= 256)
    bool padding = true;        // pad answers with the last rows of D (off: partial answers of a data shard)
    std::string err;

    // data set, raw rows n x 102 (the io.h layout) resident in HBM
    float* d_data = nullptr;
    uint32_t n = 0;
    double load_ms = 0.0;

    // resident queries + results
    float* d_q = nullptr;
    uint32_t nq = 0, nq_cap = 0;
    uint32_t* d_out_ids = nullptr;
    float* d_out_dists = nullptr;
    uint32_t res_cap = 0;

    unsigned long long* d_counters = nullptr;

    // ---- MFMA engine: index over D (two orderings) ...
    bool have_index = false;
    HvsLevels lv{};                       // same block count for both orderings
    uint64_t *d_keys_ct = nullptr, *d_keys_t = nullptr;   // sorted attribute keys
    uint32_t *d_perm_ct = nullptr, *d_perm_t = nullptr;   // position -> original row id
    uint4 *d_tiles_ct = nullptr, *d_tiles_t = nullptr;    // A-operand tiles (BF16 or INT8), level-interleaved
    uint4 *d_nrm_ct = nullptr, *d_nrm_t = nullptr;        // INT8 format: the rows' accumulator inits, [nblk][32] int32
    uint32_t *d_bpos_ct = nullptr, *d_bpos_t = nullptr;   // storage index -> block
    HvsBounds* d_bounds = nullptr;
    HvsQuant* d_quant = nullptr;                          // INT8 format: centre and scale
    int tile_fmt = HVS_FMT_NONE;                          // format of the tiles currently built
    int planned_fmt = HVS_FMT_BF16;                       // what HVS_ENGINE_AUTO uses for this data set
    bool i8_usable = false;
    bool i8_rot = false;       // the INT8 centre / scale (d_quant) and tiles live in the rotated space (HvsQuant::rot)
    bool i8_rot_built = false; // ... and the INT8 tiles currently built were cut from it
    bool i8_rejected = false;  // the INT8 tiles were built and their bound was unusable: do not try again
    bool f16_rejected = false; // likewise the FP16 tiles (components beyond the half-precision range)
    double index_ms = 0.0;
    bool index_too_large = false;  // more than 2^27 rows: no filter index (hvs_timing.flags says so)
    // (the per-batch state of the filter engines lives in the lanes)
    int num_cus = 256;
    uint32_t *d_ovf_list = nullptr, *d_ovf_count = nullptr;      // queries for the exact engine; d_ovf_count[0..1] = exact, retry
    uint32_t* d_retry_list = nullptr;                            // queries whose guessed threshold was not verified
    uint32_t* d_demote_list = nullptr;                           // a call's exact list, moved aside when the tile format changes under it
    uint32_t demote_cap = 0, demoted_queries = 0;
    uint32_t fallback_queries = 0, retry_queries = 0;
    HvsGuessTable guess_tab[13]{};  // order statistics of the guessed thresholds for k = guess_k: [0] proven (retry batches),
    bool guess_have[13] = {};       // [p] failure target 10^-p
    uint32_t guess_k = 0;

    hipEvent_t ev_q0 = nullptr, ev_q1 = nullptr;
    // start/stop event pairs around the dominant kernel's launches of the current call (grown on demand: a call
    // of 10^7 queries is 10 batches x 14 levels)
    std::vector<hipEvent_t> ev_k;
    int n_launch_events = 0;  // pairs used by the current call
    uint32_t untimed_launches = 0;
    bool timing_valid = false;
    hvs_timing timing{};
    double host_ms = 0.0;     // wall time of the last hvs_query (host memory in -> host memory out)

    // queries whose candidate lists overflowed (-> exact engine) or whose guessed threshold failed its check (-> a filter
    // batch with a proven last threshold), collected over all batches of a call; they are re-run when the call's results
    // are first needed (resolve_overflow) -- no host synchronisation inside a batch
    uint32_t* h_ovf = nullptr;  // pinned: [0] exact, [1] retry
    bool ovf_pending = false;
    uint32_t pend_sn = 0;

    // host <-> device pipeline of hvs_query: copy streams + rings of pinned staging slots
    hipStream_t s_in = nullptr, s_out = nullptr;
    static constexpr uint32_t kStageQ = 65536;  // queries per staging slot
    static constexpr int kRing = 4;
    uint32_t in_cap_q[kRing] = {0, 0, 0, 0}, out_cap_q[kRing] = {0, 0, 0, 0}, outd_cap_q[kRing] = {0, 0, 0, 0};  // slot sizes in queries
    float* h_in[kRing] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t* h_out_ids[kRing] = {nullptr, nullptr, nullptr, nullptr};
    float* h_out_dists[kRing] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_in[kRing] = {nullptr, nullptr, nullptr, nullptr}, ev_out[kRing] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_batch = nullptr, ev_stage = nullptr;
    uint32_t reserve_nq = 0;  // hvs_reserve: queries per call the caller announced
    uint32_t stage_k = 0;     // k the pinned result slots were sized for
    bool dma_warm = false;    // the copy engines of s_in / s_out have moved a DMA-sized piece
    // planner probe (probe_format): the probe batch of 1024 queries stands for the large batches the format will serve, so it
    // runs with THEIR failure target -- and with their list capacity (HVS_FCAP entries per query and level): a format whose band
    // overflows that capacity fails in production too
    uint32_t force_pfail = 0;

    // multi-GPU root (hvs_create_multi): owns one leaf context per GPU; D is replicated, the queries of a call are cut
    // into one contiguous range per leaf (optimized_parallel.hpp:91: iterations are independent) and every leaf
    // writes its block of ids straight into its slice of the caller's buffer
    void* trace = nullptr;  // HVS_TRACE: the HostTrace of the running hvs_load_data
    std::vector<int> node_cpus;  // CPUs of the NUMA node this GPU hangs off (empty: unknown); the leaf's host thread runs there
    std::vector<hvs_ctx*> kids;
    std::vector<uint32_t> kid_q0;  // resident queries: first global index of each leaf's range (kids.size() + 1 entries)
    int gather_mode = 0;           // HVS_GATHER_DIRECT / HVS_GATHER_PEER
};

namespace {

uint32_t env_u32(const char* name, uint32_t dflt, uint32_t lo, uint32_t hi)
{
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    const unsigned long x = std::strtoul(v, nullptr, 10);
    return x < lo ? lo : (x > hi ? hi : (uint32_t)x);
}
// queries answered per pass over D; HVS_EXACT_BATCH / HVS_MFMA_BATCH override (tests use small batches)
const uint32_t kBatch = env_u32("HVS_EXACT_BATCH", 65536u, 64u, 1u << 20);
const uint32_t kBatchMfma = env_u32("HVS_MFMA_BATCH", 1u << 21, 128u, 1u << 21);  // (2^21: +2.2 % queries/s over 2^20, 34 GB of batch state)
// re-scoring blocks per group (each stages the group's 128 queries in LDS): HVS_RESCORE_BLOCKS overrides
const uint32_t kRescoreBlocks = env_u32("HVS_RESCORE_BLOCKS", 0u, 0u, 64u);  // 0: chosen per batch
// small batches: level 0 (the exact seed kernel) is cut into chunks until about this many waves are in flight
const uint32_t kSeedWaves = env_u32("HVS_SEED_WAVES", 16384u, 64u, 1u << 20);
// exact full scan: rows through LDS (1) or through the scalar cache (0); HVS_SCAN_LDS overrides for A/B runs
const bool kScanRowsThroughLds = env_u32("HVS_SCAN_LDS", 1u, 0u, 1u) != 0u;
// INT8 tiles are built for v_mfma_i32_16x16x64_i8 (HVS_FMT_I8X16: 1.16x the pair rate of the 32x32x32 shape in the
// filter loop, scripts/mfma_shape_lab.hip); HVS_I8_SHAPE=32 selects the 32x32x32 layout (HVS_FMT_I8) for A/B runs
const int kI8Fmt = env_u32("HVS_I8_SHAPE", 16u, 16u, 32u) == 32u ? HVS_FMT_I8 : HVS_FMT_I8X16;
// Level radices of the index (powers of two; see "Guessed thresholds" at hvs_k_merge): HVS_GUESS=0 restores round 2's
// doubling levels with proven thresholds for A/B runs
const bool kGuess = env_u32("HVS_GUESS", 1u, 0u, 1u) != 0u;
uint32_t pow2_floor(uint32_t x) { uint32_t p = 2u; while (p * 2u <= x) p *= 2u; return p; }
const uint32_t kRadixLast = kGuess ? pow2_floor(env_u32("HVS_RADIX_LAST", HVS_RADIX_LAST, 2u, 64u)) : 2u;
const uint32_t kRadixMid = kGuess ? pow2_floor(env_u32("HVS_RADIX_MID", HVS_RADIX_MID, 2u, 64u)) : 2u;
// HVS_RADICES="4,8,32": the radices of the last levels, last level first (A/B runs); HVS_RADIX_MID continues behind them
struct RadixPlan {
    uint32_t r[16] = {};
    bool set = false;
    RadixPlan()
    {
        const char* v = std::getenv("HVS_RADICES");
        if (!v || !*v || !kGuess) return;
        int k = 0;
        bool ok = true;
        while (*v && k < 14) {
            char* end = nullptr;
            const unsigned long x = std::strtoul(v, &end, 10);
            if (end == v) {  // not a number ("4;8", "4x"): the whole variable is ignored
                ok = false;
                break;
            }
            v = end;
            if (x >= 2 && x <= 64) r[k++] = pow2_floor((uint32_t)x);
            while (*v == ',' || *v == ' ') ++v;
        }
        if (!ok) {
            for (uint32_t& x : r) x = 0u;
            k = 0;
        }
        set = k > 0;
    }
};
const RadixPlan kRadixPlan;
// smallest order statistic a guessed threshold may use, and -log10 of the chance that one guess leaves fewer than k rows
// below it (plan_guess)
const uint32_t kGuessMid = env_u32("HVS_GUESS_MID", 3u, 1u, 256u);
// HVS_GUESS_PFAIL unset (0): by batch size -- 10^-3 for batches of 2^18 queries and more, 10^-4 from 2^15, 10^-5 below: a
// retry batch costs a fixed ~0.8 ms of latency-bound rounds whatever its size, which a batch of 10^4 queries (2.6 ms)
// cannot afford every time while a batch of 2^21 (650 ms) gains 3 % from the tighter guesses
const uint32_t kGuessPfail = env_u32("HVS_GUESS_PFAIL", 0u, 0u, 12u);
// (round 4: 10^-6 below 2^15 queries -- measured on BASELINE configs[1]/[2], 10^4 queries per batch: a failed guess costs the call a
// host synchronisation and a retry batch of ~0.55 ms whose kernels are launch-bound; 10^-5 retried 0.3-0.45 queries per call,
// 10^-6 none in 20 calls for 11 % more candidates: 2.21 / 1.80 ms per call against 2.27 / 1.90, profiles/r04/configs12_sweeps.txt)
uint32_t guess_pfail_for(uint32_t nqb) { return kGuessPfail ? kGuessPfail : (nqb >= (1u << 18) ? 3u : (nqb >= (1u << 15) ? 4u : 6u)); }
constexpr uint32_t kMfmaMinRows = 32768;  // below this the exact engine is used by HVS_ENGINE_AUTO
constexpr uint32_t kIndexMinRows = 4096;  // below this no index is built (the exact engine scans all rows)

// two lanes (see HvsLane): HVS_LANES=0 keeps every batch on the main lane (A/B runs)
const bool kLanes = env_u32("HVS_LANES", 1u, 0u, 1u) != 0u;
void swap_lanes(hvs_ctx* c) { std::swap(static_cast<HvsLane&>(*c), c->spare); }
struct LaneGuard {  // the spare lane is swapped in for the duration of one batch, whatever way the batch ends
    hvs_ctx* c;
    bool on;
    ~LaneGuard()
    {
        if (on) swap_lanes(c);
    }
};
void free_lane(HvsLane& L)
{
    HvsBatch& B = L.fb;
    void* ptrs[] = {L.d_keys, L.d_keys_sorted, L.d_qidx, L.d_qorder, L.d_qra, L.d_qrb, L.d_sort_tmp, L.d_cand, L.d_cand_cnt,
                    B.qid, B.rank, B.ra, B.rb, B.gua, B.gub, B.gord, B.bfrag, B.theta, B.qn, B.normq, B.eq, B.nqb,
                    B.top, B.topcnt, B.tau, B.cand, B.candcnt, B.overflow, B.pairs, B.paircnt, B.goverflow,
                    L.d_layout, L.d_qlo, L.d_qhi, L.d_segcnt, L.d_segoff, L.d_lvloff, L.d_cursor, L.d_items};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (L.stream) (void)hipStreamDestroy(L.stream);
    L = HvsLane{};
}

// HVS_TRACE=1: hvs_query prints the host-side phases of a call (microseconds since its start) to stderr -- where a small
// call's wall time goes between the caller's buffers and the first / last kernel
const bool kTrace = env_u32("HVS_TRACE", 0u, 0u, 1u) != 0u;
struct HostTrace {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::string line;
    void mark(const char* what)
    {
        if (!kTrace) return;
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        char buf[96];
        std::snprintf(buf, sizeof(buf), " %s@%.0f", what, us);
        line += buf;
    }
    void flush(const char* tag, uint32_t nq)
    {
        if (kTrace) std::fprintf(stderr, "[hvs trace] %s nq=%u:%s\n", tag, nq, line.c_str());
    }
};

void trace_mark(hvs_ctx* c, const char* what)
{
    if (kTrace && c->trace) static_cast<HostTrace*>(c->trace)->mark(what);
}

int fail(hvs_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    return code;
}

#define HVS_HIP(ctx, call)                                                                                     \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return fail(ctx, e_ == hipErrorOutOfMemory ? HVS_ENOMEM : HVS_EHIP,                                \
                        std::string(#call) + ": " + hipGetErrorString(e_));                                    \
    } while (0)

template <typename T>
int dev_alloc(hvs_ctx* c, T** p, size_t count)
{
    if (*p) {
        (void)hipFree(*p);
        *p = nullptr;
    }
    if (count == 0) return HVS_OK;
    HVS_HIP(c, hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    return HVS_OK;
}

// Event pair around one launch of the dominant kernel (HIP events on the library's own stream: bench.py's roofline
// reads their sum).  begin returns the pair's index or -1 (events could not be created: the launch goes untimed).
int kernel_timer_begin(hvs_ctx* c)
{
    const size_t need = 2u * (size_t)(c->n_launch_events + 1);
    while (c->ev_k.size() < need) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) {
            c->untimed_launches++;
            return -1;
        }
        c->ev_k.push_back(e);
    }
    const int ev = c->n_launch_events;
    if (hipEventRecord(c->ev_k[2 * ev], c->stream) != hipSuccess) {
        c->untimed_launches++;
        return -1;
    }
    return ev;
}
void kernel_timer_end(hvs_ctx* c, int ev)
{
    if (ev < 0) return;
    if (hipEventRecord(c->ev_k[2 * ev + 1], c->stream) == hipSuccess)
        c->n_launch_events = ev + 1;
    else
        c->untimed_launches++;  // (main_kernel_ms is then a lower bound: bench.py refuses to price a roofline from it)
}

// the select / merge / scan kernels exist for two list capacities; `f` gets the capacity as a compile-time constant
template <typename F>
void with_cap(int cap, F f)
{
    if (cap == 512)
        f(std::integral_constant<int, 512>{});
    else
        f(std::integral_constant<int, 256>{});
}

// optimized_parallel.hpp:67: const uint32_t sn = uint32_t(sample_proportion * n);  (float product)
uint32_t sample_rows(float sample_proportion, uint32_t n)
{
    const float p = sample_proportion * (float)n;
    if (!(p > 0.0f)) return 0u;
    if (p >= 4294967296.0f) return n;
    const uint32_t sn = (uint32_t)p;
    return sn > n ? n : sn;
}

int ensure_results(hvs_ctx* c, uint32_t nq)
{
    if (nq <= c->res_cap) return HVS_OK;
    int rc;
    if ((rc = dev_alloc(c, &c->d_out_ids, (size_t)nq * c->k))) return rc;
    if ((rc = dev_alloc(c, &c->d_out_dists, (size_t)nq * c->k))) return rc;
    if ((rc = dev_alloc(c, &c->d_ovf_list, (size_t)nq))) return rc;
    if ((rc = dev_alloc(c, &c->d_retry_list, (size_t)nq))) return rc;
    c->res_cap = nq;
    return HVS_OK;
}

int ensure_queries(hvs_ctx* c, uint32_t nq)
{
    if (nq > c->nq_cap) {
        int rc;
        if ((rc = dev_alloc(c, &c->d_q, (size_t)nq * HVS_QCOLS))) return rc;
        c->nq_cap = nq;
    }
    return ensure_results(c, nq);
}

struct Plan {
    uint32_t nq_pad, qwaves, nchunks, rows_per_chunk;
};

Plan make_plan(uint32_t nqb, uint32_t sn)
{
    Plan p;
    p.qwaves = (nqb + 63u) / 64u;
    p.nq_pad = ((nqb + 255u) / 256u) * 256u;
    // enough (query-wave x row-chunk) work items to fill 256 CUs x 16 waves, but chunks of >= 2048 rows
    uint32_t want = (8192u + p.qwaves - 1u) / p.qwaves;
    uint32_t max_chunks = std::max(1u, sn / 2048u);
    // (a handful of queries -- the filter engines' fallback list -- is one wave per chunk: up to 512 chunks then, so that
    // a single query does not walk 10^7 rows with 64 waves)
    p.nchunks = std::max(1u, std::min(std::min(want, max_chunks), p.qwaves <= 2u ? 512u : 64u));
    p.rows_per_chunk = (sn + p.nchunks - 1u) / p.nchunks;
    if (p.rows_per_chunk == 0) p.rows_per_chunk = 1;
    return p;
}

int ensure_batch_workspace(hvs_ctx* c, uint32_t nqb, const Plan& p)
{
    int rc;
    if (nqb > c->batch_cap) {
        if ((rc = dev_alloc(c, &c->d_keys, (size_t)nqb))) return rc;
        if ((rc = dev_alloc(c, &c->d_keys_sorted, (size_t)nqb))) return rc;
        if ((rc = dev_alloc(c, &c->d_qidx, (size_t)nqb))) return rc;
        if ((rc = dev_alloc(c, &c->d_qorder, (size_t)nqb))) return rc;
        if ((rc = dev_alloc(c, &c->d_qra, (size_t)nqb))) return rc;
        if ((rc = dev_alloc(c, &c->d_qrb, (size_t)nqb))) return rc;
        size_t tmp = 0;
        HVS_HIP(c, rocprim::radix_sort_pairs(nullptr, tmp, c->d_keys, c->d_keys_sorted, c->d_qidx, c->d_qorder,
                                             (size_t)nqb, 0, 64, c->stream));
        if (tmp > c->sort_tmp_bytes) {
            if (c->d_sort_tmp) (void)hipFree(c->d_sort_tmp);
            c->d_sort_tmp = nullptr;
            HVS_HIP(c, hipMalloc(&c->d_sort_tmp, tmp));
            c->sort_tmp_bytes = tmp;
        }
        c->batch_cap = nqb;
    }
    const size_t lists = (size_t)p.nq_pad * p.nchunks;
    if (lists > c->cand_lists) {
        if ((rc = dev_alloc(c, &c->d_cand, lists * (size_t)c->cap))) return rc;
        if ((rc = dev_alloc(c, &c->d_cand_cnt, lists))) return rc;
        c->cand_lists = lists;
    }
    return HVS_OK;
}

// One batch of resident queries through the exact engine.  Either the contiguous range
// [q0, q0+nqb) (sorted into predicate groups first) or, when `list` is given, the nqb query
// indices stored in the device array `list` (the MFMA engine's overflow fallback).
int run_batch_exact(hvs_ctx* c, uint32_t q0, uint32_t nqb, uint32_t sn, const uint32_t* list = nullptr,
                    bool count_stats = true, bool record_events = true)
{
    const Plan p = make_plan(nqb, sn);
    int rc = ensure_batch_workspace(c, nqb, p);
    if (rc) return rc;

    const uint32_t* qorder = list;
    if (!list) {
        hipLaunchKernelGGL(hvs_k_query_keys, dim3((nqb + 255u) / 256u), dim3(256), 0, c->stream, c->d_q, q0, nqb,
                           c->d_keys, c->d_qidx);
        size_t tmp = c->sort_tmp_bytes;
        HVS_HIP(c, rocprim::radix_sort_pairs(c->d_sort_tmp, tmp, c->d_keys, c->d_keys_sorted, c->d_qidx, c->d_qorder,
                                             (size_t)nqb, 0, 64, c->stream));
        qorder = c->d_qorder;
    }
    HVS_HIP(c, hipMemsetAsync(c->d_cand_cnt, 0, (size_t)p.nq_pad * p.nchunks * sizeof(uint32_t), c->stream));

    const int ev = record_events ? kernel_timer_begin(c) : -1;
    if (sn > 0) {
        unsigned long long* stat = count_stats ? c->d_counters : c->d_counters + 4;
        const dim3 grid(p.nq_pad / 256u, p.nchunks);
        with_cap(c->cap, [&](auto CAPT) {
            constexpr int CAP = decltype(CAPT)::value;
            if (kScanRowsThroughLds) {
                if (c->scalar_order)
                    hipLaunchKernelGGL((hvs_k_scan_exact_lds<true, CAP>), grid, dim3(256), 0, c->stream, c->d_data, c->d_q, qorder, nqb,
                                       p.nq_pad, sn, p.rows_per_chunk, c->d_cand, c->d_cand_cnt, stat, c->k);
                else
                    hipLaunchKernelGGL((hvs_k_scan_exact_lds<false, CAP>), grid, dim3(256), 0, c->stream, c->d_data, c->d_q, qorder, nqb,
                                       p.nq_pad, sn, p.rows_per_chunk, c->d_cand, c->d_cand_cnt, stat, c->k);
            } else {
                if (c->scalar_order)
                    hipLaunchKernelGGL((hvs_k_scan_exact<true, CAP>), grid, dim3(256), 0, c->stream, c->d_data, c->d_q, qorder, nqb,
                                       p.nq_pad, sn, p.rows_per_chunk, c->d_cand, c->d_cand_cnt, stat, c->k);
                else
                    hipLaunchKernelGGL((hvs_k_scan_exact<false, CAP>), grid, dim3(256), 0, c->stream, c->d_data, c->d_q, qorder, nqb,
                                       p.nq_pad, sn, p.rows_per_chunk, c->d_cand, c->d_cand_cnt, stat, c->k);
            }
        });
    }
    kernel_timer_end(c, ev);
    with_cap(c->cap, [&](auto CAPT) {
        constexpr int CAP = decltype(CAPT)::value;
        if (c->scalar_order)
            hipLaunchKernelGGL((hvs_k_select<true, CAP>), dim3((nqb + 3u) / 4u), dim3(256), 0, c->stream, c->d_data, c->n, c->d_q,
                               qorder, nqb, p.nq_pad, p.nchunks, c->d_cand, c->d_cand_cnt, c->padding ? 1 : 0, c->d_out_ids, c->d_out_dists, c->k);
        else
            hipLaunchKernelGGL((hvs_k_select<false, CAP>), dim3((nqb + 3u) / 4u), dim3(256), 0, c->stream, c->d_data, c->n, c->d_q,
                               qorder, nqb, p.nq_pad, p.nchunks, c->d_cand, c->d_cand_cnt, c->padding ? 1 : 0, c->d_out_ids, c->d_out_dists, c->k);
    });
    HVS_HIP(c, hipGetLastError());
    return HVS_OK;
}

// ---------------------------------------------------------------------------------------------
// MFMA engine: index build
// ---------------------------------------------------------------------------------------------
void free_index(hvs_ctx* c)
{
    void* ptrs[] = {c->d_keys_ct, c->d_keys_t, c->d_perm_ct, c->d_perm_t, c->d_tiles_ct, c->d_tiles_t,
                    c->d_nrm_ct,  c->d_nrm_t,  c->d_bpos_ct, c->d_bpos_t};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    c->d_keys_ct = c->d_keys_t = nullptr;
    c->d_perm_ct = c->d_perm_t = nullptr;
    c->d_tiles_ct = c->d_tiles_t = nullptr;
    c->d_nrm_ct = c->d_nrm_t = nullptr;
    c->d_bpos_ct = c->d_bpos_t = nullptr;
    c->have_index = false;
    c->tile_fmt = HVS_FMT_NONE;
    c->i8_usable = false;
    c->i8_rot = false;
    c->i8_rejected = false;
    c->f16_rejected = false;
}

// Planner of HVS_ENGINE_AUTO, first guess: which tile format filters this data set more cheaply.  The INT8 filter runs
// at about twice the tile rate of the 16-bit float filters (profiles/) but its error band can be wider, and the FP16
// band is 8x tighter than the BF16 one at the same cost; a wider band inflates the number of survivors the exact kernel
// re-scores by about exp(z * 2 band / sigma): sigma = spread of squared distances between rows, z = how many sigmas
// below the mean the k-th neighbour of n rows sits.  Everything here is an estimate from a sample of rows and a
// unimodal model of the distances -- build_index checks the choice with a probe batch (probe_format) -- and it decides
// speed only: all formats give the same answers.
// HVS_I8_ROTATE: 0 = INT8 tiles never rotated, 1 = always (HVS_FMT_I8X16), unset = the planner's probe decides (read per
// data set, so that a test can switch it between contexts)
int rotate_policy()
{
    const char* v = std::getenv("HVS_I8_ROTATE");
    if (!v || !*v) return -1;
    return std::atoi(v) != 0 ? 1 : 0;
}

// centre and scale of the INT8 format over the vector components (`rot` false) or over their rotated images (HvsQuant)
int set_quant(hvs_ctx* c, bool rot)
{
    if (!c->d_quant) HVS_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_quant), sizeof(HvsQuant)));
    hipLaunchKernelGGL(hvs_k_quant_reset, dim3(1), dim3(128), 0, c->stream, c->d_quant, rot ? 1u : 0u);
    if (rot)
        hipLaunchKernelGGL(hvs_k_minmax_rot, dim3(std::min(c->n, 4096u)), dim3(128), 0, c->stream, c->d_data, c->n, c->d_quant);
    else
        hipLaunchKernelGGL(hvs_k_minmax, dim3(std::min(c->n, 4096u)), dim3(128), 0, c->stream, c->d_data, c->n, c->d_quant);
    hipLaunchKernelGGL(hvs_k_quant_params, dim3(1), dim3(128), 0, c->stream, c->d_quant);
    HVS_HIP(c, hipGetLastError());
    c->i8_rot = rot;
    return HVS_OK;
}

int choose_format(hvs_ctx* c)
{
    const uint32_t n = c->n;
    c->planned_fmt = HVS_FMT_F16;
    c->i8_usable = false;
    int rcq = set_quant(c, false);  // (the model below prices the plain INT8 format; HVS_I8_ROTATE=1 switches afterwards)
    if (rcq) return rcq;
    if (!c->d_bounds) HVS_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_bounds), sizeof(HvsBounds)));
    HVS_HIP(c, hipMemsetAsync(c->d_bounds, 0, sizeof(HvsBounds), c->stream));
    const uint32_t step = std::max(1u, n / 65536u);
    const uint32_t samples = hvs_ceil_div(n, step);
    hipLaunchKernelGGL(hvs_k_plan_stats, dim3(hvs_ceil_div(samples, 256u)), dim3(256), 0, c->stream, c->d_data, n, step,
                       c->d_quant, c->d_bounds);
    HVS_HIP(c, hipGetLastError());
    HvsBounds hb{};
    double sd = 0.0;
    HVS_HIP(c, hipMemcpyAsync(&hb, c->d_bounds, sizeof(hb), hipMemcpyDeviceToHost, c->stream));
    HVS_HIP(c, hipMemcpyAsync(&sd, reinterpret_cast<const char*>(c->d_quant) + offsetof(HvsQuant, sd), sizeof(double),
                              hipMemcpyDeviceToHost, c->stream));
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    c->i8_usable = sd > 0.0 && std::isfinite(sd);
    const bool f16_ok = std::isfinite(hb.nb_df) && std::isfinite(hb.e_df) && hb.hmax < 3.0e4f;
    if (!f16_ok) c->planned_fmt = HVS_FMT_BF16;
    if (hb.pair_n >= 16u) {
        const double mean = hb.pair_sum / hb.pair_n;
        const double var = hb.pair_sumsq / hb.pair_n - mean * mean;
        const double sigma = std::sqrt(std::max(var, 1e-300));
        // z of the k-th neighbour among n rows (normal tail, crude): k/n = Phi(-z)
        const double pq = std::min(0.4, (double)c->k / (double)n);
        const double z = std::sqrt(std::max(0.0, -2.0 * std::log(pq) - std::log(-2.0 * std::log(pq) * 6.2832)));
        // bands for a query that looks like a row (|q| ~ row norm, quantisation error ~ row error)
        const double acc16 = 4.0e-5 * ((double)hb.nb_d * (double)hb.nb_d + (double)hb.hmax);
        const double band_bf = 2.0 * (double)hb.nb_d * (double)hb.e_d + acc16;
        const double band_f = 2.0 * (double)hb.nb_df * (double)hb.e_df + acc16;
        const double band8 = 2.0 * (double)hb.n_d8 * (double)hb.e_d8;
        const double infl_bf = std::exp(std::min(50.0, z * 2.0 * band_bf / sigma));
        const double infl_f = std::exp(std::min(50.0, z * 2.0 * band_f / sigma));
        const double infl8 = std::exp(std::min(50.0, z * 2.0 * band8 / sigma));
        // cost model in units of one INT8 filter launch: a 16-bit float filter takes kPlan16 times as long, re-scoring
        // kPlanRescore times at inflation 1 and grows with the candidates (HVS_PLAN_BF16_COST / HVS_PLAN_RESCORE_COST, in
        // hundredths, override the measured defaults of profiles/)
        const double kPlan16 = env_u32("HVS_PLAN_BF16_COST", 194u, 100u, 1000u) / 100.0;
        const double kPlanRescore = env_u32("HVS_PLAN_RESCORE_COST", 15u, 1u, 1000u) / 100.0;
        const double cost_f = kPlan16 + kPlanRescore * (f16_ok ? infl_f : infl_bf), cost8 = 1.0 + kPlanRescore * infl8;
        if (c->i8_usable && cost8 < cost_f && infl8 < 6.0) c->planned_fmt = kI8Fmt;
    }
    if (const char* f = std::getenv("HVS_FILTER_FORMAT")) {  // A/B override: "bf16" / "f16" / "i8"
        if (!std::strcmp(f, "bf16")) c->planned_fmt = HVS_FMT_BF16;
        if (!std::strcmp(f, "f16")) c->planned_fmt = HVS_FMT_F16;
        if (!std::strcmp(f, "i8")) c->planned_fmt = kI8Fmt;
    }
    if (rotate_policy() == 1 && kI8Fmt == HVS_FMT_I8X16 && c->i8_usable) {
        if ((rcq = set_quant(c, true))) return rcq;
        double sdr = 0.0;
        HVS_HIP(c, hipMemcpyAsync(&sdr, reinterpret_cast<const char*>(c->d_quant) + offsetof(HvsQuant, sd), sizeof(double),
                                  hipMemcpyDeviceToHost, c->stream));
        HVS_HIP(c, hipStreamSynchronize(c->stream));
        if (!(sdr > 0.0 && std::isfinite(sdr)) && (rcq = set_quant(c, false))) return rcq;
    }
    return HVS_OK;
}

// (re)build the level-interleaved tiles of both orderings in format `fmt`; the orderings must exist
int build_tiles(hvs_ctx* c, int fmt)
{
    const HvsLevels L = c->lv;
    const uint32_t n = c->n;
    int rc;
    c->tile_fmt = HVS_FMT_NONE;
    c->have_index = false;
    // free first: the two formats never coexist (D = 1e8: 44.8 GB of BF16 tiles, 20.8 GB of INT8 tiles)
    if ((rc = dev_alloc(c, &c->d_tiles_ct, (size_t)0))) return rc;
    if ((rc = dev_alloc(c, &c->d_tiles_t, (size_t)0))) return rc;
    if ((rc = dev_alloc(c, &c->d_nrm_ct, (size_t)0))) return rc;
    if ((rc = dev_alloc(c, &c->d_nrm_t, (size_t)0))) return rc;
    const size_t tile_u4 = fmt == HVS_FMT_I8 ? HVS_I8_TILE_U4 : fmt == HVS_FMT_I8X16 ? HVS_I8X16_TILE_U4 : HVS_TILE_U4;
    if ((rc = dev_alloc(c, &c->d_tiles_ct, (size_t)L.nblk * tile_u4))) return rc;
    if ((rc = dev_alloc(c, &c->d_tiles_t, (size_t)L.nblk * tile_u4))) return rc;
    if (HVS_IS_I8(fmt)) {
        const size_t nrm_u4 = fmt == HVS_FMT_I8 ? HVS_I8_NRM_U4 : HVS_I8X16_NRM_U4;
        if ((rc = dev_alloc(c, &c->d_nrm_ct, (size_t)L.nblk * nrm_u4))) return rc;
        if ((rc = dev_alloc(c, &c->d_nrm_t, (size_t)L.nblk * nrm_u4))) return rc;
    }
    if (!c->d_bpos_ct && (rc = dev_alloc(c, &c->d_bpos_ct, (size_t)L.nblk))) return rc;
    if (!c->d_bpos_t && (rc = dev_alloc(c, &c->d_bpos_t, (size_t)L.nblk))) return rc;
    HVS_HIP(c, hipMemsetAsync(c->d_bounds, 0, sizeof(HvsBounds), c->stream));
    const dim3 grid((L.nblk + 3u) / 4u);
    if (fmt == HVS_FMT_I8X16 && c->i8_rot) {
        hipLaunchKernelGGL(hvs_k_build_tiles_i8x16_rot, grid, dim3(256), 0, c->stream, c->d_data, n, c->d_perm_ct, L, c->d_quant,
                           c->d_tiles_ct, reinterpret_cast<int*>(c->d_nrm_ct), c->d_bpos_ct, c->d_bounds);
        hipLaunchKernelGGL(hvs_k_build_tiles_i8x16_rot, grid, dim3(256), 0, c->stream, c->d_data, n, c->d_perm_t, L, c->d_quant,
                           c->d_tiles_t, reinterpret_cast<int*>(c->d_nrm_t), c->d_bpos_t, c->d_bounds);
    } else if (fmt == HVS_FMT_I8X16) {
        hipLaunchKernelGGL(hvs_k_build_tiles_i8x16, grid, dim3(256), 0, c->stream, c->d_data, n, c->d_perm_ct, L, c->d_quant,
                           c->d_tiles_ct, reinterpret_cast<int*>(c->d_nrm_ct), c->d_bpos_ct, c->d_bounds);
        hipLaunchKernelGGL(hvs_k_build_tiles_i8x16, grid, dim3(256), 0, c->stream, c->d_data, n, c->d_perm_t, L, c->d_quant,
                           c->d_tiles_t, reinterpret_cast<int*>(c->d_nrm_t), c->d_bpos_t, c->d_bounds);
    } else if (fmt == HVS_FMT_I8) {
        hipLaunchKernelGGL(hvs_k_build_tiles_i8, grid, dim3(256), 0, c->stream, c->d_data, n, c->d_perm_ct, L, c->d_quant,
                           c->d_tiles_ct, reinterpret_cast<int*>(c->d_nrm_ct), c->d_bpos_ct, c->d_bounds);
        hipLaunchKernelGGL(hvs_k_build_tiles_i8, grid, dim3(256), 0, c->stream, c->d_data, n, c->d_perm_t, L, c->d_quant,
                           c->d_tiles_t, reinterpret_cast<int*>(c->d_nrm_t), c->d_bpos_t, c->d_bounds);
    } else {
        hipLaunchKernelGGL(hvs_k_build_tiles, grid, dim3(256), 0, c->stream, c->d_data, n, c->d_perm_ct, L, c->d_tiles_ct,
                           c->d_bpos_ct, c->d_bounds, fmt);
        hipLaunchKernelGGL(hvs_k_build_tiles, grid, dim3(256), 0, c->stream, c->d_data, n, c->d_perm_t, L, c->d_tiles_t,
                           c->d_bpos_t, c->d_bounds, fmt);
    }
    HVS_HIP(c, hipGetLastError());
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    // the error bound needs finite row norms: data with inf/NaN components (or |d|^2 overflowing f32)
    // is answered by the exact engine only
    HvsBounds hb{};
    HVS_HIP(c, hipMemcpy(&hb, c->d_bounds, sizeof(hb), hipMemcpyDeviceToHost));
    bool ok;
    if (HVS_IS_I8(fmt))
        ok = std::isfinite(hb.e_d8) && std::isfinite(hb.n_d8) && hb.n_d8 < 1.0e15f;
    else if (fmt == HVS_FMT_F16)
        // every component and the three pieces of -|d|^2/2 inside the half-precision range (denormal flushing is part of
        // the bound: HVS_F16_FLUSH)
        ok = std::isfinite(hb.e_d) && std::isfinite(hb.nb_d) && std::isfinite(hb.hmax) && std::isfinite(hb.rho) && hb.hmax < 6.0e4f;
    else
        // (norms near the f32 denormal range: BF16 operands might be flushed by the matrix pipe, which the
        // error bound does not model)
        ok = std::isfinite(hb.e_d) && std::isfinite(hb.nb_d) && std::isfinite(hb.hmax) && std::isfinite(hb.rho) &&
             !(hb.hmax > 1.0e30f) && !(hb.hmax > 0.0f && hb.hmax < 1.0e-20f);
    if (kTrace)
        std::fprintf(stderr, "[hvs trace] tiles built: format %d usable %d e_d8 %.6g n_d8 %.6g e_d %.6g nb_d %.6g hmax %.6g rho %.6g\n", fmt, (int)ok,
                     (double)hb.e_d8, (double)hb.n_d8, (double)hb.e_d, (double)hb.nb_d, (double)hb.hmax, (double)hb.rho);
    if (!ok) return HVS_OK;  // have_index stays false
    c->tile_fmt = fmt;
    c->have_index = true;
    c->i8_rot_built = HVS_IS_I8(fmt) && c->i8_rot;
    return HVS_OK;
}

int run_batch_mfma(hvs_ctx* c, uint32_t q0, uint32_t nqb, uint32_t sn, const uint32_t* list, bool proven_last);
HvsGuessTable plan_guess(uint32_t k, bool proven, uint32_t pfail);

// the tile format the context's engine setting asks for (HVS_FMT_NONE: HVS_ENGINE_AUTO found no filter worth running)
int want_format(const hvs_ctx* c)
{
    int want;
    switch (c->engine) {
        case HVS_ENGINE_MFMA_FILTER: want = HVS_FMT_BF16; break;
        case HVS_ENGINE_MFMA_F16: want = HVS_FMT_F16; break;
        case HVS_ENGINE_MFMA_I8: want = c->i8_usable ? kI8Fmt : HVS_FMT_F16; break;
        default: want = c->planned_fmt; break;
    }
    if (HVS_IS_I8(want) && c->i8_rejected) want = HVS_FMT_F16;
    if (want == HVS_FMT_F16 && c->f16_rejected) want = HVS_FMT_BF16;
    return want;
}

// build `want`, or the next format down the chain INT8 -> FP16 -> BF16 whose bound is usable on this data set
// (have_index stays false when none is)
int build_tiles_chain(hvs_ctx* c, int want)
{
    for (;;) {
        int rc = build_tiles(c, want);
        if (rc) return rc;
        if (c->have_index) return HVS_OK;
        if (HVS_IS_I8(want)) {
            c->i8_rejected = true;
            want = c->f16_rejected ? HVS_FMT_BF16 : HVS_FMT_F16;
        } else if (want == HVS_FMT_F16) {
            c->f16_rejected = true;
            want = HVS_FMT_BF16;
        } else {
            return HVS_OK;
        }
    }
}

// Planner of HVS_ENGINE_AUTO, second step: measure instead of model.  A probe batch of 1024 rows of D used as type-0
// queries (only D is used, cf. README.md:68) runs through the filter engine in the format just built; what it hands to the
// exact kernel and what it cannot answer is priced in units of the INT8 filter's own time per query:
//     cost = F + (2650 / n) re-scored pairs per query + 3 retried fraction + 142 exact-engine fraction
// (F = 1 for INT8 tiles, 1.94 for 16-bit float tiles; re-scoring 0.07 ns per pair against 0.0266 ns per row and query of the
// INT8 filter, the exact engine 140x the filter -- profiles/r03), next to its INFLATION = re-scored pairs / what a filter
// without an error band would hand over under the same guessed thresholds, and the fraction of queries it left unanswered.  The model of choose_format assumes one bell-shaped
// distance distribution; clustered data and data with a few dominant dimensions have neighbour distances far below what
// it expects, and an 8-bit grid over the bounding box then lets thousands of rows per query through.
int probe_format(hvs_ctx* c, double* cost, double* inflation, double* failed)
{
    constexpr uint32_t P = 1024;
    *cost = 1.0e9;
    // the probe has its own query / result / re-run buffers: resident queries and results of the caller stay untouched
    float *pq = nullptr, *pdist = nullptr;
    uint32_t *pids = nullptr, *povf = nullptr, *pretry = nullptr;
    auto release = [&]() {
        void* bufs[] = {pq, pdist, pids, povf, pretry};
        for (void* b : bufs)
            if (b) (void)hipFree(b);
    };
    if (hipMalloc(reinterpret_cast<void**>(&pq), (size_t)P * HVS_QCOLS * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&pdist), (size_t)P * c->k * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&pids), (size_t)P * c->k * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&povf), (size_t)P * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&pretry), (size_t)P * sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        release();
        return fail(c, HVS_ENOMEM, "planner probe: out of device memory");
    }
    std::swap(c->d_q, pq);
    std::swap(c->d_out_ids, pids);
    std::swap(c->d_out_dists, pdist);
    std::swap(c->d_ovf_list, povf);
    std::swap(c->d_retry_list, pretry);
    const uint32_t step = std::max(1u, c->n / P);
    hipLaunchKernelGGL(hvs_k_probe_queries, dim3(hvs_ceil_div(P * HVS_QCOLS, 256u)), dim3(256), 0, c->stream, c->d_data, c->n, step, P, c->d_q);
    hipError_t e = hipMemsetAsync(c->d_counters, 0, 16 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_ovf_count, 0, 2 * sizeof(uint32_t), c->stream);
    // (the probe batch is small, but it stands for full batches: their failure target.  Its lists keep the production capacity:
    // PCA-like vectors at n = 10^7 hand ~1500 rows per type-0 query to the last level's list of 1024 -- INT8 tiles look 23 %
    // faster than FP16 tiles there only while a half-empty workspace doubles the lists, profiles/r04/nonuniform_int8.txt)
    c->force_pfail = guess_pfail_for(kBatchMfma);
    int rc = e == hipSuccess ? run_batch_mfma(c, 0, P, c->n, nullptr, false) : fail(c, HVS_EHIP, "planner probe: memset failed");
    c->force_pfail = 0u;
    unsigned long long h[4] = {0, 0, 0, 0};
    uint32_t fails[2] = {0, 0};
    if (!rc) {
        e = hipMemcpyAsync(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(fails, c->d_ovf_count, sizeof(fails), hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) rc = fail(c, HVS_EHIP, "planner probe: copy failed");
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail(c, HVS_EHIP, "planner probe: synchronisation failed");
    std::swap(c->d_q, pq);
    std::swap(c->d_out_ids, pids);
    std::swap(c->d_out_dists, pdist);
    std::swap(c->d_ovf_list, povf);
    std::swap(c->d_retry_list, pretry);
    release();
    c->n_launch_events = 0;
    if (rc) return rc;
    const double base = HVS_IS_I8(c->tile_fmt) ? 1.0 : env_u32("HVS_PLAN_BF16_COST", 194u, 100u, 1000u) / 100.0;
    *cost = base + 2650.0 / (double)c->n * ((double)h[2] / P) + 3.0 * fails[1] / P + 142.0 * fails[0] / P;
    // what a filter without any error band would have handed over: m (radix - 1) rows per level under the guessed thresholds
    double ideal = 0.0;
    {
        const HvsGuessTable G = plan_guess(c->k, false, guess_pfail_for(kBatchMfma));
        double seen = 1.0;  // fraction of the rows seen, from the last level backwards
        for (uint32_t j = c->lv.K; j >= 1u; --j) {
            seen /= (double)c->lv.radix[j];
            int idx = (int)std::ceil(8.0 * -std::log2(seen) - 0.02);
            idx = std::max(0, std::min(HVS_GUESS_STEPS - 1, idx));
            const double m = G.last_m && j == c->lv.K ? (double)G.last_m : (double)std::max(G.m[idx], G.floor_m);
            ideal += std::min(m, (double)c->k) * ((double)c->lv.radix[j] - 1.0);
        }
    }
    *inflation = ideal > 0.0 ? ((double)h[2] / P) / ideal : 1.0;
    *failed = (double)(fails[0] + fails[1]) / P;
    return HVS_OK;
}

// after build_tiles_chain in HVS_ENGINE_AUTO: keep the format the model chose if the probe agrees, else try the next one up
// the precision chain; planned_fmt = HVS_FMT_NONE when no filter beats the exact engine
int plan_by_probe(hvs_ctx* c)
{
    if (!env_u32("HVS_PLAN_PROBE", 1u, 0u, 1u) || std::getenv("HVS_FILTER_FORMAT")) return HVS_OK;
    double cost = 0.0, infl = 0.0, failed = 0.0;
    int rc = probe_format(c, &cost, &infl, &failed);
    if (rc) return rc;
    if (kTrace) std::fprintf(stderr, "[hvs trace] planner probe: format %d cost %.3f inflation %.2f failed %.4f\n", c->tile_fmt, cost, infl, failed);
    int best = c->tile_fmt;
    bool best_rot = c->i8_rot;
    double best_cost = cost;
    // INT8 tiles stay unless their band visibly lets too much through on this data (the cost formula is not trusted to
    // split hairs between formats that both work: at small n everything is launch latency).  Otherwise the candidates are
    // probed in turn -- the rotated INT8 tiles (same filter rate; HvsQuant), then the FP16 tiles (half the rate, an 8x tighter
    // band) -- and the cheapest stays.
    if (HVS_IS_I8(c->tile_fmt) && (infl > 2.5 || failed > 0.01)) {
        const int had = c->tile_fmt;
        const bool had_rot = c->i8_rot;
        if (had == HVS_FMT_I8X16 && !had_rot && rotate_policy() != 0) {
            if ((rc = set_quant(c, true))) return rc;
            if ((rc = build_tiles(c, had))) return rc;
            if (c->have_index) {
                if ((rc = probe_format(c, &cost, &infl, &failed))) return rc;
                if (kTrace) std::fprintf(stderr, "[hvs trace] planner probe: format %d (rotated) cost %.3f inflation %.2f failed %.4f\n", c->tile_fmt, cost, infl, failed);
                if (cost < best_cost) {
                    best_rot = true;
                    best_cost = cost;
                }
            }
        }
        if (!c->f16_rejected) {
            if ((rc = build_tiles_chain(c, HVS_FMT_F16))) return rc;
            if (c->have_index && (rc = probe_format(c, &cost, &infl, &failed))) return rc;
            if (kTrace) std::fprintf(stderr, "[hvs trace] planner probe: format %d cost %.3f inflation %.2f failed %.4f\n", c->tile_fmt, cost, infl, failed);
            if (c->have_index && cost < best_cost) {
                best = c->tile_fmt;
                best_cost = cost;
            }
        }
        if (HVS_IS_I8(best)) {  // an INT8 variant stays: its centre / scale and tiles come back if something else was built last
            if (c->i8_rot != best_rot && (rc = set_quant(c, best_rot))) return rc;
            if ((c->tile_fmt != best || !c->have_index || c->i8_rot_built != best_rot) && (rc = build_tiles_chain(c, had))) return rc;
        } else if (c->i8_rot && (rc = set_quant(c, false))) {  // (a later change of engine finds the plain INT8 parameters)
            return rc;
        }
    }
    if (best_cost >= 140.0) best = HVS_FMT_NONE;  // no filter beats the exact engine's range scans here
    c->planned_fmt = best;
    return HVS_OK;
}

int build_index(hvs_ctx* c)
{
    free_index(c);
    const uint32_t n = c->n;
    const HvsLevels L = hvs_make_levels(n, kRadixLast, kRadixMid, kRadixPlan.set ? kRadixPlan.r : nullptr);
    if (L.off[L.K + 1] != L.nblk) return fail(c, HVS_EINVAL, "internal: level table does not cover the blocks");
    // survivor entries carry the block position in 24 bits: above 2^29 rows per GPU the exact engine answers
    // (2^29 rows are 219 GB of rows alone: such a data set is sharded over GPUs anyway)
    c->index_too_large = L.nblk > HVS_ENTRY_MAX_BLOCKS;
    if (c->index_too_large) return HVS_OK;
    c->lv = L;
    int rc;
    uint64_t *k_ct = nullptr, *k_t = nullptr;
    uint32_t* ids = nullptr;
    void* tmp = nullptr;
    auto cleanup = [&]() {
        if (k_ct) (void)hipFree(k_ct);
        if (k_t) (void)hipFree(k_t);
        if (ids) (void)hipFree(ids);
        if (tmp) (void)hipFree(tmp);
    };
#define HVS_TRY(expr)            \
    do {                         \
        if ((rc = (expr))) {     \
            cleanup();           \
            free_index(c);       \
            return rc;           \
        }                        \
    } while (0)
    HVS_TRY(dev_alloc(c, &k_ct, (size_t)n));
    HVS_TRY(dev_alloc(c, &k_t, (size_t)n));
    HVS_TRY(dev_alloc(c, &ids, (size_t)n));
    HVS_TRY(dev_alloc(c, &c->d_keys_ct, (size_t)n));
    HVS_TRY(dev_alloc(c, &c->d_keys_t, (size_t)n));
    HVS_TRY(dev_alloc(c, &c->d_perm_ct, (size_t)n));
    HVS_TRY(dev_alloc(c, &c->d_perm_t, (size_t)n));
    hipLaunchKernelGGL(hvs_k_attr_keys, dim3((n + 255u) / 256u), dim3(256), 0, c->stream, c->d_data, n, k_ct, k_t, ids);
    size_t tmp_bytes = 0, tmp_bytes_t = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_ct, c->d_keys_ct, ids, c->d_perm_ct, (size_t)n, 0, 64,
                                             c->stream);
    if (e == hipSuccess)
        e = rocprim::radix_sort_pairs(nullptr, tmp_bytes_t, k_t, c->d_keys_t, ids, c->d_perm_t, (size_t)n, 0, 64, c->stream);
    if (tmp_bytes_t > tmp_bytes) tmp_bytes = tmp_bytes_t;  // each call sizes its own algorithm
    if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes);
    if (e == hipSuccess)
        e = rocprim::radix_sort_pairs(tmp, tmp_bytes, k_ct, c->d_keys_ct, ids, c->d_perm_ct, (size_t)n, 0, 64, c->stream);
    if (e == hipSuccess)
        e = rocprim::radix_sort_pairs(tmp, tmp_bytes, k_t, c->d_keys_t, ids, c->d_perm_t, (size_t)n, 0, 64, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        cleanup();
        free_index(c);
        return fail(c, e == hipErrorOutOfMemory ? HVS_ENOMEM : HVS_EHIP, std::string("index sort: ") + hipGetErrorString(e));
    }
    cleanup();
    k_ct = k_t = nullptr;
    ids = nullptr;
    tmp = nullptr;
#undef HVS_TRY
    trace_mark(c, "orderings");
    if ((rc = choose_format(c))) {
        free_index(c);
        return rc;
    }
    trace_mark(c, "format");
    const int fmt = want_format(c);
    if (fmt != HVS_FMT_NONE && (rc = build_tiles_chain(c, fmt))) {
        free_index(c);
        return rc;
    }
    trace_mark(c, "tiles");
    if (fmt != HVS_FMT_NONE && !c->have_index) free_index(c);  // no format has a usable bound: exact engine only
    if (fmt == HVS_FMT_NONE) c->have_index = true;              // (orderings only: the exact engine's range scans)
    if (c->have_index && c->tile_fmt != HVS_FMT_NONE && c->engine == HVS_ENGINE_AUTO && n >= kMfmaMinRows) {
        if ((rc = plan_by_probe(c))) {
            free_index(c);
            return rc;
        }
        trace_mark(c, "probe");
    }
    return HVS_OK;
}

// ---------------------------------------------------------------------------------------------
// MFMA engine: one batch
// ---------------------------------------------------------------------------------------------
// `want_fcap`: candidate keys per slot and round the batch needs (HVS_FCAP for ordinary batches; retry batches, which run
// every level with the proven threshold, ask for more); the lists take whatever the workspace offers beyond that
int ensure_filter_workspace(hvs_ctx* c, uint32_t nqb, uint32_t want_fcap = HVS_FCAP)
{
    const uint32_t slots = hvs_ceil_div(nqb + 5u * 32u + (HVS_WG_WAVES + 1u) * HVS_GROUP, HVS_GROUP) * HVS_GROUP;
    const uint32_t groups = slots / HVS_GROUP;
    if (slots > HVS_ENTRY_MAX_SLOTS) return fail(c, HVS_EINVAL, "internal: batch too large for the survivor entries' slot field");
    HvsBatch& B = c->fb;
    int rc;
    if (slots > c->fb_slots_cap) {
        c->fb_slots_cap = 0;  // (a failed allocation below leaves no half-sized workspace behind)
#define HVS_A(field, count) \
    if ((rc = dev_alloc(c, &B.field, (size_t)(count)))) return rc
        HVS_A(qid, slots);
        HVS_A(rank, slots);
        HVS_A(ra, slots);
        HVS_A(rb, slots);
        HVS_A(gua, groups);
        HVS_A(gub, groups);
        HVS_A(gord, groups);
        HVS_A(bfrag, (size_t)(slots / 32u) * HVS_TILE_U4);
        HVS_A(theta, slots);
        B.thetai = reinterpret_cast<int*>(B.theta);
        HVS_A(qn, slots);
        HVS_A(normq, slots);
        HVS_A(eq, slots);
        HVS_A(nqb, slots);
        HVS_A(top, (size_t)slots * 256u);  // stride 128 (k <= 128) or 256
        HVS_A(topcnt, slots);
        HVS_A(tau, slots);
        HVS_A(candcnt, slots);
        HVS_A(overflow, slots);
        HVS_A(paircnt, groups);
        HVS_A(goverflow, groups);
#undef HVS_A
        if (!c->d_layout) HVS_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_layout), 16 * sizeof(uint32_t)));
        c->fb_slots_cap = slots;
    }
    // candidate lists and survivor-entry lists: sized in entries, shared out over the slots / groups of the batch
    const size_t need_cand = (size_t)slots * want_fcap, need_pairs = (size_t)groups * HVS_GROUP * want_fcap;
    if (need_cand > c->fb_cand_entries) {
        c->fb_cand_entries = 0;  // (dev_alloc releases the old buffer first: nothing is held if it fails)
        if ((rc = dev_alloc(c, &B.cand, need_cand))) return rc;
        c->fb_cand_entries = need_cand;
    }
    if (need_pairs > c->fb_pair_entries) {
        c->fb_pair_entries = 0;
        if ((rc = dev_alloc(c, &B.pairs, need_pairs))) return rc;
        c->fb_pair_entries = need_pairs;
    }
    B.nslots = slots;
    B.ngroups = groups;
    B.fcap = (uint32_t)std::min<size_t>(16384u, c->fb_cand_entries / slots);
    B.gcap = (uint32_t)std::min<size_t>((size_t)HVS_GROUP * 16384u, c->fb_pair_entries / groups);
    if (c->force_pfail) {  // planner probe: a full batch's list capacity, whatever a larger workspace would offer this small batch
        B.fcap = std::min<uint32_t>(B.fcap, HVS_FCAP);
        B.gcap = std::min<uint32_t>(B.gcap, HVS_GCAP);
    }
    B.knn = c->k;
    B.topcap = c->k <= 128u ? 128u : 256u;
    // sort workspace shared with the exact engine
    Plan p{};
    p.nq_pad = 256;
    p.nchunks = 1;
    return ensure_batch_workspace(c, nqb, p);
}

// slot layout, position ranges, norms and B fragments of one batch (shared by the MFMA engine and the
// range-based exact engine)
int prep_batch(hvs_ctx* c, uint32_t q0, uint32_t nqb, bool count_pairs, int fmt, bool host_counts = false,
               const uint32_t* list = nullptr, uint32_t want_fcap = HVS_FCAP)
{
    int rc = ensure_filter_workspace(c, nqb, want_fcap);
    if (rc) return rc;
    HvsBatch& B = c->fb;
    const uint32_t n = c->n;
    // ~4096 queries of a predicate class per start-position bin (32 groups); inside a bin queries are
    // ordered by range end, so the 4 groups of a filter workgroup stream nearly the same run of tiles.  The class
    // populations stay on the device (hvs_k_query_keys2 reads them there); only the range-scan exact engine, whose
    // launch shapes depend on them, waits for a copy (`host_counts`).
    HVS_HIP(c, hipMemsetAsync(c->d_layout + 8, 0, 8 * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(hvs_k_count_classes, dim3((nqb + 255u) / 256u), dim3(256), 0, c->stream, c->d_q, q0, nqb, list,
                       c->d_layout + 8);
    if (host_counts) {
        uint32_t counts[5] = {0, 0, 0, 0, 0};
        HVS_HIP(c, hipMemcpyAsync(counts, c->d_layout + 8, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
        HVS_HIP(c, hipStreamSynchronize(c->stream));
        for (int k = 0; k < 5; ++k) c->class_counts[k] = counts[k];
    }
    hipLaunchKernelGGL(hvs_k_query_keys2, dim3((nqb + 255u) / 256u), dim3(256), 0, c->stream, c->d_q, q0, nqb, list, c->d_keys_ct,
                       c->d_keys_t, n, c->d_layout + 8, c->d_keys, c->d_qidx, c->d_qra, c->d_qrb);
    size_t tmp = c->sort_tmp_bytes;
    HVS_HIP(c, rocprim::radix_sort_pairs(c->d_sort_tmp, tmp, c->d_keys, c->d_keys_sorted, c->d_qidx, c->d_qorder,
                                         (size_t)nqb, 0, 64, c->stream));
    hipLaunchKernelGGL(hvs_k_layout, dim3(std::min(hvs_ceil_div(B.nslots, 1024u), 1024u)), dim3(1024), 0, c->stream, c->d_keys_sorted, c->d_qorder, nqb, B.nslots, q0, list,
                       c->d_qra, c->d_qrb, B.qid, B.rank, B.ra, B.rb, c->d_layout);
    // (last argument: the batch is for a filter engine -- `host_counts` is the exact engine's range scan, which answers every
    // query itself, non-finite ones included)
    hipLaunchKernelGGL(hvs_k_prep, dim3(B.ngroups), dim3(4 * HVS_GROUP), 0, c->stream, c->d_q, B, count_pairs ? 1 : 0, c->d_counters,
                       fmt, c->d_quant, c->d_bounds, host_counts ? 0 : 1);
    HVS_HIP(c, hipGetLastError());
    return HVS_OK;
}

// Exact engine on top of the index.  Queries with a categorical predicate (types 1 and 3) scan only
// their position range of the (C,T) ordering (hvs_k_scan_ranges): ~1 % / 0.25 % of the rows, 17-25x
// faster than scanning everything.  Type-0 and type-2 queries keep the sequential full scan in original
// row order (hvs_k_scan_exact streams D once through the scalar cache): a position range is a random
// permutation of the rows, and measured on D=1e7 gathering 25 % of them (type 2) is slower than
// streaming all of them (8.2 k vs 14 k queries/s).
int run_batch_exact_ranges(hvs_ctx* c, uint32_t q0, uint32_t nqb, uint32_t sn)
{
    int rc = prep_batch(c, q0, nqb, sn == c->n, HVS_FMT_BF16, true);  // (the range scan uses the ranges only)
    if (rc) return rc;
    HvsBatch& B = c->fb;
    if (sn != c->n)
        hipLaunchKernelGGL(hvs_k_count_prefix_pairs, dim3(B.nslots), dim3(64), 0, c->stream, B, c->d_perm_ct, c->d_perm_t, sn,
                           c->d_counters);
    // slot layout of hvs_k_layout: classes 0..3 padded to 32 slots each, then the T-ordering class (type 2)
    // from the next filter-workgroup boundary
    const uint32_t nq0 = c->class_counts[0], nq2 = c->class_counts[4];
    const uint32_t slot_begin = hvs_ceil_div(nq0, 32u) * 32u;
    uint32_t slot_end = slot_begin;
    for (int k = 1; k < 4; ++k) slot_end += hvs_ceil_div(c->class_counts[k], 32u) * 32u;
    const uint32_t slot2 = hvs_ceil_div(slot_end, HVS_WG_WAVES * HVS_GROUP) * (HVS_WG_WAVES * HVS_GROUP);
    if (nq0 && (rc = run_batch_exact(c, 0, nq0, sn, B.qid, false))) return rc;
    if (nq2 && (rc = run_batch_exact(c, 0, nq2, sn, B.qid + slot2, false))) return rc;
    if (slot_begin >= slot_end) return HVS_OK;
    const uint32_t waves = (slot_end - slot_begin + 63u) / 64u;
    uint32_t nchunks = std::max(1u, std::min(64u, (8192u + waves - 1u) / waves));
    nchunks = std::min(nchunks, std::max(1u, (1u << 20) / B.nslots));  // candidate lists: at most 2 GB
    const size_t lists = (size_t)B.nslots * nchunks;
    if (lists > c->cand_lists) {
        if ((rc = dev_alloc(c, &c->d_cand, lists * (size_t)c->cap))) return rc;
        if ((rc = dev_alloc(c, &c->d_cand_cnt, lists))) return rc;
        c->cand_lists = lists;
    }
    HVS_HIP(c, hipMemsetAsync(c->d_cand_cnt, 0, lists * sizeof(uint32_t), c->stream));
    const int ev = kernel_timer_begin(c);
    const dim3 grid((slot_end + 255u) / 256u, nchunks);
    with_cap(c->cap, [&](auto CAPT) {
        constexpr int CAP = decltype(CAPT)::value;
        if (c->scalar_order)
            hipLaunchKernelGGL((hvs_k_scan_ranges<true, CAP>), grid, dim3(256), 0, c->stream, c->d_data, sn, c->d_q, B, c->d_perm_ct,
                               c->d_perm_t, nchunks, slot_begin, slot_end, c->d_cand, c->d_cand_cnt, c->d_counters);
        else
            hipLaunchKernelGGL((hvs_k_scan_ranges<false, CAP>), grid, dim3(256), 0, c->stream, c->d_data, sn, c->d_q, B, c->d_perm_ct,
                               c->d_perm_t, nchunks, slot_begin, slot_end, c->d_cand, c->d_cand_cnt, c->d_counters);
    });
    kernel_timer_end(c, ev);
    const uint32_t nsel = slot_end - slot_begin;
    uint64_t* cand = c->d_cand + (size_t)slot_begin * (size_t)c->cap;
    uint32_t* cnt = c->d_cand_cnt + slot_begin;
    with_cap(c->cap, [&](auto CAPT) {
        constexpr int CAP = decltype(CAPT)::value;
        if (c->scalar_order)
            hipLaunchKernelGGL((hvs_k_select<true, CAP>), dim3((nsel + 3u) / 4u), dim3(256), 0, c->stream, c->d_data, c->n, c->d_q,
                               B.qid + slot_begin, nsel, B.nslots, nchunks, cand, cnt, c->padding ? 1 : 0, c->d_out_ids, c->d_out_dists, c->k);
        else
            hipLaunchKernelGGL((hvs_k_select<false, CAP>), dim3((nsel + 3u) / 4u), dim3(256), 0, c->stream, c->d_data, c->n, c->d_q,
                               B.qid + slot_begin, nsel, B.nslots, nchunks, cand, cnt, c->padding ? 1 : 0, c->d_out_ids, c->d_out_dists, c->k);
    });
    HVS_HIP(c, hipGetLastError());
    return HVS_OK;
}

// work-item lists of the batch just prepared (all levels at once; see HvsItems in hvs_filter.h)
// buffers of the work-item lists for the batch whose slot layout ensure_filter_workspace has just set (c->fb.ngroups)
int ensure_items(hvs_ctx* c)
{
    const HvsBatch& B = c->fb;
    const HvsLevels L = c->lv;
    const uint32_t nquads = hvs_ceil_div(B.ngroups, HVS_WG_WAVES);
    static const uint32_t kItemsPerSlot = env_u32("HVS_SEG_ITEMS", 8u, 1u, 64u);  // work items per resident workgroup a level should make
    const HvsSegs S = hvs_make_segs(L, nquads, 2u * (uint32_t)c->num_cus, kItemsPerSlot);
    c->segs = S;
    const uint32_t nseg = S.first[L.K + 1];
    if (nquads > (1u << HVS_ITEM_QUAD_BITS)) return fail(c, HVS_EINVAL, "internal: too many query quads for the item code");
    int rc;
    if (nquads > c->quads_cap) {
        c->quads_cap = 0;
        if ((rc = dev_alloc(c, &c->d_qlo, (size_t)nquads))) return rc;
        if ((rc = dev_alloc(c, &c->d_qhi, (size_t)nquads))) return rc;
        c->quads_cap = nquads;
    }
    if (nseg + 1u > c->segs_cap) {
        c->segs_cap = 0;
        if ((rc = dev_alloc(c, &c->d_segcnt, (size_t)nseg + 1u))) return rc;
        if ((rc = dev_alloc(c, &c->d_segoff, (size_t)nseg + 1u))) return rc;
        c->segs_cap = nseg + 1u;
    }
    if (!c->d_lvloff && (rc = dev_alloc(c, &c->d_lvloff, (size_t)32))) return rc;
    if (!c->d_cursor && (rc = dev_alloc(c, &c->d_cursor, (size_t)16))) return rc;
    const size_t worst = (size_t)nquads * (size_t)(nseg - S.first[1]);  // every quad meets every segment (type 0)
    if (worst > c->items_cap) {
        c->items_cap = 0;
        if ((rc = dev_alloc(c, &c->d_items, worst))) return rc;
        c->items_cap = worst;
    }
    return HVS_OK;
}

int build_items(hvs_ctx* c)
{
    int rc = ensure_items(c);
    if (rc) return rc;
    const HvsBatch& B = c->fb;
    const HvsLevels L = c->lv;
    const uint32_t nquads = hvs_ceil_div(B.ngroups, HVS_WG_WAVES);
    const HvsSegs S = c->segs;
    const uint32_t nseg = S.first[L.K + 1];
    HVS_HIP(c, hipMemsetAsync(c->d_cursor, 0, 16 * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(hvs_k_quad_ranges, dim3(hvs_ceil_div(nquads, 256u)), dim3(256), 0, c->stream, B, nquads, c->d_qlo, c->d_qhi);
    hipLaunchKernelGGL(hvs_k_item_sweep<false>, dim3(nseg), dim3(256), 0, c->stream, L, S, nquads, c->d_qlo, c->d_qhi, c->d_segcnt,
                       c->d_segoff, c->d_items);
    hipLaunchKernelGGL(hvs_k_item_scan, dim3(1), dim3(1024), 0, c->stream, S, c->d_segcnt, c->d_segoff, c->d_lvloff);
    hipLaunchKernelGGL(hvs_k_item_sweep<true>, dim3(nseg), dim3(256), 0, c->stream, L, S, nquads, c->d_qlo, c->d_qhi, c->d_segcnt,
                       c->d_segoff, c->d_items);
    HVS_HIP(c, hipGetLastError());
    return HVS_OK;
}

// Order statistics of the guessed thresholds (see "Guessed thresholds" at hvs_k_merge).  If the rows seen so far were a
// random sample holding a fraction F of the query's rows, the number X of unseen rows below the sample's m-th smallest
// distance would be negative binomial, P(X = x) = C(x + m - 1, x) F^m (1 - F)^x, and a threshold at that distance leaves
// fewer than k rows below it iff X + m < k.  guess_m: the smallest m with P(X <= k - m - 1) <= target (m = k cannot fail).
uint32_t guess_m(double F, uint32_t k, double target)
{
    if (!(F > 0.0)) return k;
    if (F >= 1.0) return k;  // every row has been seen: only the k-th smallest itself leaves k rows below it
    for (uint32_t m = 1; m < k; ++m) {
        double lp = (double)m * std::log(F), cdf = 0.0;  // log P(X = 0)
        const double l1 = std::log1p(-F);
        for (uint32_t x = 0; x + m < k; ++x) {
            cdf += std::exp(lp);
            lp += std::log((double)(x + m) / (double)(x + 1u)) + l1;
        }
        if (cdf <= target) return m;
    }
    return k;
}
// `proven` (retry batches; HVS_GUESS=0): m = k at every level -- the proven threshold, which cannot fail (a query that
// failed under a guess did so because the rows it had seen were unlucky, and a larger guess from the same rows shares
// that luck)
HvsGuessTable plan_guess(uint32_t k, bool proven, uint32_t pfail)
{
    HvsGuessTable G{};
    const double target = std::pow(10.0, -(double)pfail);
    for (int i = 0; i < HVS_GUESS_STEPS; ++i)
        G.m[i] = (uint16_t)((kGuess && !proven) ? guess_m(std::exp2(-(double)i / 8.0), k, target) : k);
    G.floor_m = (uint16_t)std::min(k, (kGuess && !proven) ? kGuessMid : k);
    G.last_m = (uint16_t)((proven || !kGuess) ? k : 0u);
    return G;
}

// candidate keys per slot and round of a batch that runs every level with the PROVEN threshold (retry batches): up to
// k (radix - 1) candidates per query and level, twice that for the band
uint32_t proven_fcap(const hvs_ctx* c)
{
    uint32_t rmax = 2u;
    for (uint32_t j = 1; j <= c->lv.K; ++j) rmax = std::max(rmax, c->lv.radix[j]);
    return std::max<uint32_t>(HVS_FCAP, hvs_ceil_div(2u * c->k * (rmax - 1u), 256u) * 256u);
}
// queries per retry batch: as many as the candidate workspace the call's own batches already allocated can serve with
// proven_fcap keys per slot -- re-running a long list must not ask for 3x the workspace of the batches it came from
uint32_t proven_batch_step(const hvs_ctx* c)
{
    const size_t fit = c->fb_cand_entries / proven_fcap(c);
    const size_t pad = 5u * 32u + (HVS_WG_WAVES + 1u) * HVS_GROUP + HVS_GROUP;  // (slot padding of ensure_filter_workspace)
    uint32_t step = fit > pad + 4096u ? (uint32_t)std::min<size_t>(fit - pad, kBatchMfma) : 4096u;
    step = std::max(512u, step / 512u * 512u);
    return std::min(step, kBatchMfma);
}

// One batch through the filter engine: the resident range [q0, q0 + nqb), or the nqb query indices in the device array
// `list` (retry batches: `proven_last`, failures go to the exact engine).
int run_batch_mfma(hvs_ctx* c, uint32_t q0, uint32_t nqb, uint32_t sn, const uint32_t* list, bool proven_last)
{
    const int fmt = c->tile_fmt;
    const uint32_t want_fcap = proven_last ? proven_fcap(c) : HVS_FCAP;
    int rc = prep_batch(c, q0, nqb, sn == c->n && !list, fmt, false, list, want_fcap);
    if (rc) return rc;
    if ((rc = build_items(c))) return rc;
    HvsBatch& B = c->fb;
    const HvsLevels L = c->lv;
    HvsItems W{c->d_items, c->d_lvloff, c->d_cursor, HVS_SEG};
    const uint32_t n = c->n;
    if (c->guess_k != c->k) {  // (tables are made on first use: 168 x ~100 negative-binomial sums each, ~10 ms on the host)
        for (bool& h : c->guess_have) h = false;
        c->guess_k = c->k;
    }
    const uint32_t gslot = proven_last ? 0u : (c->force_pfail ? c->force_pfail : guess_pfail_for(nqb));  // slot 0: the proven table
    if (!c->guess_have[gslot]) {
        c->guess_tab[gslot] = plan_guess(c->k, proven_last, gslot ? gslot : 3u);
        c->guess_have[gslot] = true;
    }
    const HvsGuessTable G = c->guess_tab[gslot];
    B.fail_code = (kGuess && !proven_last) ? HVS_FAIL_RETRY : HVS_FAIL_EXACT;
    if (sn != n && !list)
        hipLaunchKernelGGL(hvs_k_count_prefix_pairs, dim3(B.nslots), dim3(64), 0, c->stream, B, c->d_perm_ct, c->d_perm_t, sn,
                           c->d_counters);

    if (c->gate_heavy) HVS_HIP(c, hipStreamWaitEvent(c->stream, c->gate_heavy, 0));  // (two lanes: see hvs_ctx::ev_pdone)
    // level 0 by the exact kernel; small batches cut it into chunks so that enough waves are in flight
    const uint32_t l0blocks = L.off[1] - L.off[0];
    const uint32_t seed_waves = hvs_ceil_div(B.nslots, 64u);
    uint32_t seed_chunks = 1u;
    if (l0blocks <= B.fcap / 32u && seed_waves < kSeedWaves) seed_chunks = std::min(l0blocks, hvs_ceil_div(kSeedWaves, seed_waves));
    with_cap(c->cap, [&](auto CAPT) {
        hipLaunchKernelGGL((hvs_k_seed_exact<decltype(CAPT)::value>), dim3((B.nslots + 255u) / 256u, std::max(1u, seed_chunks)), dim3(256), 0,
                           c->stream, c->d_data, n, sn, c->d_q, B, c->d_perm_ct, c->d_perm_t, c->d_bpos_ct, c->d_bpos_t, L, c->d_counters,
                           std::max(1u, seed_chunks));
    });
    // merge behind a level: top-k, and the threshold of level `next` (its order statistic from the guess plan)
    auto launch_merge = [&](bool final, uint32_t next) {
        with_cap(c->cap, [&](auto CAPT) {
            constexpr int CAP = decltype(CAPT)::value;
            if (final)
                hipLaunchKernelGGL((hvs_k_merge<true, CAP>), dim3((B.nslots + 3u) / 4u), dim3(256), 0, c->stream, c->d_data, n, c->d_q, B,
                                   c->d_bounds, c->padding ? 1 : 0, c->d_out_ids, c->d_out_dists, fmt, c->d_quant, L, next, G);
            else
                hipLaunchKernelGGL((hvs_k_merge<false, CAP>), dim3((B.nslots + 3u) / 4u), dim3(256), 0, c->stream, c->d_data, n, c->d_q, B,
                                   c->d_bounds, c->padding ? 1 : 0, c->d_out_ids, c->d_out_dists, fmt, c->d_quant, L, next, G);
        });
    };
    launch_merge(L.K == 0u, 1u);
    // re-scoring blocks per group: each block stages the group's 128 queries in LDS first, so large batches use
    // few long-lived blocks per group (2: -4 % of the step at 262144 queries) and small batches enough blocks to
    // fill the chip
    const uint32_t rescore_blocks = kRescoreBlocks ? kRescoreBlocks : std::max(2u, std::min(8u, hvs_ceil_div(4096u, B.ngroups)));
    // One filter -> re-score -> merge round per level.  (Sharing a round between the two low levels of a small batch --
    // BASELINE configs[1]/[2], 10^4 queries: m (R - 1) = 1000 rows per query to the exact kernel instead of 2 x 60 for one
    // re-score + merge less -- was measured in round 3: 2.85 ms instead of 2.08 ms per 10^4 mixed queries.  Round 2 measured
    // the same for doubling levels.)
    for (uint32_t level = 1; level <= L.K;) {
        const uint32_t last = level;
        // (the groups' entry counters are zero here: hvs_k_prep clears them for the first level, every merge for the next)
        for (; level <= last; ++level) {
            const int ev = kernel_timer_begin(c);
            // a fixed crew of workgroups pulls the level's work items (two resident per CU + spares)
            static const uint32_t kWgsPerCu = env_u32("HVS_FILTER_WGS_PER_CU", 4u, 1u, 16u);
            const dim3 fgrid(kWgsPerCu * (uint32_t)c->num_cus);
            W.segsize = c->segs.seg[level];
            if (fmt == HVS_FMT_I8X16)
                hipLaunchKernelGGL(hvs_k_filter_i8x16, fgrid, dim3(64 * HVS_WG_WAVES), 0, c->stream, c->d_tiles_ct, c->d_tiles_t,
                                   c->d_nrm_ct, c->d_nrm_t, c->d_bpos_ct, c->d_bpos_t, L, level, B, W, c->d_counters);
            else if (fmt == HVS_FMT_I8)
                hipLaunchKernelGGL(hvs_k_filter_mfma<HVS_FMT_I8>, fgrid, dim3(64 * HVS_WG_WAVES), 0, c->stream, c->d_tiles_ct,
                                   c->d_tiles_t, c->d_nrm_ct, c->d_nrm_t, c->d_bpos_ct, c->d_bpos_t, L, level, B, W, c->d_counters);
            else if (fmt == HVS_FMT_F16)
                hipLaunchKernelGGL(hvs_k_filter_mfma<HVS_FMT_F16>, fgrid, dim3(64 * HVS_WG_WAVES), 0, c->stream, c->d_tiles_ct,
                                   c->d_tiles_t, c->d_nrm_ct, c->d_nrm_t, c->d_bpos_ct, c->d_bpos_t, L, level, B, W, c->d_counters);
            else
                hipLaunchKernelGGL(hvs_k_filter_mfma<HVS_FMT_BF16>, fgrid, dim3(64 * HVS_WG_WAVES), 0, c->stream, c->d_tiles_ct,
                                   c->d_tiles_t, c->d_nrm_ct, c->d_nrm_t, c->d_bpos_ct, c->d_bpos_t, L, level, B, W, c->d_counters);
            kernel_timer_end(c, ev);
            if (level == L.K && c->ev_fdone_cur) HVS_HIP(c, hipEventRecord(c->ev_fdone_cur, c->stream));
            if (level + 1u == L.K && c->ev_pdone_cur) HVS_HIP(c, hipEventRecord(c->ev_pdone_cur, c->stream));
        }
        if (fmt == HVS_FMT_I8X16)
            hipLaunchKernelGGL(hvs_k_rescore<true>, dim3(rescore_blocks, B.ngroups), dim3(64 * HVS_RESCORE_WAVES), 0, c->stream, c->d_data, n,
                               sn, c->d_q, B, c->d_perm_ct, c->d_perm_t, c->d_counters);
        else
            hipLaunchKernelGGL(hvs_k_rescore<false>, dim3(rescore_blocks, B.ngroups), dim3(64 * HVS_RESCORE_WAVES), 0, c->stream, c->d_data, n,
                               sn, c->d_q, B, c->d_perm_ct, c->d_perm_t, c->d_counters);
        launch_merge(last == L.K, last + 1u);
    }
    // queries this batch could not answer go on the call's lists (retry with a proven threshold / exact engine); they
    // are answered again when the call's results are first needed (resolve_overflow) -- the batch never waits for the host
    hipLaunchKernelGGL(hvs_k_collect_overflow, dim3((B.nslots + 255u) / 256u), dim3(256), 0, c->stream, B, c->d_ovf_list,
                       c->d_ovf_count, c->d_retry_list, c->d_ovf_count + 1);
    HVS_HIP(c, hipGetLastError());
    return HVS_OK;
}

// Queries the call's batches could not answer.  First the retry list (a guessed threshold that failed its check, or a list
// that overflowed under it): filter batches whose last level uses the proven threshold; what fails there joins the
// exact list.  Then the exact engine re-runs the exact list (rare: thousands of equal distances, queries far outside the
// data's bounding box, non-finite components).  Called wherever a call's results or timing leave the library; costs
// one stream synchronisation per call (two more when there is something to re-run).
int resolve_overflow(hvs_ctx* c)
{
    if (!c->ovf_pending) return HVS_OK;
    HVS_HIP(c, hipSetDevice(c->device));
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    c->ovf_pending = false;
    c->demoted_queries = 0;
    uint32_t novf = c->h_ovf[0];
    const uint32_t nretry = c->h_ovf[1];
    if (novf == 0u && nretry == 0u) return HVS_OK;
    // what is left of a list when a re-run batch cannot get its workspace: appended to the exact engine's list, which needs none
    auto rest_to_exact = [&](const uint32_t* list, uint32_t count) -> int {
        (void)hipGetLastError();
        HVS_HIP(c, hipMemcpyAsync(c->h_ovf, c->d_ovf_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HVS_HIP(c, hipStreamSynchronize(c->stream));
        const uint32_t have = c->h_ovf[0];
        HVS_HIP(c, hipMemcpyAsync(c->d_ovf_list + have, list, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        const uint32_t total = have + count;
        HVS_HIP(c, hipMemcpyAsync(c->d_ovf_count, &total, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HVS_HIP(c, hipStreamSynchronize(c->stream));
        c->err.clear();
        return HVS_OK;
    };
    if (nretry) {
        c->retry_queries = nretry;
        const uint32_t step = proven_batch_step(c);
        for (uint32_t off = 0; off < nretry; off += step) {
            const uint32_t m = std::min(step, nretry - off);
            int rc = run_batch_mfma(c, 0, m, c->pend_sn, c->d_retry_list + off, true);
            if (rc == HVS_ENOMEM) {
                if ((rc = rest_to_exact(c->d_retry_list + off, nretry - off))) return rc;
                break;
            }
            if (rc) return rc;
        }
        HVS_HIP(c, hipMemcpyAsync(c->h_ovf, c->d_ovf_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HVS_HIP(c, hipStreamSynchronize(c->stream));
        novf = c->h_ovf[0];
    }
    // Many queries without a usable INT8 bound in one call (the planner's probe uses rows of D as queries: real queries can
    // lie far outside the data's box, where the clip term drowns the band): under HVS_ENGINE_AUTO the 16-bit float tiles
    // are built now -- for this call's list and for the calls to come -- instead of sending them all to the exact engine
    // at 1 % of a filter's rate.  The list moves aside (the filter batches append their own failures to the exact list).
    c->demoted_queries = 0;
    if (c->engine == HVS_ENGINE_AUTO && HVS_IS_I8(c->tile_fmt) && novf >= std::max(256u, c->timing.nq / 50u) &&
        env_u32("HVS_DEMOTE", 1u, 0u, 1u)) {
        if (c->demote_cap < c->res_cap) {
            int rc = dev_alloc(c, &c->d_demote_list, (size_t)c->res_cap);
            if (rc) return rc;
            c->demote_cap = c->res_cap;
        }
        // The 16-bit float tiles first: when HBM has no room for them next to D (they are 1.7x the INT8 tiles) the INT8
        // tiles come back and the exact engine answers the list as it always did -- slower, never an error.
        const int had = c->tile_fmt;
        int rc = build_tiles_chain(c, c->f16_rejected ? HVS_FMT_BF16 : HVS_FMT_F16);
        if (rc == HVS_ENOMEM || (!rc && !c->have_index)) {
            (void)hipGetLastError();
            c->err.clear();
            rc = build_tiles_chain(c, had);
            if (rc == HVS_ENOMEM) {
                (void)hipGetLastError();
                c->err.clear();
                rc = HVS_OK;
            }
            if (rc) return rc;
            if (!c->have_index) c->have_index = true;  // (orderings only: the exact engine's range scans)
        } else if (rc) {
            return rc;
        } else {
            HVS_HIP(c, hipMemcpyAsync(c->d_demote_list, c->d_ovf_list, (size_t)novf * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
            HVS_HIP(c, hipMemsetAsync(c->d_ovf_count, 0, sizeof(uint32_t), c->stream));
            c->planned_fmt = c->tile_fmt;
            const uint32_t step = proven_batch_step(c);
            uint32_t done = novf;
            for (uint32_t off = 0; off < novf; off += step) {
                const uint32_t m = std::min(step, novf - off);
                rc = run_batch_mfma(c, 0, m, c->pend_sn, c->d_demote_list + off, true);
                if (rc == HVS_ENOMEM) {
                    if ((rc = rest_to_exact(c->d_demote_list + off, novf - off))) return rc;
                    done = off;
                    break;
                }
                if (rc) return rc;
            }
            c->demoted_queries = done;
            HVS_HIP(c, hipMemcpyAsync(c->h_ovf, c->d_ovf_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HVS_HIP(c, hipStreamSynchronize(c->stream));
            novf = c->h_ovf[0];
            c->timing.flags |= HVS_TIMING_FORMAT_CHANGED;
            c->timing.engine = c->tile_fmt == HVS_FMT_F16 ? HVS_ENGINE_MFMA_F16 : HVS_ENGINE_MFMA_FILTER;
        }
    }
    c->fallback_queries = novf;
    c->timing.fallback_queries = novf;
    c->timing.retry_queries = nretry;
    for (uint32_t off = 0; off < novf; off += kBatch) {
        const uint32_t m = std::min(kBatch, novf - off);
        int rc = run_batch_exact(c, 0, m, c->pend_sn, c->d_ovf_list + off, false, false);
        if (rc) return rc;
    }
    HVS_HIP(c, hipEventRecord(c->ev_q1, c->stream));  // the re-runs belong to the call's device time
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    return HVS_OK;
}

// Host copy between pageable and pinned memory.  One thread moves ~10 GB/s; the first batch's queries and the last
// batch's results of a call sit on the critical path (nothing computes under them), so pieces of 4 MB and more are cut
// over up to 4 threads.
void staging_copy(void* dst, const void* src, size_t bytes)
{
    static const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    static const unsigned kMaxParts = env_u32("HVS_STAGE_THREADS", 8u, 1u, 64u);  // (4 until round 4: ~15 GB/s; 8: ~25 GB/s on the GPU boxes' hosts)
    const unsigned parts = bytes >= (4u << 20) ? std::min(kMaxParts, std::max(1u, hw / 4u)) : 1u;
    if (parts <= 1u) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t chunk = ((bytes / parts) + 4095u) & ~(size_t)4095u;
    std::vector<std::thread> th;
    th.reserve(parts - 1u);
    for (unsigned p = 1; p < parts; ++p) {
        const size_t off = (size_t)p * chunk;
        if (off >= bytes) break;
        const size_t len = std::min(chunk, bytes - off);
        th.emplace_back([=]() { std::memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, len); });
    }
    std::memcpy(dst, src, std::min(chunk, bytes));
    for (auto& t : th) t.join();
}

// Batch schedule of one call (sizes in queries, in order).  Resident runs and the exact engines cut the call into
// full batches.  `host_pipeline` (hvs_query): the first batch's queries and the last batch's results are the only
// transfers the pipeline cannot hide, so those two batches are small (kBatchMfma / 8) once the call is large enough to
// pay for two extra batches, and the batches between share the rest evenly in whole quads of 512 queries.
// The host pipeline of hvs_query stages its input by THIS schedule (the hooks receive the sizes), never by a guess of it.
std::vector<uint32_t> batch_schedule(uint32_t nq, uint32_t step, bool ramp_allowed)
{
    std::vector<uint32_t> out;
    // (calls below 2^20 queries -- e.g. one GPU's share of BASELINE configs[3] on an 8-GPU node, 5 x 10^5 -- stay ONE batch:
    // with the staging copies cut over 4 threads the two exposed transfers cost ~9 ms, less than what two small edge
    // batches lose in efficiency; measured in round 3, 5 x 10^5 queries host -> host: one batch 178.5 ms, ramped 185.1 ms,
    // two / three equal batches 183.3 / 189.0 ms, resident 169 ms)
    const uint32_t edge = kBatchMfma / 8u;
    const bool ramp = ramp_allowed && edge >= 1024u && nq >= 4u * edge;
    // (A/B, HVS_SPLIT_SMALL=1: a small call as two half batches on the two lanes, ungated -- see run_queries)
    static const bool kSplitSmall = env_u32("HVS_SPLIT_SMALL", 0u, 0u, 1u) != 0u;
    if (kSplitSmall && step == kBatchMfma && nq >= 4096u && nq <= 65536u) {
        const uint32_t half = hvs_ceil_div(hvs_ceil_div(nq, 2u), 512u) * 512u;
        out.push_back(half);
        out.push_back(nq - half);
        return out;
    }
    if (!ramp) {
        for (uint32_t off = 0; off < nq; off += step) out.push_back(std::min(step, nq - off));
        return out;
    }
    out.push_back(edge);
    uint32_t middle_left = nq - 2u * edge;
    uint32_t middle_batches = hvs_ceil_div(middle_left, step);
    while (middle_left) {
        const uint32_t nqb = std::min(middle_left, hvs_ceil_div(hvs_ceil_div(middle_left, middle_batches), 512u) * 512u);
        out.push_back(nqb);
        middle_left -= nqb;
        middle_batches -= 1u;
    }
    out.push_back(edge);
    return out;
}

// Hooks of run_queries: `before(off, nqb)` is called before the kernels of a batch are enqueued (hvs_query makes the
// compute stream wait for the batch's queries there: a batch never starts before its input is on the device);
// `after(off, nqb, next_nqb)` after they have been enqueued (the host pipeline of hvs_query hangs its copies there);
// queries [q0 + off, q0 + off + nqb) are complete on the stream at that point, except for overflowed ones
// (resolve_overflow).
template <typename Hooks>
int run_queries(hvs_ctx* c, uint32_t q0, uint32_t nq, float sample_proportion, Hooks hooks, bool host_pipeline = false)
{
    if (!c->d_data) return fail(c, HVS_ESTATE, "no data set loaded (hvs_load_data / hvs_gen_data)");
    if ((uint64_t)q0 + nq > c->nq) return fail(c, HVS_EINVAL, "query range outside the resident query set");
    HVS_HIP(c, hipSetDevice(c->device));
    {
        int rc = resolve_overflow(c);  // an earlier call whose results were never fetched
        if (rc) return rc;
    }
    const uint32_t sn = sample_rows(sample_proportion, c->n);
    // The index orders ALL rows: with a sampled prefix [0,sn) the filter still proposes rows >= sn and the
    // exact stages drop them, so its candidate lists grow by n/sn -- used down to sn = n/4, below that
    // the exact engine answers.
    bool mfma = c->have_index && sn >= c->n / 4u && sn > 0u && !c->scalar_order &&
                (c->engine == HVS_ENGINE_MFMA_FILTER || c->engine == HVS_ENGINE_MFMA_I8 || c->engine == HVS_ENGINE_MFMA_F16 ||
                 (c->engine == HVS_ENGINE_AUTO && c->n >= kMfmaMinRows && c->planned_fmt != HVS_FMT_NONE));
    if (mfma) {
        // the tiles exist in one format at a time: an engine choice made after the load rebuilds them
        const int want = want_format(c);
        if (want != c->tile_fmt) {
            int rc = build_tiles_chain(c, want);
            if (rc) return rc;
            if (!c->have_index) {  // no usable bound at all: orderings only, the exact engine answers
                c->have_index = true;
                mfma = false;
            }
        }
    }
    c->timing_valid = false;
    c->n_launch_events = 0;
    c->untimed_launches = 0;
    c->fallback_queries = 0;
    c->retry_queries = 0;
    c->demoted_queries = 0;  // (a list of an earlier call must never be scattered into this call's results)
    HVS_HIP(c, hipMemsetAsync(c->d_counters, 0, 16 * sizeof(unsigned long long), c->stream));
    HVS_HIP(c, hipMemsetAsync(c->d_ovf_count, 0, 2 * sizeof(uint32_t), c->stream));
    HVS_HIP(c, hipEventRecord(c->ev_q0, c->stream));
    const bool ranges = !mfma && c->have_index;  // exact engine: scan position ranges when the index exists
    const std::vector<uint32_t> sched = batch_schedule(nq, mfma ? kBatchMfma : kBatch, host_pipeline && mfma);
    // Two lanes (HvsLane): every other batch of a filter-engine call runs on the spare lane's stream and workspace, so that
    // its preparation, seed and low levels share the chip with the previous batch's last re-scoring and final merge.  The
    // spare lane starts behind the call's counter resets (ev_q0) and the main stream joins it before the call ends.
    bool two_lanes = mfma && kLanes && sched.size() >= 2 && !c->lanes_failed;
    bool spare_used = false;
    struct JoinOnError {  // a call that fails half-way leaves no kernels running on the spare lane behind the caller's back
        hvs_ctx* c;
        const bool* used;
        bool ok = false;
        ~JoinOnError()
        {
            if (!ok && *used && c->spare.stream) (void)hipStreamSynchronize(c->spare.stream);
        }
    } join_on_error{c, &spare_used};
    uint32_t off = 0;
    for (size_t b = 0; b < sched.size(); ++b) {
        const uint32_t nqb = sched[b];
        LaneGuard lane{c, false};
        if (two_lanes && (b & 1u)) {
            swap_lanes(c);
            lane.on = true;
            // the spare workspace is allocated here, ahead of the batch: without room for it the call stays on one lane
            // (HVS_TEST_NO_SPARE=1: tests take the out-of-memory path without filling 288 GB)
            static const bool kTestNoSpare = env_u32("HVS_TEST_NO_SPARE", 0u, 0u, 1u) != 0u;
            int rcw = kTestNoSpare ? HVS_ENOMEM : ensure_filter_workspace(c, nqb);
            if (!rcw) rcw = ensure_items(c);
            if (rcw == HVS_ENOMEM) {
                (void)hipGetLastError();
                c->err.clear();
                swap_lanes(c);
                lane.on = false;
                two_lanes = false;
                c->lanes_failed = true;
            } else if (rcw) {
                return rcw;
            } else if (!spare_used) {
                HVS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_q0, 0));
                spare_used = true;
            }
        }
        if (two_lanes) {
            // this batch's preparation starts behind the previous batch's (other lane) second-to-last filter launch, its seed
            // behind the last one; it records its own two events
            const bool have_prev = b > 0 && c->lv.K >= 1u && nq > 65536u;  // (small calls, HVS_SPLIT_SMALL: no gates)
            if (have_prev) HVS_HIP(c, hipStreamWaitEvent(c->stream, c->lv.K >= 2u ? c->ev_pdone[(b - 1u) & 1u] : c->ev_fdone[(b - 1u) & 1u], 0));
            c->gate_heavy = have_prev ? c->ev_fdone[(b - 1u) & 1u] : nullptr;
            c->ev_fdone_cur = c->ev_fdone[b & 1u];
            c->ev_pdone_cur = c->ev_pdone[b & 1u];
        }
        int rc = hooks.before(off, nqb);
        if (!rc)
            rc = mfma ? run_batch_mfma(c, q0 + off, nqb, sn, nullptr, false)
                      : (ranges ? run_batch_exact_ranges(c, q0 + off, nqb, sn) : run_batch_exact(c, q0 + off, nqb, sn));
        c->ev_fdone_cur = c->ev_pdone_cur = c->gate_heavy = nullptr;
        if (rc) return rc;
        if ((rc = hooks.after(off, nqb, b + 1 < sched.size() ? sched[b + 1] : 0u))) return rc;
        if (lane.on) HVS_HIP(c, hipEventRecord(c->ev_lane, c->stream));
        off += nqb;
    }
    if (spare_used) HVS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_lane, 0));
    HVS_HIP(c, hipEventRecord(c->ev_q1, c->stream));
    if (mfma) {
        HVS_HIP(c, hipMemcpyAsync(c->h_ovf, c->d_ovf_count, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        c->ovf_pending = true;
        c->pend_sn = sn;
    }
    c->timing = hvs_timing{};
    c->timing.nq = nq;
    c->timing.engine = mfma ? (HVS_IS_I8(c->tile_fmt) ? HVS_ENGINE_MFMA_I8 : c->tile_fmt == HVS_FMT_F16 ? HVS_ENGINE_MFMA_F16 : HVS_ENGINE_MFMA_FILTER)
                          : HVS_ENGINE_EXACT_SCAN;
    c->timing.load_ms = c->load_ms;
    c->timing.n_gpus = 1;
    c->timing.flags = (c->index_too_large ? HVS_TIMING_INDEX_TOO_LARGE : 0u) | (mfma && HVS_IS_I8(c->tile_fmt) && c->i8_rot_built ? HVS_TIMING_I8_ROTATED : 0u);
    c->timing_valid = true;
    join_on_error.ok = true;
    return HVS_OK;
}

// ---------------------------------------------------------------------------------------------
// leaf (one GPU) implementations of the C ABI; the multi-GPU root dispatches to them
// ---------------------------------------------------------------------------------------------
struct NoHook {
    int before(uint32_t, uint32_t) const { return HVS_OK; }
    int after(uint32_t, uint32_t, uint32_t) const { return HVS_OK; }
};

int leaf_create(hvs_ctx** out, int device)
{
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        g_global_err = std::string("hvs_create: no HIP device available (") + hipGetErrorString(e) +
                       "); this library has no CPU fallback";
        return HVS_EHIP;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= count) {
        g_global_err = "hvs_create: device index out of range";
        return HVS_EINVAL;
    }
    hvs_ctx* c = new (std::nothrow) hvs_ctx();
    if (!c) {
        g_global_err = "hvs_create: out of host memory";
        return HVS_ENOMEM;
    }
    c->device = device;
    auto bail = [&](const char* what, hipError_t err) {
        g_global_err = std::string("hvs_create: ") + what + ": " + hipGetErrorString(err);
        hvs_destroy(c);
        return HVS_EHIP;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->num_cus = cus;
    }
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipStreamCreateWithFlags(&c->spare.stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_lane, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    for (hipEvent_t& ev : c->ev_fdone)
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    for (hipEvent_t& ev : c->ev_pdone)
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreate(&c->ev_q0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev_q1)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_batch, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    for (int i = 0; i < hvs_ctx::kRing; ++i) {
        if ((e = hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    }
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->d_counters), 16 * sizeof(unsigned long long))) != hipSuccess)
        return bail("hipMalloc", e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->d_ovf_count), 2 * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->d_layout), 16 * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_ovf), 2 * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess)
        return bail("hipHostMalloc", e);
    c->h_ovf[0] = c->h_ovf[1] = 0u;
    // The first operation on a stream creates its hardware queue (and the first copy in each direction its DMA path): a few
    // milliseconds that belong here, not inside the first query (round 3's cold hvs_query of 10^4 queries took 8 ms for 2 ms
    // of device work).
    (void)hipMemsetAsync(c->d_ovf_count, 0, 2 * sizeof(uint32_t), c->stream);
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemsetAsync(c->d_ovf_count, 0, 2 * sizeof(uint32_t), c->spare.stream);
    (void)hipStreamSynchronize(c->spare.stream);
    (void)hipMemcpyAsync(c->d_ovf_count, c->h_ovf, 2 * sizeof(uint32_t), hipMemcpyHostToDevice, c->s_in);
    (void)hipStreamSynchronize(c->s_in);
    (void)hipMemcpyAsync(c->h_ovf, c->d_ovf_count, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->s_out);
    (void)hipEventRecord(c->ev_batch, c->stream);
    (void)hipStreamWaitEvent(c->s_out, c->ev_batch, 0);
    (void)hipStreamSynchronize(c->s_out);
    (void)hipGetLastError();
    *out = c;
    return HVS_OK;
}

void leaf_destroy(hvs_ctx* c)
{
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->s_in) (void)hipStreamSynchronize(c->s_in);
    if (c->s_out) (void)hipStreamSynchronize(c->s_out);
    if (c->spare.stream) (void)hipStreamSynchronize(c->spare.stream);
    void* ptrs[] = {c->d_data, c->d_q, c->d_out_ids, c->d_out_dists, c->d_counters, c->d_bounds, c->d_quant,
                    c->d_ovf_list, c->d_ovf_count, c->d_retry_list, c->d_demote_list};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    free_index(c);
    free_lane(c->spare);
    free_lane(static_cast<HvsLane&>(*c));
    if (c->ev_lane) (void)hipEventDestroy(c->ev_lane);
    for (hipEvent_t ev : c->ev_fdone)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : c->ev_pdone)
        if (ev) (void)hipEventDestroy(ev);
    if (c->h_ovf) (void)hipHostFree(c->h_ovf);
    for (int i = 0; i < hvs_ctx::kRing; ++i) {
        if (c->h_in[i]) (void)hipHostFree(c->h_in[i]);
        if (c->h_out_ids[i]) (void)hipHostFree(c->h_out_ids[i]);
        if (c->h_out_dists[i]) (void)hipHostFree(c->h_out_dists[i]);
        if (c->ev_in[i]) (void)hipEventDestroy(c->ev_in[i]);
        if (c->ev_out[i]) (void)hipEventDestroy(c->ev_out[i]);
    }
    if (c->ev_batch) (void)hipEventDestroy(c->ev_batch);
    if (c->ev_stage) (void)hipEventDestroy(c->ev_stage);
    if (c->ev_q0) (void)hipEventDestroy(c->ev_q0);
    if (c->ev_q1) (void)hipEventDestroy(c->ev_q1);
    for (hipEvent_t e : c->ev_k) (void)hipEventDestroy(e);
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
}

int ensure_staging(hvs_ctx* c, bool dists, uint64_t nq_in, uint64_t nq_out);

// query / result buffers and the batch workspace for calls of up to nq queries (hvs_reserve, hvs_query): keeps the
// ~34 GB of per-batch state of a 2^21-query batch out of the first query's own time
int leaf_reserve(hvs_ctx* c, uint32_t nq)
{
    HVS_HIP(c, hipSetDevice(c->device));
    c->reserve_nq = std::max(c->reserve_nq, nq);
    if (nq == 0u) return HVS_OK;
    int rc = ensure_queries(c, nq);
    if (rc) return rc;
    // pinned staging of the host path (ids only: the distance slots follow the first call that asks for distances)
    if ((rc = ensure_staging(c, false, nq, nq))) return rc;
    if (!c->dma_warm && c->nq == 0u && c->h_out_ids[0] && c->h_in[0] && c->d_out_ids && c->d_q) {  // (no resident queries to overwrite)
        // the first DMA-sized copy in each direction sets up its engine queue (a cold 4 MB D2H took 6 ms): here, not in the first query
        const size_t bytes = std::min<size_t>((size_t)1 << 20, std::min((size_t)c->out_cap_q[0] * c->stage_k * sizeof(uint32_t),
                                                                      (size_t)c->in_cap_q[0] * HVS_QCOLS * sizeof(float)));
        if (bytes && (size_t)nq * c->k * sizeof(uint32_t) >= bytes) {
            (void)hipMemcpyAsync(c->h_out_ids[0], c->d_out_ids, bytes, hipMemcpyDeviceToHost, c->s_out);
            (void)hipMemcpyAsync(c->d_q, c->h_in[0], bytes, hipMemcpyHostToDevice, c->s_in);
            (void)hipStreamSynchronize(c->s_out);
            (void)hipStreamSynchronize(c->s_in);
            (void)hipGetLastError();
            c->dma_warm = true;
        }
    }
    // (the filter workspace only when a filter engine is going to run: 4096 <= n < 32768 under AUTO has an index for the
    // range scans of the exact engine, whose batches are kBatch queries)
    const bool filter_runs = c->have_index && (c->engine == HVS_ENGINE_MFMA_FILTER || c->engine == HVS_ENGINE_MFMA_I8 ||
                                               c->engine == HVS_ENGINE_MFMA_F16 ||
                                               (c->engine == HVS_ENGINE_AUTO && c->n >= kMfmaMinRows && c->planned_fmt != HVS_FMT_NONE));
    if (filter_runs) {
        if ((rc = ensure_filter_workspace(c, std::min(nq, kBatchMfma)))) return rc;
        if ((rc = ensure_items(c))) return rc;
        // calls of two batches and more alternate between two lanes: the spare lane's largest batch under either schedule (the
        // resident API's full batches, hvs_query's ramped ones -- a call of 2 x 10^6 queries is ONE batch resident and three from
        // host memory).  Without this the first such call allocated ~38 GB inside its own time (2.7 s on a 4 x 10^6-query call
        // over two virtual ranks).
        uint32_t spare_nq = 0;
        for (int host = 0; host < 2; ++host) {
            const std::vector<uint32_t> sched = batch_schedule(nq, kBatchMfma, host != 0);
            for (size_t b = 1; b < sched.size(); b += 2) spare_nq = std::max(spare_nq, sched[b]);
        }
        if (kLanes && spare_nq && !c->lanes_failed) {
            LaneGuard lane{c, true};
            swap_lanes(c);
            rc = ensure_filter_workspace(c, spare_nq);
            if (!rc) rc = ensure_items(c);
            if (rc == HVS_ENOMEM) {
                (void)hipGetLastError();
                c->err.clear();
                c->lanes_failed = true;
                rc = HVS_OK;
            }
        }
        return rc;
    }
    if (c->have_index) return ensure_filter_workspace(c, std::min(nq, kBatch));
    const uint32_t nqb = std::min(nq, kBatch);
    return ensure_batch_workspace(c, nqb, make_plan(nqb, c->n ? c->n : 1u));
}

int begin_data(hvs_ctx* c, uint32_t n)
{
    if (n < c->k)
        return fail(c, HVS_EINVAL,
                    "data set needs at least k (default 100) rows (the reference pads results with rows n-1, n-2, ...)");
    HVS_HIP(c, hipSetDevice(c->device));
    int rc = resolve_overflow(c);
    if (rc) return rc;
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    c->n = 0;
    return dev_alloc(c, &c->d_data, (size_t)n * HVS_DCOLS);
}

// upload/generation is timed by ev_q0..ev_q1; the index build (orderings + tiles) follows
int finish_data(hvs_ctx* c)
{
    float ms = 0.f;
    HVS_HIP(c, hipEventElapsedTime(&ms, c->ev_q0, c->ev_q1));
    c->load_ms = ms;
    free_index(c);
    // the index (two orderings + tiles) serves both engines: the exact engine scans position ranges
    if (c->n < kIndexMinRows && c->engine != HVS_ENGINE_MFMA_FILTER && c->engine != HVS_ENGINE_MFMA_I8 &&
        c->engine != HVS_ENGINE_MFMA_F16)
        return HVS_OK;
    HVS_HIP(c, hipEventRecord(c->ev_q0, c->stream));
    int rc = build_index(c);
    if (rc == HVS_ENOMEM) {
        // not enough HBM for the index next to D: the data set stays usable through the exact engine
        (void)hipGetLastError();
        free_index(c);
        c->err = "index not built (out of device memory): exact engine only";
        rc = HVS_OK;
    }
    if (rc) return rc;
    HVS_HIP(c, hipEventRecord(c->ev_q1, c->stream));
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    HVS_HIP(c, hipEventElapsedTime(&ms, c->ev_q0, c->ev_q1));
    c->index_ms = ms;
    c->load_ms += ms;
    trace_mark(c, "index");
    if (c->reserve_nq) {  // the caller announced its call size (hvs_reserve)
        // not fatal: D and the index are loaded; the first query allocates what it needs (or reports the failure itself)
        if (leaf_reserve(c, c->reserve_nq) == HVS_ENOMEM) {
            (void)hipGetLastError();
            c->err = "batch workspace not reserved (out of device memory): allocated by the first query";
        }
    }
    return HVS_OK;
}

bool host_pointer_is_pinned(const void* p)
{
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // ordinary (pageable) host memory is "invalid value" to this query
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// Pinned staging slots of the host pipeline, sized by the call: a call of nq_in queries (or the same volume of data rows)
// going in and nq_out result rows coming out uses ceil(nq / 65536) slots of each ring, at most 4, each as large as its
// piece (whole slots are 27 + 26 + 26 MB; round 3 allocated all twelve inside the first call, 6 of the 8 ms a cold call of
// 10^4 queries took).  Pinned allocations cost ~1 ms per 25 MB: hvs_reserve makes them ahead of the first call.
int ensure_staging(hvs_ctx* c, bool dists, uint64_t nq_in, uint64_t nq_out)
{
    constexpr uint32_t SQ = hvs_ctx::kStageQ;
    if (c->stage_k < c->k) {  // k grew (hvs_set_k): the result slots are too small
        for (int i = 0; i < hvs_ctx::kRing; ++i) {
            if (c->h_out_ids[i]) (void)hipHostFree(c->h_out_ids[i]);
            if (c->h_out_dists[i]) (void)hipHostFree(c->h_out_dists[i]);
            c->h_out_ids[i] = nullptr;
            c->h_out_dists[i] = nullptr;
            c->out_cap_q[i] = c->outd_cap_q[i] = 0u;
        }
        c->stage_k = c->k;
    }
    auto slot_q = [&](uint64_t nq, int i) -> uint32_t {  // queries slot i has to hold (0: the call does not reach it)
        if (nq <= (uint64_t)i * SQ) return 0u;
        const uint64_t piece = nq > (uint64_t)hvs_ctx::kRing * SQ ? SQ : std::min<uint64_t>(SQ, nq - (uint64_t)i * SQ);
        return (uint32_t)std::min<uint64_t>(SQ, (piece + 4095u) / 4096u * 4096u);
    };
    for (int i = 0; i < hvs_ctx::kRing; ++i) {
        const uint32_t qi = slot_q(nq_in, i), qo = slot_q(nq_out, i);
        if (qi > c->in_cap_q[i]) {
            if (c->h_in[i]) (void)hipHostFree(c->h_in[i]);
            c->h_in[i] = nullptr;
            c->in_cap_q[i] = 0u;
            HVS_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_in[i]), (size_t)qi * HVS_QCOLS * sizeof(float), hipHostMallocDefault));
            c->in_cap_q[i] = qi;
        }
        if (qo > c->out_cap_q[i]) {
            if (c->h_out_ids[i]) (void)hipHostFree(c->h_out_ids[i]);
            c->h_out_ids[i] = nullptr;
            c->out_cap_q[i] = 0u;
            HVS_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_out_ids[i]), (size_t)qo * c->stage_k * sizeof(uint32_t), hipHostMallocDefault));
            c->out_cap_q[i] = qo;
        }
        if (dists && qo > c->outd_cap_q[i]) {
            if (c->h_out_dists[i]) (void)hipHostFree(c->h_out_dists[i]);
            c->h_out_dists[i] = nullptr;
            c->outd_cap_q[i] = 0u;
            HVS_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_out_dists[i]), (size_t)qo * c->stage_k * sizeof(float), hipHostMallocDefault));
            c->outd_cap_q[i] = qo;
        }
    }
    return HVS_OK;
}

// H2D of rows into c->d_data (or any device buffer), from pageable or pinned host memory.  Pageable sources go
// through the pinned ring in 27 MB pieces so that the host copy of piece i+1 overlaps the DMA of piece i
// (reference io.h:111-136 is replaced by the caller's bulk read; this replaces the staging an H2D of pageable memory
// would do inside the runtime, at roughly twice its rate).
int upload_rows(hvs_ctx* c, float* dst, const float* src, size_t nfloats)
{
    if (nfloats == 0) return HVS_OK;
    if (host_pointer_is_pinned(src)) {
        HVS_HIP(c, hipMemcpyAsync(dst, src, nfloats * sizeof(float), hipMemcpyHostToDevice, c->stream));
        return HVS_OK;
    }
    // (in units of query rows: 104 floats; data rows move through the same slots)
    int rc = ensure_staging(c, false, (nfloats + HVS_QCOLS - 1u) / HVS_QCOLS, 0u);
    if (rc) return rc;
    const size_t slot = (size_t)hvs_ctx::kStageQ * HVS_QCOLS;  // floats per slot
    size_t off = 0;
    for (int i = 0; off < nfloats; ++i, off += slot) {
        const int k = i % hvs_ctx::kRing;
        const size_t m = std::min(slot, nfloats - off);
        if (i >= hvs_ctx::kRing) HVS_HIP(c, hipEventSynchronize(c->ev_in[k]));  // the slot's previous DMA is done
        staging_copy(c->h_in[k], src + off, m * sizeof(float));
        HVS_HIP(c, hipMemcpyAsync(dst + off, c->h_in[k], m * sizeof(float), hipMemcpyHostToDevice, c->s_in));
        HVS_HIP(c, hipEventRecord(c->ev_in[k], c->s_in));
    }
    HVS_HIP(c, hipStreamSynchronize(c->s_in));
    return HVS_OK;
}

// upload only (the multi-GPU root lets the other GPUs copy D from this one while it builds its index)
int leaf_upload_data(hvs_ctx* c, const float* rows, uint32_t n)
{
    int rc = begin_data(c, n);
    if (rc) return rc;
    trace_mark(c, "alloc");
    HVS_HIP(c, hipEventRecord(c->ev_q0, c->stream));
    if ((rc = upload_rows(c, c->d_data, rows, (size_t)n * HVS_DCOLS))) return rc;
    HVS_HIP(c, hipEventRecord(c->ev_q1, c->stream));
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    c->n = n;
    return HVS_OK;
}

int leaf_load_data(hvs_ctx* c, const float* rows, uint32_t n)
{
    HostTrace tr;
    c->trace = &tr;
    int rc = leaf_upload_data(c, rows, n);
    trace_mark(c, "upload");
    if (!rc) rc = finish_data(c);
    trace_mark(c, "done");
    tr.flush("hvs_load_data", n);
    c->trace = nullptr;
    return rc;
}

// D replicated from another GPU's copy (xGMI peer copy; same-device contexts: a device-to-device copy)
int leaf_load_data_from_peer(hvs_ctx* c, const hvs_ctx* src)
{
    const uint32_t n = src->n;
    int rc = begin_data(c, n);
    if (rc) return rc;
    HVS_HIP(c, hipEventRecord(c->ev_q0, c->stream));
    HVS_HIP(c, hipMemcpyPeerAsync(c->d_data, c->device, src->d_data, src->device, (size_t)n * HVS_DCOLS * sizeof(float), c->stream));
    HVS_HIP(c, hipEventRecord(c->ev_q1, c->stream));
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    c->n = n;
    return finish_data(c);
}

int leaf_gen_data(hvs_ctx* c, uint32_t n, uint64_t seed, int profile, uint32_t ncat)
{
    int rc = begin_data(c, n);
    if (rc) return rc;
    HVS_HIP(c, hipEventRecord(c->ev_q0, c->stream));
    hipLaunchKernelGGL(hvs_k_gen_data, dim3(256 * 8), dim3(256), 0, c->stream, c->d_data, (uint64_t)n * HVS_DCOLS,
                       seed, profile, ncat);
    HVS_HIP(c, hipGetLastError());
    HVS_HIP(c, hipEventRecord(c->ev_q1, c->stream));
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    c->n = n;
    return finish_data(c);
}

int leaf_begin_queries(hvs_ctx* c, uint32_t nq)
{
    HVS_HIP(c, hipSetDevice(c->device));
    int rc = resolve_overflow(c);
    if (rc) return rc;
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    c->nq = 0;
    return ensure_queries(c, nq);
}

int leaf_upload_queries(hvs_ctx* c, const float* q_rows, uint32_t nq)
{
    int rc = leaf_begin_queries(c, nq);
    if (rc) return rc;
    if ((rc = upload_rows(c, c->d_q, q_rows, (size_t)nq * HVS_QCOLS))) return rc;
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    c->nq = nq;
    return HVS_OK;
}

int leaf_gen_queries(hvs_ctx* c, uint32_t nq, uint64_t seed, int profile, uint32_t ncat, int force_type, uint64_t first_row)
{
    int rc = leaf_begin_queries(c, nq);
    if (rc) return rc;
    if (nq) {
        hipLaunchKernelGGL(hvs_k_gen_queries, dim3(256 * 4), dim3(256), 0, c->stream, c->d_q, (uint64_t)nq * HVS_QCOLS,
                           seed, profile, ncat, force_type, first_row);
        HVS_HIP(c, hipGetLastError());
    }
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    c->nq = nq;
    return HVS_OK;
}

int leaf_sync(hvs_ctx* c)
{
    HVS_HIP(c, hipSetDevice(c->device));
    int rc = resolve_overflow(c);
    if (rc) return rc;
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    return HVS_OK;
}

int leaf_download_results(hvs_ctx* c, uint32_t q0, uint32_t nq, uint32_t* out_ids, float* out_dists)
{
    if (!out_ids || (uint64_t)q0 + nq > c->nq) return fail(c, HVS_EINVAL, "hvs_download_results: bad range");
    int rc = leaf_sync(c);
    if (rc) return rc;
    HVS_HIP(c, hipMemcpyAsync(out_ids, c->d_out_ids + (size_t)q0 * c->k, (size_t)nq * c->k * sizeof(uint32_t),
                              hipMemcpyDeviceToHost, c->stream));
    if (out_dists)
        HVS_HIP(c, hipMemcpyAsync(out_dists, c->d_out_dists + (size_t)q0 * c->k,
                                  (size_t)nq * c->k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    return HVS_OK;
}

// The vec_query-equivalent region with host buffers on both sides (reference src/test.cpp:82-88), pipelined:
//   copy-in stream   pinned slot <- caller's queries (host copy), H2D, 65536 queries per piece, one BATCH ahead
//   compute stream   the engine, batch by batch (each batch waits for its last piece's H2D)
//   copy-out stream  D2H of a finished batch's ids (+ distances) into pinned slots while the next batch computes;
//                    the calling thread drains the slots into the caller's arrays
// Buffers the caller pinned itself (hipHostMalloc / hipHostRegister) skip the staging copies.  Overflowed queries
// (rare) are re-run at the end and their rows fetched again.
// `sink` (HVS_GATHER_PEER): finished pieces do not leave for the host but for GPU `sink->ctx`'s result buffer, rows
// sink->row0 + ..., over xGMI (hipMemcpyPeerAsync on this GPU's copy-out stream, under the next batch's compute); the
// root downloads the gathered block once every GPU is done.  out_ids / out_dists are then unused.
struct PeerSink {
    hvs_ctx* ctx;
    uint32_t row0;
    bool want_dists;
};

int leaf_query(hvs_ctx* c, const float* q_rows, uint32_t nq, float sample_proportion, uint32_t* out_ids, float* out_dists,
               const PeerSink* sink = nullptr)
{
    if (!c->d_data) return fail(c, HVS_ESTATE, "no data set loaded (hvs_load_data / hvs_gen_data)");
    if (nq == 0) return HVS_OK;
    if (!q_rows || (!out_ids && !sink)) return fail(c, HVS_EINVAL, "hvs_query: q_rows / out_ids is NULL");
    const auto t_host0 = std::chrono::steady_clock::now();
    HostTrace tr;
    int rc = leaf_begin_queries(c, nq);
    if (rc) return rc;
    tr.mark("begin");
    const bool in_pinned = host_pointer_is_pinned(q_rows);
    // (a peer sink behaves like a pinned destination: asynchronous copies straight from the result buffer, nothing to drain)
    const bool out_pinned = sink || (host_pointer_is_pinned(out_ids) && (!out_dists || host_pointer_is_pinned(out_dists)));
    if ((rc = ensure_staging(c, out_dists != nullptr && !sink, in_pinned ? 0u : nq, out_pinned ? 0u : nq))) return rc;
    tr.mark("staging");
    c->nq = nq;
    const bool sink_dists = sink && sink->want_dists;
    constexpr uint32_t SQ = hvs_ctx::kStageQ;
    constexpr int R = hvs_ctx::kRing;
    const uint32_t npieces = hvs_ceil_div(nq, SQ);

    // --- copy-in: pieces [p0, p1) -> device
    uint32_t in_next = 0;  // pieces enqueued so far
    auto stage_in_until = [&](uint32_t p1) -> int {
        for (; in_next < p1; ++in_next) {
            const uint32_t q0 = in_next * SQ, m = std::min(SQ, nq - q0);
            const size_t bytes = (size_t)m * HVS_QCOLS * sizeof(float);
            float* dst = c->d_q + (size_t)q0 * HVS_QCOLS;
            if (in_pinned) {
                HVS_HIP(c, hipMemcpyAsync(dst, q_rows + (size_t)q0 * HVS_QCOLS, bytes, hipMemcpyHostToDevice, c->s_in));
            } else {
                const int k = (int)(in_next % R);
                if (in_next >= (uint32_t)R) HVS_HIP(c, hipEventSynchronize(c->ev_in[k]));
                staging_copy(c->h_in[k], q_rows + (size_t)q0 * HVS_QCOLS, bytes);
                HVS_HIP(c, hipMemcpyAsync(dst, c->h_in[k], bytes, hipMemcpyHostToDevice, c->s_in));
                HVS_HIP(c, hipEventRecord(c->ev_in[k], c->s_in));
            }
        }
        return HVS_OK;
    };
    // --- copy-out of queries [q0, q0 + m): D2H pieces on s_out (after `ready`), drained into the caller's arrays
    uint32_t out_enq = 0, out_drained = 0;  // pieces (global numbering over the call)
    auto drain_one = [&]() -> int {
        const int k = (int)(out_drained % R);
        HVS_HIP(c, hipEventSynchronize(c->ev_out[k]));
        const uint32_t q0 = out_drained * SQ, m = std::min(SQ, nq - q0);
        staging_copy(out_ids + (size_t)q0 * c->k, c->h_out_ids[k], (size_t)m * c->k * sizeof(uint32_t));
        if (out_dists) staging_copy(out_dists + (size_t)q0 * c->k, c->h_out_dists[k], (size_t)m * c->k * sizeof(float));
        ++out_drained;
        return HVS_OK;
    };
    auto copy_out_until = [&](uint32_t p1) -> int {
        for (; out_enq < p1; ++out_enq) {
            const uint32_t q0 = out_enq * SQ, m = std::min(SQ, nq - q0);
            const size_t nb = (size_t)m * c->k * sizeof(uint32_t);
            if (sink) {
                const size_t dst = ((size_t)sink->row0 + q0) * c->k;
                HVS_HIP(c, hipMemcpyPeerAsync(sink->ctx->d_out_ids + dst, sink->ctx->device, c->d_out_ids + (size_t)q0 * c->k, c->device, nb, c->s_out));
                if (sink_dists)
                    HVS_HIP(c, hipMemcpyPeerAsync(sink->ctx->d_out_dists + dst, sink->ctx->device, c->d_out_dists + (size_t)q0 * c->k, c->device, nb,
                                                  c->s_out));
                continue;
            }
            if (out_pinned) {
                HVS_HIP(c, hipMemcpyAsync(out_ids + (size_t)q0 * c->k, c->d_out_ids + (size_t)q0 * c->k, nb, hipMemcpyDeviceToHost, c->s_out));
                if (out_dists)
                    HVS_HIP(c, hipMemcpyAsync(out_dists + (size_t)q0 * c->k, c->d_out_dists + (size_t)q0 * c->k, nb, hipMemcpyDeviceToHost, c->s_out));
                continue;
            }
            if (out_enq - out_drained >= (uint32_t)R) {
                int rc2 = drain_one();
                if (rc2) return rc2;
            }
            const int k = (int)(out_enq % R);
            HVS_HIP(c, hipMemcpyAsync(c->h_out_ids[k], c->d_out_ids + (size_t)q0 * c->k, nb, hipMemcpyDeviceToHost, c->s_out));
            if (out_dists)
                HVS_HIP(c, hipMemcpyAsync(c->h_out_dists[k], c->d_out_dists + (size_t)q0 * c->k, nb, hipMemcpyDeviceToHost, c->s_out));
            HVS_HIP(c, hipEventRecord(c->ev_out[k], c->s_out));
        }
        return HVS_OK;
    };

    // The engine runs batch by batch on run_queries' own schedule.  Before the kernels of batch b are enqueued the
    // compute stream is made to wait for the batch's queries (staged here if the previous batch's hook has not done so:
    // the first batch, or a schedule that changed) -- a batch never starts before its input is on the device.  After
    // they have been enqueued (the GPU is busy with them): batch b+1's queries are staged and sent, the results of
    // everything BEFORE batch b are sent out and drained, and the copy-out stream is told to wait for batch b.  The
    // host work of a batch (two staging copies, ~0.2 s per 2^20 queries) hides under the batch's own compute.
    uint32_t staged_q = 0;  // queries [0, staged_q) are enqueued on the copy-in stream and the compute stream waits for them
    auto send_input = [&](uint32_t upto_q, bool wait_here) -> int {
        upto_q = std::min(nq, upto_q);
        if (upto_q <= staged_q) {
            // staged by the previous batch's hook, which made ITS lane's stream wait: this batch's stream waits itself
            if (wait_here && staged_q) HVS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_stage, 0));
            return HVS_OK;
        }
        int r2 = stage_in_until(std::min(npieces, hvs_ceil_div(upto_q, SQ)));
        if (r2) return r2;
        HVS_HIP(c, hipEventRecord(c->ev_stage, c->s_in));
        if (wait_here) HVS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_stage, 0));
        if (staged_q == 0u) tr.mark("in0");
        staged_q = std::min(nq, in_next * SQ);
        return HVS_OK;
    };
    struct Hooks {
        decltype(send_input)& send;
        decltype(copy_out_until)& copy_out;
        hvs_ctx* c;
        uint32_t nq;
        int before(uint32_t off, uint32_t nqb) const { return send(off + nqb, true); }
        int after(uint32_t off, uint32_t nqb, uint32_t next_nqb) const
        {
            int r2;
            if (next_nqb && (r2 = send(off + nqb + next_nqb, false))) return r2;
            if ((r2 = copy_out(off / hvs_ctx::kStageQ))) return r2;  // whole pieces of the batches before this one
            HVS_HIP(c, hipEventRecord(c->ev_batch, c->stream));
            HVS_HIP(c, hipStreamWaitEvent(c->s_out, c->ev_batch, 0));
            return HVS_OK;
        }
    };
    const Hooks hooks{send_input, copy_out_until, c, nq};
    tr.mark("pointers");
    if ((rc = run_queries(c, 0, nq, sample_proportion, hooks, true))) return rc;
    tr.mark("enqueued");
    if ((rc = copy_out_until(npieces))) return rc;
    while (!out_pinned && out_drained < out_enq)
        if ((rc = drain_one())) return rc;
    tr.mark("drained");
    HVS_HIP(c, hipStreamSynchronize(c->s_out));
    tr.mark("s_out");
    // overflowed queries: re-run by the exact engine, their rows fetched again
    const bool had_ovf = c->ovf_pending;
    if ((rc = resolve_overflow(c))) return rc;
    tr.mark("resolved");
    if (had_ovf && (c->fallback_queries || c->retry_queries || c->demoted_queries)) {
        const uint32_t novf = c->fallback_queries, nretry = c->retry_queries, ndem = c->demoted_queries;
        std::vector<uint32_t> list((size_t)novf + nretry + ndem);
        if (novf) HVS_HIP(c, hipMemcpy(list.data(), c->d_ovf_list, (size_t)novf * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (nretry)
            HVS_HIP(c, hipMemcpy(list.data() + novf, c->d_retry_list, (size_t)nretry * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (ndem)
            HVS_HIP(c, hipMemcpy(list.data() + novf + nretry, c->d_demote_list, (size_t)ndem * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (sink) {
            for (uint32_t qi : list) {
                const size_t dst = ((size_t)sink->row0 + qi) * c->k;
                HVS_HIP(c, hipMemcpyPeerAsync(sink->ctx->d_out_ids + dst, sink->ctx->device, c->d_out_ids + (size_t)qi * c->k, c->device,
                                              c->k * sizeof(uint32_t), c->stream));
                if (sink_dists)
                    HVS_HIP(c, hipMemcpyPeerAsync(sink->ctx->d_out_dists + dst, sink->ctx->device, c->d_out_dists + (size_t)qi * c->k,
                                                  c->device, c->k * sizeof(float), c->stream));
            }
            HVS_HIP(c, hipStreamSynchronize(c->stream));
        } else {
            // the re-run rows are packed on the device (the idle candidate workspace is the scratch), leave in ONE copy and
            // are scattered here: thousands of 400-byte copies cost ~10 us each
            const size_t cnt = list.size(), words = cnt * c->k;
            uint32_t* scratch = reinterpret_cast<uint32_t*>(c->fb.cand);
            const bool fits = scratch && 2u * words * sizeof(uint32_t) <= c->fb_cand_entries * sizeof(uint64_t);
            if (fits) {
                uint32_t* d_list = scratch + 2u * words;  // (the list itself rides behind the rows when there is room)
                const bool list_fits = (2u * words + cnt) * sizeof(uint32_t) <= c->fb_cand_entries * sizeof(uint64_t);
                uint32_t* d_list_tmp = nullptr;
                if (!list_fits) {
                    HVS_HIP(c, hipMalloc(reinterpret_cast<void**>(&d_list_tmp), cnt * sizeof(uint32_t)));
                    d_list = d_list_tmp;
                }
                std::vector<uint32_t> packed(words * (out_dists ? 2u : 1u));
                hipError_t e = hipMemcpyAsync(d_list, list.data(), cnt * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(hvs_k_gather_rows, dim3(hvs_ceil_div((uint32_t)words, 256u)), dim3(256), 0, c->stream, d_list, (uint32_t)cnt,
                                       c->d_out_ids, c->k, scratch);
                    if (out_dists)
                        hipLaunchKernelGGL(hvs_k_gather_rows, dim3(hvs_ceil_div((uint32_t)words, 256u)), dim3(256), 0, c->stream, d_list,
                                           (uint32_t)cnt, reinterpret_cast<const uint32_t*>(c->d_out_dists), c->k, scratch + words);
                    e = hipMemcpyAsync(packed.data(), scratch, packed.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
                }
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (d_list_tmp) (void)hipFree(d_list_tmp);
                if (e != hipSuccess) return fail(c, HVS_EHIP, std::string("hvs_query: fetching the re-run rows: ") + hipGetErrorString(e));
                for (size_t i = 0; i < cnt; ++i) {
                    std::memcpy(out_ids + (size_t)list[i] * c->k, packed.data() + i * c->k, c->k * sizeof(uint32_t));
                    if (out_dists) std::memcpy(out_dists + (size_t)list[i] * c->k, packed.data() + words + i * c->k, c->k * sizeof(float));
                }
            } else {
                for (uint32_t qi : list) {
                    HVS_HIP(c, hipMemcpyAsync(out_ids + (size_t)qi * c->k, c->d_out_ids + (size_t)qi * c->k, c->k * sizeof(uint32_t),
                                              hipMemcpyDeviceToHost, c->stream));
                    if (out_dists)
                        HVS_HIP(c, hipMemcpyAsync(out_dists + (size_t)qi * c->k, c->d_out_dists + (size_t)qi * c->k,
                                                  c->k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
                }
                HVS_HIP(c, hipStreamSynchronize(c->stream));
            }
        }
    }
    c->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    tr.mark("done");
    tr.flush("hvs_query", nq);
    return HVS_OK;
}

int leaf_last_timing(hvs_ctx* c, hvs_timing* out)
{
    if (!c->timing_valid) return fail(c, HVS_ESTATE, "no query has run yet");
    int rc = leaf_sync(c);
    if (rc) return rc;
    HVS_HIP(c, hipEventSynchronize(c->ev_q1));
    float ms = 0.f;
    HVS_HIP(c, hipEventElapsedTime(&ms, c->ev_q0, c->ev_q1));
    c->timing.query_ms = ms;
    double k = 0.0;
    for (int i = 0; i < c->n_launch_events; ++i) {
        HVS_HIP(c, hipEventElapsedTime(&ms, c->ev_k[2 * i], c->ev_k[2 * i + 1]));
        k += ms;
    }
    c->timing.main_kernel_ms = k;
    c->timing.main_kernel_launches = (uint32_t)c->n_launch_events;
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HVS_HIP(c, hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    c->timing.pairs = h[0];
    c->timing.scanned_pairs = h[1];
    c->timing.rescored_pairs = h[2];
    c->timing.fallback_queries = c->fallback_queries;
    c->timing.retry_queries = c->retry_queries;
    c->timing.untimed_launches = c->untimed_launches;
    c->timing.host_ms = c->host_ms;
    *out = c->timing;
    return HVS_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-GPU root
// ---------------------------------------------------------------------------------------------
// contiguous, balanced range of part r of `world` (the first total % world parts get one more): sharding.shard_range
void shard_range(uint32_t total, uint32_t r, uint32_t world, uint32_t& a, uint32_t& b)
{
    const uint32_t base = total / world, rem = total % world;
    a = r * base + std::min(r, rem);
    b = a + base + (r < rem ? 1u : 0u);
}

// CPUs of the NUMA node of a GPU: PCI bus id -> /sys/bus/pci/devices/<id>/numa_node -> /sys/devices/system/node/node<k>/cpulist
// (best effort: empty when the platform does not say)
std::vector<int> device_node_cpus(int device)
{
    std::vector<int> cpus;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) {
        (void)hipGetLastError();
        return cpus;
    }
    for (char* p = bus; *p; ++p) *p = (char)std::tolower((unsigned char)*p);
    int node = -1;
    if (FILE* f = std::fopen((std::string("/sys/bus/pci/devices/") + bus + "/numa_node").c_str(), "r")) {
        if (std::fscanf(f, "%d", &node) != 1) node = -1;
        std::fclose(f);
    }
    if (node < 0) return cpus;
    if (FILE* f = std::fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r")) {
        int a = 0, b = 0;
        for (;;) {  // "0-15,128-143"
            if (std::fscanf(f, "%d", &a) != 1) break;
            b = a;
            int ch = std::fgetc(f);
            if (ch == '-') {
                if (std::fscanf(f, "%d", &b) != 1) break;
                ch = std::fgetc(f);
            }
            for (int x = a; x <= b && x < CPU_SETSIZE; ++x) cpus.push_back(x);
            if (ch != ',') break;
        }
        std::fclose(f);
    }
    return cpus;
}

// the calling thread (one leaf's host thread: staging copies, launches) moves next to its GPU; helper threads it starts
// inherit the mask
void pin_to_node(const hvs_ctx* leaf)
{
    if (leaf->node_cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int x : leaf->node_cpus) CPU_SET(x, &set);
    (void)sched_setaffinity(0, sizeof(set), &set);
}

// run fn(leaf index) on one host thread per leaf (the reference's vec_query owns its worker threads the same way,
// optimized_parallel.hpp:82-89, threading.hpp:100-141) and return the first failure
template <typename Fn>
int for_each_leaf(hvs_ctx* root, Fn fn)
{
    const size_t N = root->kids.size();
    std::vector<int> rcs(N, HVS_OK);
    if (N == 1) {
        rcs[0] = fn(0u);
    } else {
        std::vector<std::thread> th;
        th.reserve(N);
        for (size_t r = 0; r < N; ++r)
            th.emplace_back([&, r]() {
                pin_to_node(root->kids[r]);  // (threads of this call only: the caller's own thread keeps its affinity)
                rcs[r] = fn((uint32_t)r);
            });
        for (auto& t : th) t.join();
    }
    for (size_t r = 0; r < N; ++r)
        if (rcs[r] != HVS_OK) {
            root->err = "GPU " + std::to_string(root->kids[r]->device) + " (part " + std::to_string(r) + "): " + root->kids[r]->err;
            return rcs[r];
        }
    return HVS_OK;
}

// leaves that hold part of the resident range [q0, q0 + nq): fn(leaf, local q0, count, offset into the caller's range)
template <typename Fn>
int for_each_resident_part(hvs_ctx* root, uint32_t q0, uint32_t nq, Fn fn)
{
    if (root->kid_q0.size() != root->kids.size() + 1u || (uint64_t)q0 + nq > root->kid_q0.back())
        return fail(root, HVS_EINVAL, "query range outside the resident query set");
    return for_each_leaf(root, [&](uint32_t r) -> int {
        const uint32_t a = std::max(q0, root->kid_q0[r]), b = std::min(q0 + nq, root->kid_q0[r + 1]);
        if (a >= b) return HVS_OK;
        return fn(root->kids[r], a - root->kid_q0[r], b - a, a - q0);
    });
}

}  // namespace

extern "C" {

const char* hvs_version(void) { return "hvs-mi355x 0.4 (gfx950)"; }

uint32_t hvs_plan_guess_m(uint32_t k, double seen_fraction, uint32_t pfail)
{
    return guess_m(seen_fraction, k, std::pow(10.0, -(double)pfail));
}

uint32_t hvs_plan_batches(uint32_t nq, int host_pipeline, uint32_t* out, uint32_t cap)
{
    const std::vector<uint32_t> sched = batch_schedule(nq, kBatchMfma, host_pipeline != 0);
    for (size_t i = 0; out && i < sched.size() && i < cap; ++i) out[i] = sched[i];
    return (uint32_t)sched.size();
}

const char* hvs_last_global_error(void) { return g_global_err.c_str(); }

int hvs_create(hvs_ctx** out, int device)
{
    if (!out) {
        g_global_err = "hvs_create: out is NULL";
        return HVS_EINVAL;
    }
    return leaf_create(out, device);
}

int hvs_create_on_devices(hvs_ctx** out, const int* devices, int n)
{
    if (!out || !devices || n < 1 || n > 64) {
        g_global_err = "hvs_create_on_devices: bad argument (1..64 device indices)";
        if (out) *out = nullptr;
        return HVS_EINVAL;
    }
    *out = nullptr;
    hvs_ctx* root = new (std::nothrow) hvs_ctx();
    if (!root) {
        g_global_err = "hvs_create_on_devices: out of host memory";
        return HVS_ENOMEM;
    }
    root->device = devices[0];
    for (int i = 0; i < n; ++i) {
        hvs_ctx* leaf = nullptr;
        const int rc = leaf_create(&leaf, devices[i]);
        if (rc != HVS_OK) {
            hvs_destroy(root);
            return rc;
        }
        if (n > 1 && env_u32("HVS_NUMA_PIN", 1u, 0u, 1u)) leaf->node_cpus = device_node_cpus(devices[i]);
        root->kids.push_back(leaf);
    }
    // peer access lets the replication of D and the peer gather use xGMI directly (best effort: without it the
    // runtime stages peer copies through host memory)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (devices[i] != devices[j]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) {
                    (void)hipSetDevice(devices[i]);
                    (void)hipDeviceEnablePeerAccess(devices[j], 0);
                }
                (void)hipGetLastError();  // "already enabled" is fine
            }
    (void)hipSetDevice(devices[0]);
    *out = root;
    return HVS_OK;
}

int hvs_create_multi(hvs_ctx** out, int n_gpus)
{
    if (!out) {
        g_global_err = "hvs_create_multi: out is NULL";
        return HVS_EINVAL;
    }
    *out = nullptr;
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        g_global_err = std::string("hvs_create_multi: no HIP device available (") + hipGetErrorString(e) +
                       "); this library has no CPU fallback";
        return HVS_EHIP;
    }
    if (n_gpus < 0 || n_gpus > count) {
        g_global_err = "hvs_create_multi: n_gpus outside 0..visible devices";
        return HVS_EINVAL;
    }
    if (n_gpus == 0) n_gpus = count;
    std::vector<int> devs(n_gpus);
    for (int i = 0; i < n_gpus; ++i) devs[i] = i;
    return hvs_create_on_devices(out, devs.data(), n_gpus);
}

int hvs_num_gpus(const hvs_ctx* c) { return !c ? 0 : (c->kids.empty() ? 1 : (int)c->kids.size()); }

int hvs_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

int hvs_set_gather(hvs_ctx* c, int mode)
{
    if (!c) return HVS_EINVAL;
    if (mode != HVS_GATHER_DIRECT && mode != HVS_GATHER_PEER) return fail(c, HVS_EINVAL, "hvs_set_gather: unknown mode");
    c->gather_mode = mode;
    return HVS_OK;
}

void hvs_destroy(hvs_ctx* c)
{
    if (!c) return;
    for (hvs_ctx* k : c->kids) hvs_destroy(k);
    if (c->kids.empty()) leaf_destroy(c);
    delete c;
}

const char* hvs_last_error(const hvs_ctx* c) { return c ? c->err.c_str() : "hvs: NULL context"; }

int hvs_set_engine(hvs_ctx* c, int engine)
{
    if (!c) return HVS_EINVAL;
    if (engine != HVS_ENGINE_AUTO && engine != HVS_ENGINE_EXACT_SCAN && engine != HVS_ENGINE_MFMA_FILTER &&
        engine != HVS_ENGINE_MFMA_I8 && engine != HVS_ENGINE_MFMA_F16)
        return fail(c, HVS_EINVAL, "hvs_set_engine: unknown engine");
    c->engine = engine;
    if (!c->kids.empty()) return for_each_leaf(c, [&](uint32_t r) { return hvs_set_engine(c->kids[r], engine); });
    if (c->d_data && !c->have_index &&
        (engine == HVS_ENGINE_MFMA_FILTER || engine == HVS_ENGINE_MFMA_I8 || engine == HVS_ENGINE_MFMA_F16 || c->n >= kIndexMinRows)) {
        HVS_HIP(c, hipSetDevice(c->device));
        return build_index(c);
    }
    return HVS_OK;
}

int hvs_set_padding(hvs_ctx* c, int enabled)
{
    if (!c) return HVS_EINVAL;
    c->padding = enabled != 0;
    for (hvs_ctx* k : c->kids) k->padding = c->padding;
    return HVS_OK;
}

int hvs_set_distance_order(hvs_ctx* c, int order)
{
    if (!c) return HVS_EINVAL;
    if (order != HVS_ORDER_SIMD && order != HVS_ORDER_SCALAR) return fail(c, HVS_EINVAL, "hvs_set_distance_order: unknown order");
    c->scalar_order = order == HVS_ORDER_SCALAR;
    for (hvs_ctx* k : c->kids) k->scalar_order = c->scalar_order;
    return HVS_OK;
}

int hvs_set_k(hvs_ctx* c, uint32_t k)
{
    if (!c) return HVS_EINVAL;
    if (k < 8u || k > HVS_KMAX) return fail(c, HVS_EINVAL, "hvs_set_k: k outside 8..256 (the reference asserts KNN_LIMIT >= 8)");
    if (!c->kids.empty()) {
        const int rc = for_each_leaf(c, [&](uint32_t r) { return hvs_set_k(c->kids[r], k); });
        if (!rc) c->k = k;
        return rc;
    }
    if (c->n && c->n < k) return fail(c, HVS_EINVAL, "hvs_set_k: the loaded data set has fewer than k rows");
    if (k == c->k) return HVS_OK;
    HVS_HIP(c, hipSetDevice(c->device));
    int rc = resolve_overflow(c);
    if (rc) return rc;
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    c->k = k;
    c->cap = k <= 128u ? 256 : 512;
    // result rows change size: resident queries stay, their results do not; list workspaces are re-sized on demand
    const uint32_t had = c->res_cap;
    c->res_cap = 0;
    c->cand_lists = 0;
    c->timing_valid = false;
    return had ? ensure_results(c, had) : HVS_OK;
}

uint32_t hvs_get_k(const hvs_ctx* c) { return c ? c->k : 0u; }

uint32_t hvs_num_rows(const hvs_ctx* c) { return !c ? 0u : (c->kids.empty() ? c->n : c->kids[0]->n); }

int hvs_reserve(hvs_ctx* c, uint32_t nq)
{
    if (!c) return HVS_EINVAL;
    if (c->kids.empty()) return leaf_reserve(c, nq);
    const uint32_t N = (uint32_t)c->kids.size();
    return for_each_leaf(c, [&](uint32_t r) { return leaf_reserve(c->kids[r], hvs_ceil_div(nq, N)); });
}

int hvs_load_data(hvs_ctx* c, const float* rows, uint32_t n)
{
    if (!c) return HVS_EINVAL;
    if (!rows) return fail(c, HVS_EINVAL, "hvs_load_data: rows is NULL");
    if (c->kids.empty()) return leaf_load_data(c, rows, n);
    // D reaches the GPUs in two parallel phases: every GPU uploads ITS slice of the rows over its own PCIe link, then takes
    // the other slices from its peers over xGMI (each pair has its own link) -- an all-gather by peer copies -- and builds
    // its index.  (Round 2 uploaded everything to GPU 0 and let 7 peers pull 4 GB each from it after its index build;
    // HVS_LOAD_GATHER=0 keeps one upload + peer copies for A/B runs.)
    const uint32_t N = (uint32_t)c->kids.size();
    if (N == 1u || !env_u32("HVS_LOAD_GATHER", 1u, 0u, 1u)) {
        int rc = leaf_upload_data(c->kids[0], rows, n);
        if (rc) return fail(c, rc, c->kids[0]->err);
        return for_each_leaf(c, [&](uint32_t r) {
            if (r == 0u) {
                HVS_HIP(c->kids[0], hipSetDevice(c->kids[0]->device));
                return finish_data(c->kids[0]);
            }
            return leaf_load_data_from_peer(c->kids[r], c->kids[0]);
        });
    }
    std::vector<uint32_t> r0(N + 1u, 0u);
    for (uint32_t r = 0; r < N; ++r) shard_range(n, r, N, r0[r], r0[r + 1]);
    int rc = for_each_leaf(c, [&](uint32_t r) -> int {  // phase 1: own slice, host -> device
        hvs_ctx* k = c->kids[r];
        int r2 = begin_data(k, n);
        if (r2) return r2;
        HVS_HIP(k, hipEventRecord(k->ev_q0, k->stream));
        const size_t off = (size_t)r0[r] * HVS_DCOLS, cnt = (size_t)(r0[r + 1] - r0[r]) * HVS_DCOLS;
        if ((r2 = upload_rows(k, k->d_data + off, rows + off, cnt))) return r2;
        HVS_HIP(k, hipStreamSynchronize(k->stream));
        return HVS_OK;
    });
    if (rc) return rc;
    return for_each_leaf(c, [&](uint32_t r) -> int {  // phase 2: the other slices, device -> device; then the index
        hvs_ctx* k = c->kids[r];
        HVS_HIP(k, hipSetDevice(k->device));
        for (uint32_t step = 1; step < N; ++step) {  // (rank r starts with its right neighbour: the pairs of a step are disjoint)
            const uint32_t p = (r + step) % N;
            const size_t off = (size_t)r0[p] * HVS_DCOLS, cnt = (size_t)(r0[p + 1] - r0[p]) * HVS_DCOLS;
            if (cnt)
                HVS_HIP(k, hipMemcpyPeerAsync(k->d_data + off, k->device, c->kids[p]->d_data + off, c->kids[p]->device, cnt * sizeof(float),
                                              k->stream));
        }
        HVS_HIP(k, hipEventRecord(k->ev_q1, k->stream));
        HVS_HIP(k, hipStreamSynchronize(k->stream));
        k->n = n;
        return finish_data(k);
    });
}

int hvs_gen_data(hvs_ctx* c, uint32_t n, uint64_t seed, int profile, uint32_t ncat)
{
    if (!c) return HVS_EINVAL;
    if (ncat == 0) return fail(c, HVS_EINVAL, "hvs_gen_data: ncat must be > 0");
    if (c->kids.empty()) return leaf_gen_data(c, n, seed, profile, ncat);
    return for_each_leaf(c, [&](uint32_t r) { return leaf_gen_data(c->kids[r], n, seed, profile, ncat); });
}

int hvs_download_data(hvs_ctx* c, uint32_t row0, uint32_t nrows, float* out_rows)
{
    if (!c) return HVS_EINVAL;
    if (!c->kids.empty()) {
        const int rc = hvs_download_data(c->kids[0], row0, nrows, out_rows);
        return rc ? fail(c, rc, c->kids[0]->err) : HVS_OK;
    }
    if (!c->d_data) return fail(c, HVS_ESTATE, "no data set loaded");
    if (!out_rows || (uint64_t)row0 + nrows > c->n) return fail(c, HVS_EINVAL, "hvs_download_data: bad range");
    HVS_HIP(c, hipSetDevice(c->device));
    HVS_HIP(c, hipMemcpyAsync(out_rows, c->d_data + (size_t)row0 * HVS_DCOLS, (size_t)nrows * HVS_DCOLS * sizeof(float),
                              hipMemcpyDeviceToHost, c->stream));
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    return HVS_OK;
}

int hvs_upload_queries(hvs_ctx* c, const float* q_rows, uint32_t nq)
{
    if (!c) return HVS_EINVAL;
    if (!q_rows && nq) return fail(c, HVS_EINVAL, "hvs_upload_queries: q_rows is NULL");
    if (c->kids.empty()) return leaf_upload_queries(c, q_rows, nq);
    const uint32_t N = (uint32_t)c->kids.size();
    c->kid_q0.assign(N + 1u, 0u);
    for (uint32_t r = 0; r < N; ++r) shard_range(nq, r, N, c->kid_q0[r], c->kid_q0[r + 1]);
    return for_each_leaf(c, [&](uint32_t r) {
        return leaf_upload_queries(c->kids[r], q_rows + (size_t)c->kid_q0[r] * HVS_QCOLS, c->kid_q0[r + 1] - c->kid_q0[r]);
    });
}

int hvs_gen_queries(hvs_ctx* c, uint32_t nq, uint64_t seed, int profile, uint32_t ncat, int force_type,
                    uint64_t first_row)
{
    if (!c) return HVS_EINVAL;
    if (ncat == 0 || force_type > 3) return fail(c, HVS_EINVAL, "hvs_gen_queries: bad ncat / force_type");
    if (c->kids.empty()) return leaf_gen_queries(c, nq, seed, profile, ncat, force_type, first_row);
    const uint32_t N = (uint32_t)c->kids.size();
    c->kid_q0.assign(N + 1u, 0u);
    for (uint32_t r = 0; r < N; ++r) shard_range(nq, r, N, c->kid_q0[r], c->kid_q0[r + 1]);
    return for_each_leaf(c, [&](uint32_t r) {
        return leaf_gen_queries(c->kids[r], c->kid_q0[r + 1] - c->kid_q0[r], seed, profile, ncat, force_type, first_row + c->kid_q0[r]);
    });
}

int hvs_download_queries(hvs_ctx* c, uint32_t q0, uint32_t nq, float* out_rows)
{
    if (!c) return HVS_EINVAL;
    if (!c->kids.empty())
        return for_each_resident_part(c, q0, nq, [&](hvs_ctx* k, uint32_t lq0, uint32_t m, uint32_t off) {
            return hvs_download_queries(k, lq0, m, out_rows + (size_t)off * HVS_QCOLS);
        });
    if (!out_rows || (uint64_t)q0 + nq > c->nq) return fail(c, HVS_EINVAL, "hvs_download_queries: bad range");
    HVS_HIP(c, hipSetDevice(c->device));
    HVS_HIP(c, hipMemcpyAsync(out_rows, c->d_q + (size_t)q0 * HVS_QCOLS, (size_t)nq * HVS_QCOLS * sizeof(float),
                              hipMemcpyDeviceToHost, c->stream));
    HVS_HIP(c, hipStreamSynchronize(c->stream));
    return HVS_OK;
}

int hvs_query_resident(hvs_ctx* c, uint32_t q0, uint32_t nq, float sample_proportion)
{
    if (!c) return HVS_EINVAL;
    if (!c->kids.empty())
        return for_each_resident_part(c, q0, nq, [&](hvs_ctx* k, uint32_t lq0, uint32_t m, uint32_t) {
            return run_queries(k, lq0, m, sample_proportion, NoHook{});
        });
    return run_queries(c, q0, nq, sample_proportion, NoHook{});
}

int hvs_sync(hvs_ctx* c)
{
    if (!c) return HVS_EINVAL;
    if (!c->kids.empty()) return for_each_leaf(c, [&](uint32_t r) { return leaf_sync(c->kids[r]); });
    return leaf_sync(c);
}

int hvs_download_results(hvs_ctx* c, uint32_t q0, uint32_t nq, uint32_t* out_ids, float* out_dists)
{
    if (!c) return HVS_EINVAL;
    if (!c->kids.empty()) {
        if (!out_ids) return fail(c, HVS_EINVAL, "hvs_download_results: out_ids is NULL");
        return for_each_resident_part(c, q0, nq, [&](hvs_ctx* k, uint32_t lq0, uint32_t m, uint32_t off) {
            return leaf_download_results(k, lq0, m, out_ids + (size_t)off * c->k, out_dists ? out_dists + (size_t)off * c->k : nullptr);
        });
    }
    return leaf_download_results(c, q0, nq, out_ids, out_dists);
}

int hvs_export_results_device(hvs_ctx* c, uint32_t q0, uint32_t nq, uint32_t* d_ids, float* d_dists)
{
    if (!c) return HVS_EINVAL;
    if (!c->kids.empty()) return fail(c, HVS_EINVAL, "hvs_export_results_device: single-GPU contexts only (device pointers belong to one GPU)");
    if (!d_ids || (uint64_t)q0 + nq > c->nq) return fail(c, HVS_EINVAL, "hvs_export_results_device: bad range");
    HVS_HIP(c, hipSetDevice(c->device));
    int rc = resolve_overflow(c);
    if (rc) return rc;
    HVS_HIP(c, hipMemcpyAsync(d_ids, c->d_out_ids + (size_t)q0 * c->k, (size_t)nq * c->k * sizeof(uint32_t),
                              hipMemcpyDeviceToDevice, c->stream));
    if (d_dists)
        HVS_HIP(c, hipMemcpyAsync(d_dists, c->d_out_dists + (size_t)q0 * c->k,
                                  (size_t)nq * c->k * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return HVS_OK;
}

int hvs_stream_wait(hvs_ctx* c, void* stream)
{
    if (!c) return HVS_EINVAL;
    if (!c->kids.empty()) return fail(c, HVS_EINVAL, "hvs_stream_wait: single-GPU contexts only (a stream belongs to one GPU)");
    HVS_HIP(c, hipSetDevice(c->device));
    int rc = resolve_overflow(c);
    if (rc) return rc;
    HVS_HIP(c, hipEventRecord(c->ev_batch, c->stream));
    HVS_HIP(c, hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->ev_batch, 0));
    return HVS_OK;
}

int hvs_query(hvs_ctx* c, const float* q_rows, uint32_t nq, float sample_proportion, uint32_t* out_ids,
              float* out_dists)
{
    if (!c) return HVS_EINVAL;
    if (c->kids.empty()) return leaf_query(c, q_rows, nq, sample_proportion, out_ids, out_dists);
    if (nq == 0) return HVS_OK;
    if (!q_rows || !out_ids) return fail(c, HVS_EINVAL, "hvs_query: q_rows / out_ids is NULL");
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t N = (uint32_t)c->kids.size();
    c->kid_q0.assign(N + 1u, 0u);
    for (uint32_t r = 0; r < N; ++r) shard_range(nq, r, N, c->kid_q0[r], c->kid_q0[r + 1]);
    int rc;
    if (c->gather_mode == HVS_GATHER_DIRECT) {
        // zero-collective path: every GPU's pipeline reads its slice of the caller's queries and writes its slice of
        // the caller's result arrays
        rc = for_each_leaf(c, [&](uint32_t r) -> int {
            const uint32_t a = c->kid_q0[r], m = c->kid_q0[r + 1] - a;
            if (m == 0u) {
                c->kids[r]->timing_valid = false;
                return HVS_OK;
            }
            return leaf_query(c->kids[r], q_rows + (size_t)a * HVS_QCOLS, m, sample_proportion, out_ids + (size_t)a * c->k,
                              out_dists ? out_dists + (size_t)a * c->k : nullptr);
        });
    } else {
        // peer gather (A/B partner of the direct path): every GPU runs the same host pipeline on its slice of the caller's
        // queries, but its finished pieces travel GPU -> GPU 0 over xGMI under the next batch's compute (PeerSink) instead
        // of going to the host; the gathered block leaves GPU 0 in one D2H.  GPU 0's own slice starts at row 0 of its
        // result buffer, which holds everybody's rows.
        hvs_ctx* k0 = c->kids[0];
        (void)hipSetDevice(k0->device);
        rc = ensure_results(k0, nq);
        if (rc) rc = fail(c, rc, k0->err);
        if (!rc)
            rc = for_each_leaf(c, [&](uint32_t r) -> int {
                const uint32_t a = c->kid_q0[r], m = c->kid_q0[r + 1] - a;
                hvs_ctx* k = c->kids[r];
                if (m == 0u) {
                    k->timing_valid = false;
                    return HVS_OK;
                }
                if (r == 0u) {  // its results are already where the gather wants them: plain resident run of its slice
                    int r2 = leaf_upload_queries(k, q_rows, m);
                    if (r2) return r2;
                    if ((r2 = run_queries(k, 0, m, sample_proportion, NoHook{}))) return r2;
                    return leaf_sync(k);
                }
                const PeerSink sink{k0, a, out_dists != nullptr};
                return leaf_query(k, q_rows + (size_t)a * HVS_QCOLS, m, sample_proportion, nullptr, nullptr, &sink);
            });
        if (!rc) {
            (void)hipSetDevice(k0->device);
            if (hipMemcpyAsync(out_ids, k0->d_out_ids, (size_t)nq * c->k * sizeof(uint32_t), hipMemcpyDeviceToHost, k0->stream) != hipSuccess ||
                (out_dists && hipMemcpyAsync(out_dists, k0->d_out_dists, (size_t)nq * c->k * sizeof(float), hipMemcpyDeviceToHost,
                                             k0->stream) != hipSuccess) ||
                hipStreamSynchronize(k0->stream) != hipSuccess)
                rc = fail(c, HVS_EHIP, "hvs_query: download of the gathered results failed");
        }
    }
    c->host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int hvs_merge_shards_device(hvs_ctx* c, uint32_t nshards, uint32_t nq, const uint32_t* d_ids_all, const float* d_dists_all,
                            const uint64_t* shard_row0, uint32_t n_total, const float* d_pad_dists, uint32_t* d_out_ids,
                            float* d_out_dists)
{
    if (!c) return HVS_EINVAL;
    if (!c->kids.empty()) return fail(c, HVS_EINVAL, "hvs_merge_shards_device: single-GPU contexts only");
    if (!d_ids_all || !d_dists_all || !shard_row0 || !d_pad_dists || !d_out_ids || nshards == 0u || nshards > 16u ||
        n_total < c->k)
        return fail(c, HVS_EINVAL, "hvs_merge_shards_device: bad argument (1..16 shards, n_total >= k, non-NULL buffers)");
    if (nq == 0u) return HVS_OK;
    HVS_HIP(c, hipSetDevice(c->device));
    HvsShardRows rows{};
    for (uint32_t s = 0; s < nshards; ++s) {
        if (shard_row0[s] >= n_total) return fail(c, HVS_EINVAL, "hvs_merge_shards_device: shard row offset outside the data set");
        rows.row0[s] = shard_row0[s];
    }
    with_cap(c->cap, [&](auto CAPT) {
        hipLaunchKernelGGL((hvs_k_merge_shards<decltype(CAPT)::value>), dim3((nq + 3u) / 4u), dim3(256), 0, c->stream, d_ids_all, d_dists_all,
                           nshards, nq, rows, n_total, d_pad_dists, d_out_ids, d_out_dists, c->k);
    });
    HVS_HIP(c, hipGetLastError());
    return HVS_OK;
}

int hvs_last_reruns(hvs_ctx* c, int which, uint32_t* out_idx, uint32_t cap)
{
    if (!c) return HVS_EINVAL;
    if (!c->kids.empty()) return fail(c, HVS_EINVAL, "hvs_last_reruns: single-GPU contexts only");
    if (which != 0 && which != 1) return fail(c, HVS_EINVAL, "hvs_last_reruns: which is 0 (exact) or 1 (retry)");
    int rc = leaf_sync(c);
    if (rc) return rc;
    const uint32_t len = which ? c->retry_queries : c->fallback_queries;
    const uint32_t m = std::min(len, cap);
    if (m && out_idx)
        HVS_HIP(c, hipMemcpy(out_idx, which ? c->d_retry_list : c->d_ovf_list, (size_t)m * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return (int)len;
}

int hvs_last_timing(hvs_ctx* c, hvs_timing* out)
{
    if (!c || !out) return HVS_EINVAL;
    if (c->kids.empty()) return leaf_last_timing(c, out);
    // whole-job view: device time = the slowest GPU's, work counters summed over the GPUs that took part
    hvs_timing agg{};
    bool any = false;
    for (hvs_ctx* k : c->kids) {
        if (!k->timing_valid) continue;
        hvs_timing t{};
        const int rc = leaf_last_timing(k, &t);
        if (rc) return fail(c, rc, k->err);
        agg.query_ms = std::max(agg.query_ms, t.query_ms);
        agg.main_kernel_ms += t.main_kernel_ms;
        agg.main_kernel_launches += t.main_kernel_launches;
        agg.nq += t.nq;
        agg.pairs += t.pairs;
        agg.scanned_pairs += t.scanned_pairs;
        agg.rescored_pairs += t.rescored_pairs;
        agg.fallback_queries += t.fallback_queries;
        agg.retry_queries += t.retry_queries;
        agg.flags |= t.flags;
        agg.untimed_launches += t.untimed_launches;
        agg.load_ms = std::max(agg.load_ms, t.load_ms);
        agg.engine = t.engine;
        agg.n_gpus += 1;
        any = true;
    }
    if (!any) return fail(c, HVS_ESTATE, "no query has run yet");
    agg.host_ms = c->host_ms;
    *out = agg;
    return HVS_OK;
}

}  // extern "C"
