/*
 * hvs.h -- C ABI of the MI355X-native filtered brute-force k-NN engine (libhvs.so).
 *
 * Drop-in boundary for ONE path of atalantus/Project---Hybrid-Vector-Search-Queries:
 * the `vec_query` seam every engine header of the reference defines
 *     void vec_query(vector<vector<float>>& nodes, vector<vector<float>>& queries,
 *                    float sample_proportion, vector<vector<uint32_t>>& knn_results);
 * (reference include/optimized_parallel.hpp:61-62, include/optimized.hpp:54-55,
 * include/baseline.hpp:68-69; called once from src/test.cpp:85) together with the
 * binary formats of include/io.h (D rows = 102 f32 [C,T,x0..x99], Q rows = 104 f32
 * [type,v,l,r,x0..x99], output.bin = nq x 100 uint32 ids in ascending distance).
 *
 * std::vector cannot cross a C ABI, so the boundary takes flat row-major buffers; the
 * header-only shim include/hvs_vec_query.hpp restores the exact C++ signature.
 *
 * Conventions: plain pointers and sizes, no exceptions, int status (0 = ok, negative =
 * HVS_E*), one hvs_ctx is used from one thread at a time (the reference has a single
 * caller, src/test.cpp:85).  A context is one GPU (hvs_create) or all GPUs of the node
 * (hvs_create_multi); per GPU the engine runs on the context's own HIP stream, hvs_query's
 * host transfers on two more (copy-in, copy-out), and a multi-GPU context drives each GPU
 * from its own host thread for the duration of a call.  dim = 100 is a compile-time constant like the reference's VEC_DIM (include/optimized_impl.h:28);
 * k defaults to the reference's KNN_LIMIT = 100 (optimized_impl.h:26) and can be changed per context (hvs_set_k).
 */
#ifndef HVS_H
#define HVS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HVS_OK 0
#define HVS_EINVAL (-1)  /* bad argument (NULL, n < 100, nq range ...) */
#define HVS_ENOMEM (-2)  /* host or device allocation failed */
#define HVS_EHIP (-3)    /* a HIP runtime call failed; see hvs_last_error */
#define HVS_ESTATE (-4)  /* call order: no data / no queries / no results loaded */

typedef struct hvs_ctx hvs_ctx;

/* Engines.  HVS_ENGINE_AUTO picks the fastest exact engine for the loaded data. */
#define HVS_ENGINE_AUTO 0
#define HVS_ENGINE_EXACT_SCAN 1 /* FP32 exact-order scan of every candidate row (VALU)       */
#define HVS_ENGINE_MFMA_FILTER 2 /* BF16 MFMA bound filter + exact-order re-scoring (same answers) */
#define HVS_ENGINE_MFMA_I8 3     /* INT8 MFMA bound filter + exact-order re-scoring (same answers); falls back to
                                    a 16-bit float filter for data the INT8 format cannot bound */
#define HVS_ENGINE_MFMA_F16 4    /* FP16 MFMA bound filter + exact-order re-scoring (same answers): the BF16 filter's cost,
                                    an 8x tighter bound; what HVS_ENGINE_AUTO picks for clustered / low-dimensional data */

typedef struct hvs_timing {
    double query_ms;      /* whole vec_query-equivalent region on the device stream (HIP events)   */
    double main_kernel_ms;/* sum of the dominant kernel's launches inside that region               */
    uint32_t main_kernel_launches;
    uint32_t nq;          /* queries answered                                                       */
    uint64_t pairs;       /* sum over queries of rows in [0,sn) passing the predicate (P of SURVEY 8d) */
    uint64_t scanned_pairs;/* (query,row) pairs the dominant kernel actually evaluated              */
    double load_ms;       /* last hvs_load_data / hvs_gen_data: upload + index build                */
    uint32_t engine;      /* engine that ran                                                        */
    uint32_t fallback_queries; /* queries re-run by the exact scan after a filter overflow          */
    uint64_t rescored_pairs;   /* MFMA engine: (query,row) pairs handed to the exact re-scoring kernel */
    uint32_t n_gpus;      /* GPUs that took part (multi-GPU context: query_ms = the slowest GPU's, counters summed)   */
    uint32_t untimed_launches; /* launches of the dominant kernel that could not be timed (event creation failed):
                                  main_kernel_ms is then a lower bound and must not feed a roofline                    */
    double host_ms;       /* last hvs_query: wall time of the whole call, host memory in -> host memory out (the
                             reference's timing scope, src/test.cpp:82-88); 0 for the device-resident calls          */
    uint32_t retry_queries;    /* filter engines: queries whose guessed threshold failed its check and that were run
                                  again with a proven one (same answers either way)                                     */
    uint32_t flags;            /* HVS_TIMING_* bits                                                                     */
} hvs_timing;
#define HVS_TIMING_INDEX_TOO_LARGE 1u /* the data set has more than 2^29 rows per GPU: no filter index, exact engine only */
#define HVS_TIMING_I8_ROTATED 4u      /* the INT8 tiles that ran were cut from the rotated vectors (csrc/hvs_filter.h, HvsQuant): same answers */
#define HVS_TIMING_FORMAT_CHANGED 2u  /* HVS_ENGINE_AUTO: so many queries of this call had no usable INT8 bound (far outside the
                                         data's box) that the 16-bit float tiles were built in mid-call; later calls use them */

/* ---- lifetime ---------------------------------------------------------------------------- */

/* One GPU.  device < 0: use the calling thread's current HIP device. */
int hvs_create(hvs_ctx **out, int device);
/* All GPUs of the node behind ONE context -- what the reference's vec_query does with the cores of the machine
 * (optimized_parallel.hpp:73-89: it sizes and owns its worker pool itself).  n_gpus = 0: every visible GPU.  D is
 * replicated (one upload over PCIe, GPU-to-GPU copies over xGMI), the queries of a call are cut into one contiguous
 * range per GPU (optimized_parallel.hpp:91: iterations are independent), one host thread drives each GPU, and every GPU
 * writes its block of ids straight into its slice of the caller's out_ids -- no collective.  Every function below
 * accepts such a context unless it says "single-GPU contexts only". */
int hvs_create_multi(hvs_ctx **out, int n_gpus);
/* The same on an explicit device list; an index may repeat ("virtual ranks": several parts on one GPU -- how the
 * multi-GPU plan is tested on a one-GPU box). */
int hvs_create_on_devices(hvs_ctx **out, const int *devices, int n);
int hvs_num_gpus(const hvs_ctx *ctx);
/* GPUs visible to this process (0 when there is none: the library has no CPU fallback). */
int hvs_device_count(void);
/* How hvs_query of a multi-GPU context brings the ids home.  DIRECT (default): each GPU's pipeline D2H-copies into its
 * slice of the caller's array.  PEER: the blocks travel GPU -> GPU 0 over xGMI and leave in one D2H (A/B partner; a
 * process-per-GPU driver gathers with RCCL instead, bench.py / sharding.py). */
#define HVS_GATHER_DIRECT 0
#define HVS_GATHER_PEER 1
int hvs_set_gather(hvs_ctx *ctx, int mode);
/* Announce the size of the coming calls: query/result buffers and the per-batch workspace (~34 GB for batches of 2^21
 * queries) are allocated now (and again after a later hvs_load_data) instead of inside the first query. */
int hvs_reserve(hvs_ctx *ctx, uint32_t nq);
void hvs_destroy(hvs_ctx *ctx);
/* Message of the last failing call on this context ("" if none). Valid until the next call. */
const char *hvs_last_error(const hvs_ctx *ctx);
/* Library-level message for failures that have no context (hvs_create). */
const char *hvs_last_global_error(void);
int hvs_set_engine(hvs_ctx *ctx, int engine);
/* Summation order of the distances.  HVS_ORDER_SIMD (default) is the hot path's AVX2 order
 * (optimized_impl.h:96-125) used by optimized.hpp / optimized_parallel.hpp; HVS_ORDER_SCALAR is the
 * sequential order of the reference's baseline engine (baseline.hpp:53-64; BASELINE.json configs[0]),
 * answered by the exact engine.  The two orders give different f32 distances and, now and then,
 * different neighbours (reference optimized.hpp:34-42). */
#define HVS_ORDER_SIMD 0
#define HVS_ORDER_SCALAR 1
int hvs_set_distance_order(hvs_ctx *ctx, int order);
/* Neighbours per query: the reference's compile-time KNN_LIMIT (include/optimized_impl.h:26, static_assert >= 8),
 * a run-time property here.  8 <= k <= 256, default 100; every "100" in the layouts below reads "k" after the call
 * (out_ids / out_dists rows hold k entries, the data set needs n >= k rows, padding appends rows n-1, n-2, ... up to
 * k).  Results of earlier queries are dropped. */
int hvs_set_k(hvs_ctx *ctx, uint32_t k);
uint32_t hvs_get_k(const hvs_ctx *ctx);
/* Padding (default on) appends rows n-1, n-2, ... when fewer than 100 rows match
 * (optimized_parallel.hpp:149-157).  A context that holds only a SHARD of D (D-sharded multi-GPU mode,
 * sharding.py) turns it off: unmatched slots then carry id 0xFFFFFFFF / distance +inf and the merge of
 * the shards' partial answers applies the padding once, from the tail of the whole data set. */
int hvs_set_padding(hvs_ctx *ctx, int enabled);

/* ---- data set D (replaces `nodes`, reference src/test.cpp:71-73 + io.h:111-136) ---------- */

/* rows: host memory, n x 102 f32.  Requires n >= 100 (the reference's padding index n-s
 * underflows below that, optimized.hpp:125).  Uploads and builds the device-side layout. */
int hvs_load_data(hvs_ctx *ctx, const float *rows, uint32_t n);
/* Generate gen-v1 rows (include/hvs_gen.h) directly in HBM. */
int hvs_gen_data(hvs_ctx *ctx, uint32_t n, uint64_t seed, int profile, uint32_t ncat);
/* Copy raw rows [row0,row0+nrows) back to the host (n x 102 layout). */
int hvs_download_data(hvs_ctx *ctx, uint32_t row0, uint32_t nrows, float *out_rows);
uint32_t hvs_num_rows(const hvs_ctx *ctx);

/* ---- the vec_query seam ------------------------------------------------------------------ */

/*
 * q_rows: host, nq x 104 f32.  out_ids: host, nq x 100 u32 (output.bin row layout, ascending
 * distance, canonical tie rule (dist asc, id asc)).  out_dists: host, nq x 100 f32 exact-order
 * distances of those ids, or NULL.  sample_proportion as in the reference: rows [0, sn) are
 * searched with sn = uint32(float(sample_proportion) * float(n)) (optimized_parallel.hpp:67);
 * padding ids come from the end of the full set (n-1, n-2, ...), optimized_parallel.hpp:149-157.
 * The call is a pipeline: queries go to the GPU in pieces through pinned staging slots one batch ahead of the engine,
 * finished batches' ids come back while the next batch computes (buffers the caller pinned itself skip the staging
 * copies).  hvs_last_timing().host_ms is the wall time of the whole call.
 */
int hvs_query(hvs_ctx *ctx, const float *q_rows, uint32_t nq, float sample_proportion, uint32_t *out_ids,
              float *out_dists);

/* ---- device-resident variant (benchmarks, multi-GPU drivers: inputs already in HBM) ------ */

int hvs_upload_queries(hvs_ctx *ctx, const float *q_rows, uint32_t nq);
int hvs_gen_queries(hvs_ctx *ctx, uint32_t nq, uint64_t seed, int profile, uint32_t ncat, int force_type,
                    uint64_t first_row);
int hvs_download_queries(hvs_ctx *ctx, uint32_t q0, uint32_t nq, float *out_rows);
/* Answer resident queries [q0, q0+nq); results stay on the device (rows q0..q0+nq of the result
 * buffer).  Asynchronous on the context stream(s); hvs_sync waits (and lets the exact engine re-run the few queries
 * whose filter lists overflowed, if any). */
int hvs_query_resident(hvs_ctx *ctx, uint32_t q0, uint32_t nq, float sample_proportion);
int hvs_sync(hvs_ctx *ctx);
int hvs_download_results(hvs_ctx *ctx, uint32_t q0, uint32_t nq, uint32_t *out_ids, float *out_dists);
/* Copy result rows [q0,q0+nq) into caller-owned DEVICE buffers (same GPU), e.g. a collective's
 * send buffer.  d_dists may be NULL.  Asynchronous on the context stream.  Single-GPU contexts only. */
int hvs_export_results_device(hvs_ctx *ctx, uint32_t q0, uint32_t nq, uint32_t *d_ids, float *d_dists);
/* Stream-ordered hand-off: work enqueued on `stream` (a hipStream_t of the context's GPU, e.g. the stream a collective
 * runs on) after this call starts only when everything the context has enqueued so far is complete -- no host wait for
 * the kernels (queries the filter left unanswered are re-run first, which costs the call's one host synchronisation).
 * Single-GPU contexts only. */
int hvs_stream_wait(hvs_ctx *ctx, void *stream);
/* D-sharded mode (rows partitioned over GPUs, every GPU answers all queries on its rows with hvs_set_padding(ctx, 0)):
 * merges the shards' partial answers on the device -- the multi-GPU counterpart of Knn::merge (reference
 * include/optimized_impl.h:337-385) -- and applies the reference's padding (optimized_parallel.hpp:149-157) once.
 * d_ids_all / d_dists_all: DEVICE, [nshards][nq][100] as an all_gather of the per-shard results lays them out (ids
 * shard-local, 0xFFFFFFFF = empty slot); shard_row0: HOST, first global row of each shard (nshards <= 16);
 * d_pad_dists: DEVICE, [nq][100], exact-order distance of query q to row n_total-1-s; outputs: DEVICE, [nq][100], global
 * ids in ascending (dist, id) order (d_out_dists may be NULL).  Asynchronous on the context stream.  Single-GPU
 * contexts only. */
int hvs_merge_shards_device(hvs_ctx *ctx, uint32_t nshards, uint32_t nq, const uint32_t *d_ids_all,
                            const float *d_dists_all, const uint64_t *shard_row0, uint32_t n_total,
                            const float *d_pad_dists, uint32_t *d_out_ids, float *d_out_dists);
/* Timing of the last hvs_query / hvs_query_resident (call after hvs_sync). */
int hvs_last_timing(hvs_ctx *ctx, hvs_timing *out);
/* Diagnostics: which queries of the last call were answered a second time (call after hvs_sync / hvs_query).
 * which = 0: the exact engine's list (hvs_timing.fallback_queries), 1: the retry list (hvs_timing.retry_queries).
 * Copies up to cap query indices (relative to the call's resident query set) to out_idx and returns the list's length,
 * or a negative HVS_E* code.  Single-GPU contexts only. */
int hvs_last_reruns(hvs_ctx *ctx, int which, uint32_t *out_idx, uint32_t cap);

/* Host-side planning rules, exposed for tests (pure arithmetic: no GPU, no context).
 * hvs_plan_guess_m: the order statistic m of a guessed threshold (csrc/hvs_filter.h, "Guessed thresholds") for k neighbours when
 * a fraction `seen_fraction` of the query's rows has been seen and one guess may fail with probability 10^-pfail.
 * hvs_plan_batches: the batch sizes hvs_query (host_pipeline != 0) or hvs_query_resident cuts a call of nq queries into for
 * the filter engines; writes up to cap sizes to out (may be NULL) and returns their number. */
uint32_t hvs_plan_guess_m(uint32_t k, double seen_fraction, uint32_t pfail);
uint32_t hvs_plan_batches(uint32_t nq, int host_pipeline, uint32_t *out, uint32_t cap);

const char *hvs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HVS_H */
