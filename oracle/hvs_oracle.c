#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* sched_setaffinity: the timed CPU baseline pins its threads */
#endif
/*
 * hvs_oracle.c -- CPU restatement of the reference's filtered brute-force k-NN path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (libhvs.so) never links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
 * against (a) the reference's own known-answer values from
 * src/fp_inaccuracy_test.cpp:77-97 and (b) output.bin files produced in the build
 * container by the real reference binaries (oracle/_ref/, built by oracle/Makefile
 * straight from /root/reference) and committed under tests/golden/.
 *
 * Each function cites the reference lines it restates.  Build with
 *   gcc -O3 -mavx2 -ffp-contract=off -fopenmp   (no -ffast-math, no FMA contraction:
 *   the reference is built -O3 -mavx2 without -mfma, CMakeLists.txt:8).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#ifdef __linux__
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <sched.h>
#endif
#endif

#include "../include/hvs_gen.h"

/* Thread t of T on a CPU of its own, spread evenly over the CPUs this process may use, for the duration of a parallel region of
 * the timed baseline (the partition a thread first touches is then the one it scans, on the same memory node).  The calling
 * thread (t = 0) gets its old mask back from pin_restore: threads it starts later must not inherit a one-CPU mask. */
#ifdef __linux__
static int g_pin_threads = 0; /* hvs_oracle_pin_threads */
static void pin_self(uint32_t t, uint32_t T, cpu_set_t *saved)
{
    if (!g_pin_threads) return;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    if (saved) *saved = allowed;
    /* one hardware thread per core first (the lowest-numbered CPU of each thread_siblings_list), the SMT siblings behind them:
     * T <= cores threads are spread evenly over the cores (and thereby over the sockets), more threads fill the siblings */
    static int order[CPU_SETSIZE], nprim = -1, nall = 0;
#pragma omp critical(hvs_pin_topology)
    if (nprim < 0) {
        int prim[CPU_SETSIZE], sec[CPU_SETSIZE], np = 0, ns = 0;
        for (int c = 0; c < CPU_SETSIZE; ++c) {
            if (!CPU_ISSET(c, &allowed)) continue;
            char path[128];
            int first = c;
            snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
            FILE *f = fopen(path, "r");
            if (f) {
                if (fscanf(f, "%d", &first) != 1) first = c;
                fclose(f);
            }
            if (first == c) prim[np++] = c; else sec[ns++] = c;
        }
        for (int i = 0; i < np; ++i) order[i] = prim[i];
        for (int i = 0; i < ns; ++i) order[np + i] = sec[i];
        nall = np + ns;
        nprim = np;
    }
    if (nall < (int)T) return; /* fewer CPUs than threads: leave it to the scheduler */
    const int cpu = (int)T <= nprim ? order[(size_t)t * (size_t)nprim / T] : order[t];
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(cpu, &one);
    (void)sched_setaffinity(0, sizeof(one), &one);
}
static void pin_restore(const cpu_set_t *saved)
{
    if (g_pin_threads) (void)sched_setaffinity(0, sizeof(*saved), saved);
}
#endif
void hvs_oracle_pin_threads(int on)
{
#ifdef __linux__
    g_pin_threads = on;
#else
    (void)on;
#endif
}


#define DCOLS 102
#define QCOLS 104
/* k: the reference's compile-time KNN_LIMIT (optimized_impl.h:26).  100 unless a test changes it with
 * hvs_oracle_set_k (8..256; process-wide, set before the calls that use it -- the checker is single-tenant). */
#define KNN_MAX 256
static int g_knn = 100;
#define KNN g_knn
int hvs_oracle_set_k(int k)
{
    if (k < 8 || k > KNN_MAX) return -1;
    g_knn = k;
    return 0;
}
int hvs_oracle_get_k(void) { return g_knn; }

/* ------------------------------------------------------------------------- *
 * Distances
 * ------------------------------------------------------------------------- */

/*
 * Exact-order squared L2 of the compiled hot-path variant (DIST_SIMD=1,
 * DIST_BAIL_OUT=0; optimized_parallel.hpp:49,55).
 *   optimized_impl.h:96-105  12 full 8-wide steps over row indices 2..97:
 *                            acc[j] = acc[j] + ((d-q)*(d-q)), each op rounded to f32
 *   optimized_impl.h:110-123 masked tail: lanes 4..7 take row indices 98..101,
 *                            lanes 0..3 add (0-0)^2 = +0
 *   optimized_impl.h:37-47   hsum: s_j = acc[j]+acc[j+4]; (s0+s1)+(s2+s3)
 * `dvec` / `qvec` point at the 100 vector dims (row+2 / query+4).
 */
float hvs_oracle_dist_simd_order(const float *dvec, const float *qvec)
{
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < 12; ++b) {
        for (int j = 0; j < 8; ++j) {
            float t = dvec[8 * b + j] - qvec[8 * b + j];
            t = t * t;
            acc[j] = acc[j] + t;
        }
    }
    for (int j = 0; j < 4; ++j) acc[j] = acc[j] + 0.0f;
    for (int j = 4; j < 8; ++j) {
        float t = dvec[92 + j] - qvec[92 + j];
        t = t * t;
        acc[j] = acc[j] + t;
    }
    const float s0 = acc[0] + acc[4];
    const float s1 = acc[1] + acc[5];
    const float s2 = acc[2] + acc[6];
    const float s3 = acc[3] + acc[7];
    const float a = s0 + s1;
    const float b2 = s2 + s3;
    return a + b2;
}

/*
 * Sequential-order squared L2: baseline.hpp:53-64 (compare_with_id) and the
 * `.dist` side file's calc_dist (io.h:38-48).
 */
float hvs_oracle_dist_scalar_order(const float *dvec, const float *qvec)
{
    float sum = 0.0f;
    for (int i = 0; i < 100; ++i) {
        float diff = dvec[i] - qvec[i];
        diff = diff * diff;
        sum = sum + diff;
    }
    return sum;
}

/* ------------------------------------------------------------------------- *
 * Query parsing and predicate
 * ------------------------------------------------------------------------- */

/* optimized_parallel.hpp:67: sn = uint32_t(sample_proportion * n), a float product */
uint32_t hvs_oracle_sn(float sample_proportion, uint32_t n)
{
    const float p = sample_proportion * (float)n;
    if (!(p > 0.0f)) return 0u;
    if (p >= 4294967296.0f) return n;
    const uint32_t sn = (uint32_t)p;
    return sn > n ? n : sn;
}

typedef struct {
    uint32_t type;
    float vf; /* float(int32(q[1])): what `nodes[j][0] == v` compares against */
    float l, r;
} qparams;

/* optimized_parallel.hpp:93-96: type=uint32(q[0]); v=int32(q[1]) (truncation); l=q[2]; r=q[3].
 * Out-of-range / NaN type or v is undefined behaviour in the reference; here such a
 * query matches no row (type -> 4). */
static qparams parse_query(const float *q)
{
    qparams p;
    const float t = q[0];
    p.type = (t > -1.0f && t < 4.0f) ? (uint32_t)(int32_t)t : 4u; /* uint32(-0.5f) == 0: defined, type 0 */
    const float v = q[1];
    if (v >= -2147483648.0f && v < 2147483648.0f) { /* int32(-2^31) is INT_MIN: defined */
        p.vf = (float)(int32_t)v;
    } else {
        p.vf = 0.0f;
        if (p.type == 1u || p.type == 3u) p.type = 4u;
    }
    p.l = q[2];
    p.r = q[3];
    return p;
}

/* optimized_parallel.hpp:105-138 */
static inline int row_passes(const qparams *p, const float *row)
{
    switch (p->type) {
    case 0: return 1;
    case 1: return row[0] == p->vf;
    case 2: return row[1] >= p->l && row[1] <= p->r;
    case 3: return row[0] == p->vf && row[1] >= p->l && row[1] <= p->r;
    default: return 0;
    }
}

int hvs_oracle_predicate(const float *row, const float *q)
{
    const qparams p = parse_query(q);
    return row_passes(&p, row);
}

/* ------------------------------------------------------------------------- *
 * Canonical answer: optimized.hpp:72-131 (serial path) with the build's
 * deterministic tie rule (dist asc, id asc); SURVEY.md section 8c.
 * ------------------------------------------------------------------------- */

typedef struct {
    float d;
    uint32_t id;
} cand;

/* Canonical order (dist asc, id asc) as a TOTAL order on the distance bits: distances are sums of squares (>= +0), so
 * their bit patterns order like the values; +inf follows every finite value and NaN -- one canonical pattern, x86 and
 * gfx950 produce different ones -- follows +inf.  The reference admits any row while its list is not full
 * (optimized_impl.h:301-304) and then calls std::sort, whose result for NaN keys is unspecified: for NaN this order is
 * the build's own choice, for +inf it is the canonical rule. */
static inline uint32_t dist_bits(float d)
{
    uint32_t u;
    if (d != d) return 0x7FC00000u;
    memcpy(&u, &d, 4);
    return u;
}
static inline float canon_nan(float d)
{
    const uint32_t u = dist_bits(d);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static inline int cand_less(const cand *a, const cand *b)
{
    const uint32_t ua = dist_bits(a->d), ub = dist_bits(b->d);
    return ua < ub || (ua == ub && a->id < b->id);
}

static int cand_cmp(const void *a, const void *b)
{
    const cand *x = (const cand *)a, *y = (const cand *)b;
    if (cand_less(x, y)) return -1;
    if (cand_less(y, x)) return 1;
    return 0;
}

/* max-heap on (d,id): root is the worst kept candidate */
static void heap_sift_down(cand *h, int n, int i)
{
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && cand_less(&h[m], &h[l])) m = l;
        if (r < n && cand_less(&h[m], &h[r])) m = r;
        if (m == i) return;
        cand t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}

static void heap_sift_up(cand *h, int i)
{
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!cand_less(&h[p], &h[i])) return;
        cand t = h[i];
        h[i] = h[p];
        h[p] = t;
        i = p;
    }
}

/* scan rows [start,end) into a top-KNN heap; returns new fill */
static int scan_range(const float *nodes, uint32_t start, uint32_t end, const qparams *p, const float *qvec,
                      cand *heap, int fill)
{
    for (uint32_t j = start; j < end; ++j) {
        const float *row = nodes + (size_t)j * DCOLS;
        if (!row_passes(p, row)) continue;
        cand c;
        c.d = hvs_oracle_dist_simd_order(row + 2, qvec);
        c.id = j;
        if (fill < KNN) {
            heap[fill] = c;
            heap_sift_up(heap, fill);
            ++fill;
        } else if (cand_less(&c, &heap[0])) {
            heap[0] = c;
            heap_sift_down(heap, KNN, 0);
        }
    }
    return fill;
}

/* optimized.hpp:120-128 / optimized_parallel.hpp:149-157: pad with rows n-1, n-2, ...
 * regardless of predicate and of duplicates, distances by the same exact-order kernel */
static int pad_tail(const float *nodes, uint32_t n, const float *qvec, cand *out, int fill)
{
    uint32_t s = 1;
    while (fill < KNN) {
        const uint32_t id = n - s;
        out[fill].d = hvs_oracle_dist_simd_order(nodes + (size_t)id * DCOLS + 2, qvec);
        out[fill].id = id;
        ++fill;
        ++s;
    }
    return fill;
}

static void one_query_canonical(const float *nodes, uint32_t n, uint32_t sn, const float *q, uint32_t *out_ids,
                                float *out_dists)
{
    cand heap[KNN];
    const qparams p = parse_query(q);
    int fill = scan_range(nodes, 0, sn, &p, q + 4, heap, 0);
    fill = pad_tail(nodes, n, q + 4, heap, fill);
    qsort(heap, KNN, sizeof(cand), cand_cmp);
    for (int k = 0; k < KNN; ++k) {
        out_ids[k] = heap[k].id;
        if (out_dists) out_dists[k] = canon_nan(heap[k].d);
    }
}

/*
 * nodes: n x 102, queries: nq x 104, out_ids: nq x 100, out_dists: nq x 100 or NULL.
 * Returns 0, or -1 when n < 100 (the reference's padding index n-s would underflow,
 * optimized.hpp:125).  Queries are independent, so the loop runs under OpenMP
 * when threads > 1 (answer is thread-count independent).
 */
int hvs_oracle_vec_query(const float *nodes, uint32_t n, const float *queries, uint32_t nq,
                         float sample_proportion, uint32_t *out_ids, float *out_dists, int threads)
{
    if (n < (uint32_t)KNN) return -1;
    const uint32_t sn = hvs_oracle_sn(sample_proportion, n);
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < (int64_t)nq; ++i) {
        one_query_canonical(nodes, n, sn, queries + (size_t)i * QCOLS, out_ids + (size_t)i * KNN,
                            out_dists ? out_dists + (size_t)i * KNN : NULL);
    }
    return 0;
}

/* ------------------------------------------------------------------------- *
 * Faithful Knn container: optimized_impl.h:179-438 (unsorted 100-slot array with
 * cached worst slot).  Reproduces which id survives inside equal-distance groups
 * the way the reference's slot-order eviction does; only the final order inside a
 * tie group (unstable std::sort, optimized_impl.h:408-410) is not reproducible.
 * ------------------------------------------------------------------------- */

typedef struct {
    float dist[KNN_MAX];
    uint32_t idx[KNN_MAX];
    uint32_t fill, worst;
} knn_t;

/* optimized_impl.h:205-255 (FIND_WORST_SIMD=1): per AVX lane strict '>' keeps the
 * earliest slot; full 8-wide steps over slots 8 .. K - K%8, then one overlapping step on the last
 * 8 slots (92..99 at K = 100); across lanes the largest slot index among the maxima wins
 * (_mm256_max_epu32). */
static uint32_t knn_find_worst(const knn_t *k)
{
    float lane_d[8];
    uint32_t lane_i[8];
    for (int j = 0; j < 8; ++j) {
        lane_d[j] = k->dist[j];
        lane_i[j] = (uint32_t)j;
    }
    for (int i = 8; i < KNN - (KNN % 8); i += 8) /* optimized_impl.h:212 */
        for (int j = 0; j < 8; ++j)
            if (k->dist[i + j] > lane_d[j]) {
                lane_d[j] = k->dist[i + j];
                lane_i[j] = (uint32_t)(i + j);
            }
    for (int j = 0; j < 8; ++j) /* optimized_impl.h:224-233: slots K-8 .. K-1 */
        if (k->dist[KNN - 8 + j] > lane_d[j]) {
            lane_d[j] = k->dist[KNN - 8 + j];
            lane_i[j] = (uint32_t)(KNN - 8 + j);
        }
    float m = lane_d[0];
    for (int j = 1; j < 8; ++j)
        if (lane_d[j] > m) m = lane_d[j];
    uint32_t sel = 0;
    for (int j = 0; j < 8; ++j)
        if (lane_d[j] == m && lane_i[j] > sel) sel = lane_i[j];
    return sel;
}

/* optimized_impl.h:284-311 (branchless variant); also the body of merge, :337-385 */
static inline void knn_offer(knn_t *k, float d, uint32_t id)
{
    const int not_full = k->fill < (uint32_t)KNN;
    const float worst_dist = k->dist[k->worst];
    const int better = d < worst_dist;
    const int add = not_full || better;
    const uint32_t upd = not_full ? k->fill : k->worst;
    k->fill += (uint32_t)not_full;
    k->dist[upd] = add ? d : worst_dist;
    k->idx[upd] = add ? id : k->idx[k->worst];
    k->worst = better ? k->worst : upd;
    if (better && !not_full) k->worst = knn_find_worst(k);
}

static void knn_scan(knn_t *k, const float *nodes, uint32_t start, uint32_t end, const qparams *p,
                     const float *qvec)
{
    for (uint32_t j = start; j < end; ++j) {
        const float *row = nodes + (size_t)j * DCOLS;
        if (row_passes(p, row)) knn_offer(k, hvs_oracle_dist_simd_order(row + 2, qvec), j);
    }
}

static void knn_finish(knn_t *k, const float *nodes, uint32_t n, const float *qvec, uint32_t *out_ids,
                       float *out_dists)
{
    uint32_t s = 1;
    while (k->fill < (uint32_t)KNN) { /* optimized_parallel.hpp:149-157 */
        const uint32_t id = n - s;
        knn_offer(k, hvs_oracle_dist_simd_order(nodes + (size_t)id * DCOLS + 2, qvec), id);
        ++s;
    }
    cand tmp[KNN];
    for (int i = 0; i < KNN; ++i) {
        tmp[i].d = k->dist[i];
        tmp[i].id = k->idx[i];
    }
    qsort(tmp, KNN, sizeof(cand), cand_cmp);
    for (int i = 0; i < KNN; ++i) {
        out_ids[i] = tmp[i].id;
        if (out_dists) out_dists[i] = tmp[i].d;
    }
}

/*
 * Reference-faithful engine.  part_threads = 1 restates optimized.hpp:72-131
 * (one Knn).  part_threads = T > 1 restates optimized_parallel.hpp:91-160 with
 * threading.hpp:116-118's static partition (worker t scans [t*floor(sn/T),
 * (t+1)*floor(sn/T)), the last one takes the remainder) and the serial merge of
 * the per-thread containers in thread order (:142-146).  part_threads = 0 applies
 * the reference's own rule T = max(1, min(hw, sn/100000)) (:76-77) with hw =
 * `hw_threads`.  When run_parallel != 0 the T partitions of one query really run
 * on T OpenMP threads (this is the timed CPU baseline); otherwise they run one
 * after another (same answer).  The Knn containers are never reset between
 * queries beyond fill/worst (init(), optimized_impl.h:277-282).
 */
int hvs_oracle_vec_query_knn(const float *nodes, uint32_t n, const float *queries, uint32_t nq,
                             float sample_proportion, uint32_t *out_ids, float *out_dists, int part_threads,
                             int hw_threads, int run_parallel)
{
    if (n < (uint32_t)KNN) return -1;
    const uint32_t sn = hvs_oracle_sn(sample_proportion, n);
    uint32_t T = (uint32_t)part_threads;
    if (part_threads <= 0) {
        uint32_t hw = hw_threads > 0 ? (uint32_t)hw_threads : 1u;
        uint32_t w = sn / 100000u;
        T = hw < w ? hw : w;
        if (T < 1u) T = 1u;
    }
    knn_t *knns = (knn_t *)calloc(T, sizeof(knn_t));
    if (!knns) return -2;
    const uint32_t wsize = sn / T;

    if (run_parallel && T > 1) {
#pragma omp parallel num_threads((int)T)
        {
#ifdef _OPENMP
            const uint32_t t = (uint32_t)omp_get_thread_num();
#else
            const uint32_t t = 0;
#endif
#ifdef __linux__
            cpu_set_t saved;
            CPU_ZERO(&saved);
            pin_self(t, T, t == 0 ? &saved : NULL);
#endif
            for (uint32_t i = 0; i < nq; ++i) {
                const float *q = queries + (size_t)i * QCOLS;
                const qparams p = parse_query(q);
                knns[t].fill = 0;
                knns[t].worst = 0;
                const uint32_t start = t * wsize;
                const uint32_t end = (t == T - 1) ? sn : start + wsize;
                knn_scan(&knns[t], nodes, start, end, &p, q + 4);
#pragma omp barrier
#pragma omp single
                {
                    knn_t fin = knns[0];
                    for (uint32_t j = 1; j < T; ++j)
                        for (uint32_t e = 0; e < knns[j].fill; ++e)
                            knn_offer(&fin, knns[j].dist[e], knns[j].idx[e]);
                    knn_finish(&fin, nodes, n, q + 4, out_ids + (size_t)i * KNN,
                               out_dists ? out_dists + (size_t)i * KNN : NULL);
                } /* implicit barrier */
            }
#ifdef __linux__
            if (t == 0) pin_restore(&saved);
#endif
        }
    } else {
        for (uint32_t i = 0; i < nq; ++i) {
            const float *q = queries + (size_t)i * QCOLS;
            const qparams p = parse_query(q);
            for (uint32_t t = 0; t < T; ++t) {
                knns[t].fill = 0;
                knns[t].worst = 0;
                const uint32_t start = t * wsize;
                const uint32_t end = (t == T - 1) ? sn : start + wsize;
                knn_scan(&knns[t], nodes, start, end, &p, q + 4);
            }
            knn_t fin = knns[0];
            for (uint32_t j = 1; j < T; ++j)
                for (uint32_t e = 0; e < knns[j].fill; ++e) knn_offer(&fin, knns[j].dist[e], knns[j].idx[e]);
            knn_finish(&fin, nodes, n, q + 4, out_ids + (size_t)i * KNN,
                       out_dists ? out_dists + (size_t)i * KNN : NULL);
        }
    }
    free(knns);
    return 0;
}

/*
 * Placement of D for the timed CPU baseline on a multi-socket host.  The reference reads D into per-row heap blocks from
 * its main thread (io.h:111-136), i.e. on ONE memory node, and its workers then scan fixed partitions
 * (threading.hpp:116-118); on a two-socket GPU host that leaves half of the threads reading remote memory.  To time the
 * reference's algorithm at its best, thread t of the SAME static partition (and, with OMP_PROC_BIND set, the same core)
 * copies its rows into `dst` first, so that every partition's pages are first touched -- and therefore placed -- next to
 * the thread that will scan them.  `dst` must be freshly allocated, untouched memory of n x 102 floats.
 */
int hvs_oracle_place_rows(float *dst, const float *src, uint32_t n, float sample_proportion, int part_threads, int hw_threads)
{
    const uint32_t sn = hvs_oracle_sn(sample_proportion, n);
    uint32_t T = (uint32_t)part_threads;
    if (part_threads <= 0) {
        uint32_t hw = hw_threads > 0 ? (uint32_t)hw_threads : 1u;
        uint32_t w = sn / 100000u;
        T = hw < w ? hw : w;
        if (T < 1u) T = 1u;
    }
    const uint32_t wsize = sn / T;
#pragma omp parallel num_threads((int)T)
    {
#ifdef _OPENMP
        const uint32_t t = (uint32_t)omp_get_thread_num();
#else
        const uint32_t t = 0;
#endif
#ifdef __linux__
        cpu_set_t saved;
        CPU_ZERO(&saved);
        pin_self(t, T, t == 0 ? &saved : NULL); /* (worker threads of the OpenMP pool stay where they are put) */
#endif
        const size_t start = (size_t)t * wsize;
        const size_t end = (t == T - 1) ? (size_t)n : start + wsize; /* the last thread also takes rows [sn, n) */
        memcpy(dst + start * DCOLS, src + start * DCOLS, (end - start) * DCOLS * sizeof(float));
#ifdef __linux__
        if (t == 0) pin_restore(&saved);
#endif
    }
    return (int)T;
}

/* ------------------------------------------------------------------------- *
 * Baseline engine (BASELINE.json configs[0]): baseline.hpp:68-190.
 * candidates = passing rows ascending, padded with n-1, n-2, ... (:139-147),
 * scalar-order distances (:53-64,152-155), sort of the candidate positions by
 * distance (:159-166; std::sort is unstable, we break ties by candidate position),
 * first 100 (:167-172).
 * ------------------------------------------------------------------------- */

typedef struct {
    float d;
    uint32_t pos;
} bcand;

static int bcand_cmp(const void *a, const void *b)
{
    const bcand *x = (const bcand *)a, *y = (const bcand *)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);
}

int hvs_oracle_vec_query_baseline(const float *nodes, uint32_t n, const float *queries, uint32_t nq,
                                  float sample_proportion, uint32_t *out_ids, float *out_dists)
{
    if (n < (uint32_t)KNN) return -1;
    const uint32_t sn = hvs_oracle_sn(sample_proportion, n);
    uint32_t *ids = (uint32_t *)malloc(((size_t)sn + KNN) * sizeof(uint32_t));
    bcand *bc = (bcand *)malloc(((size_t)sn + KNN) * sizeof(bcand));
    if (!ids || !bc) {
        free(ids);
        free(bc);
        return -2;
    }
    for (uint32_t i = 0; i < nq; ++i) {
        const float *q = queries + (size_t)i * QCOLS;
        const qparams p = parse_query(q);
        size_t m = 0;
        for (uint32_t j = 0; j < sn; ++j)
            if (row_passes(&p, nodes + (size_t)j * DCOLS)) ids[m++] = j;
        uint32_t s = 1;
        while (m < (size_t)KNN) ids[m++] = n - s++;
        for (size_t c = 0; c < m; ++c) {
            bc[c].d = hvs_oracle_dist_scalar_order(nodes + (size_t)ids[c] * DCOLS + 2, q + 4);
            bc[c].pos = (uint32_t)c;
        }
        qsort(bc, m, sizeof(bcand), bcand_cmp);
        for (int k = 0; k < KNN; ++k) {
            out_ids[(size_t)i * KNN + k] = ids[bc[k].pos];
            if (out_dists) out_dists[(size_t)i * KNN + k] = bc[k].d;
        }
    }
    free(ids);
    free(bc);
    return 0;
}

/* ------------------------------------------------------------------------- *
 * .dist side file values: src/test.cpp:97-110 + io.h:50-78 -- for every output id
 * the *scalar-order* distance between that row and the query.
 * ------------------------------------------------------------------------- */
void hvs_oracle_dist_file_values(const float *nodes, const float *queries, uint32_t nq, const uint32_t *ids,
                                 float *out)
{
    for (uint32_t i = 0; i < nq; ++i)
        for (int k = 0; k < KNN; ++k)
            out[(size_t)i * KNN + k] = hvs_oracle_dist_scalar_order(
                nodes + (size_t)ids[(size_t)i * KNN + k] * DCOLS + 2, queries + (size_t)i * QCOLS + 4);
}

/* exact-order distances for given ids (used to build tie groups for goldens) */
void hvs_oracle_dists_for_ids(const float *nodes, const float *queries, uint32_t nq, const uint32_t *ids,
                              float *out)
{
    for (uint32_t i = 0; i < nq; ++i)
        for (int k = 0; k < KNN; ++k)
            out[(size_t)i * KNN + k] = canon_nan(hvs_oracle_dist_simd_order(
                nodes + (size_t)ids[(size_t)i * KNN + k] * DCOLS + 2, queries + (size_t)i * QCOLS + 4));
}

/* ------------------------------------------------------------------------- *
 * gen v1 fills (host twin of the device generator; include/hvs_gen.h)
 * ------------------------------------------------------------------------- */
void hvs_oracle_gen_data(float *out, uint64_t row0, uint64_t nrows, uint64_t seed, int profile, uint32_t ncat)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)nrows; ++i)
        for (uint32_t c = 0; c < DCOLS; ++c)
            out[(size_t)i * DCOLS + c] = hvs_gen_data_elem(seed, profile, ncat, row0 + (uint64_t)i, c);
}

void hvs_oracle_gen_queries(float *out, uint64_t row0, uint64_t nrows, uint64_t seed, int profile, uint32_t ncat,
                            int force_type)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)nrows; ++i)
        for (uint32_t c = 0; c < QCOLS; ++c)
            out[(size_t)i * QCOLS + c] = hvs_gen_query_elem(seed, profile, ncat, force_type, row0 + (uint64_t)i, c);
}
