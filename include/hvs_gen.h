/*
 * hvs_gen.h -- "gen v1" synthetic D / Q generator (counter based, stateless).
 *
 * The reference ships two rand()-based generators seeded with time(NULL)
 * (reference src/write_data.c:16, src/write_query.c:18), so their output cannot
 * be reproduced.  This header defines the seeded replacement used by the tests,
 * bench.py and the GPU-side generator kernels.  Every value is a pure function
 * of (seed, row, column), so host C, numpy and HIP device code produce the same
 * bits and a GPU box can regenerate multi-GB inputs from three integers.
 *
 * File formats are the reference's (include/io.h:111-136, README.md:31-44):
 *   D row : 102 f32  [C, T, x0..x99]
 *   Q row : 104 f32  [type, v, l, r, x0..x99]   (unused predicate fields = -1)
 *
 * Profiles
 *   HVS_GEN_V1       integer categorical C in [0,ncat), T in [0,1), x in [-6,6),
 *                    query type uniform in {0,1,2,3} (or forced), v integer in
 *                    [0,ncat), l in [0,1), r = l + u*(1-l).  Value ranges follow
 *                    write_data.c:8-13 / write_query.c:30-50 except that C is
 *                    integer so that `C == v` can actually match.
 *   HVS_GEN_V0       the reference generators' ranges: continuous C in [-1,1],
 *                    T in [-3,3], v continuous in [-1,1] (truncated to int by the
 *                    query loop), l in [-3,3], r in [l,4].  Types 1/3 then match
 *                    (almost) nothing, which exercises the padding path
 *                    (optimized_parallel.hpp:149-157).
 *
 *   Non-uniform profiles (same C / T / predicate fields as HVS_GEN_V1; only the vectors differ).  The reference's
 *   own generator is uniform (write_data.c:8-13,28-33) but the contest data it was written for is not (README.md:58-60):
 *   HVS_GEN_CLUSTER  64 Gaussian-like clusters: centre (uniform in [-4.5,4.5)^100, a function of the seed and the
 *                    cluster) + 0.6 g per coordinate, g = Irwin-Hall(4) scaled to unit variance.
 *   HVS_GEN_PCA      per-dimension scale w_k decaying from 1 (k = 0) to 0.014 (k = 99): x_k = 6 w_k g
 *                    (w_k = (32 - (k & 15)) / 32 * 2^-(k >> 4), exact in f32).
 *   HVS_GEN_HEAVY    heavy-tailed row norms: x_k = 1.5 (2u - 1) r, r = 1 + 15 v^4 with ONE v in [0,1) per row (most
 *                    rows r ~ 1, one in 10^4 above 11).
 *   Queries follow the law of the data set (with their own seed); one query in 100 (hash of the row number) lies
 *   outside the data's bounding box: its first three coordinates are 10 % beyond the law's support.
 *   HVS_GEN_V1_OUT   HVS_GEN_V1 data and queries, with the same 1 % of out-of-box queries (first three coordinates +-6.625).
 *
 * All float arithmetic below is single operations on f32 values (no FMA
 * contraction allowed: build with -ffp-contract=off).
 */
#ifndef HVS_GEN_H
#define HVS_GEN_H

#include <stdint.h>

#if defined(__HIPCC__)
#define HVS_HD __host__ __device__ static inline
#else
#define HVS_HD static inline
#endif

#define HVS_DATA_COLS 102u
#define HVS_QUERY_COLS 104u
#define HVS_DIM 100u
#define HVS_K 100u

#define HVS_GEN_V0 0
#define HVS_GEN_V1 1
#define HVS_GEN_CLUSTER 2
#define HVS_GEN_PCA 3
#define HVS_GEN_HEAVY 4
#define HVS_GEN_V1_OUT 5   /* HVS_GEN_V1 whose queries include 1 % outside the data's box */
#define HVS_GEN_NCLUSTERS 64u

#define HVS_SEED_DATA 0xD47A5EEDull
#define HVS_SEED_QUERY 0x9E3779B9ull

/* splitmix64 finaliser */
HVS_HD uint64_t hvs_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* 24 uniform bits for (seed, row, col) */
HVS_HD uint32_t hvs_u24(uint64_t seed, uint64_t row, uint32_t col)
{
    return (uint32_t)(hvs_mix64(hvs_mix64(seed) + row * 128ull + (uint64_t)col) >> 40);
}

/* uniform f32 in [0,1), exactly u24 * 2^-24 */
HVS_HD float hvs_u01(uint32_t u24) { return (float)u24 * 5.9604644775390625e-08f; }

/* lo + (hi-lo)*u as two rounded f32 operations */
HVS_HD float hvs_affine(float u, float scale, float lo)
{
    float t = scale * u;
    return t + lo;
}

/* unit-variance bell-shaped value from four uniforms (Irwin-Hall), single f32 operations in a fixed order */
HVS_HD float hvs_gen_bell(uint64_t seed, uint64_t row, uint32_t col)
{
    const float u0 = hvs_u01(hvs_u24(seed, row, col));
    const float u1 = hvs_u01(hvs_u24(seed + 0x1234567ull, row, col));
    const float u2 = hvs_u01(hvs_u24(seed + 0x2468ACEull, row, col));
    const float u3 = hvs_u01(hvs_u24(seed + 0x369D035ull, row, col));
    const float a = u0 + u1;
    const float b = u2 + u3;
    const float s = a + b;
    const float c = s - 2.0f;
    return c * 1.7320508f;
}

/* vector component k (0..99) of row `row` in the non-uniform profiles; `law_seed` selects the data set's law (cluster
 * centres), `seed` the draw */
HVS_HD float hvs_gen_vec_elem(uint64_t seed, uint64_t law_seed, int profile, uint64_t row, uint32_t k)
{
    const uint32_t col = k + 2u;
    if (profile == HVS_GEN_CLUSTER) {
        const uint64_t cl = hvs_mix64(hvs_mix64(seed) ^ (row * 0x9E3779B97F4A7C15ull)) % HVS_GEN_NCLUSTERS;
        const float centre = hvs_affine(hvs_u01(hvs_u24(law_seed ^ 0xC1057E5ull, cl, col)), 9.0f, -4.5f);
        const float g = hvs_gen_bell(seed, row, col);
        const float t = 0.6f * g;
        return centre + t;
    }
    if (profile == HVS_GEN_PCA) {
        const float w = (float)(32u - (k & 15u)) * 0.03125f;             /* (32 - j) / 32 */
        float p2 = 1.0f;                                                 /* 2^-(k >> 4): exact */
        for (uint32_t i = 0; i < (k >> 4); ++i) p2 = p2 * 0.5f;
        const float wk = w * p2;
        const float g = hvs_gen_bell(seed, row, col);
        const float t = 6.0f * wk;
        return t * g;
    }
    /* HVS_GEN_HEAVY */
    {
        const float v = hvs_u01(hvs_u24(seed + 0x51ED270ull, row, 127u));
        const float v2 = v * v;
        const float v4 = v2 * v2;
        const float r15 = 15.0f * v4;
        const float r = 1.0f + r15;
        const float x = hvs_affine(hvs_u01(hvs_u24(seed, row, col)), 3.0f, -1.5f);
        return x * r;
    }
}

/* one query in 100 is pushed outside the data's bounding box */
HVS_HD int hvs_gen_query_is_outlier(uint64_t seed, uint64_t row)
{
    return (hvs_mix64(hvs_mix64(seed + 0x0B0Full) + row) % 100ull) == 0ull;
}

/* one element of a data row */
HVS_HD float hvs_gen_data_elem(uint64_t seed, int profile, uint32_t ncat, uint64_t row, uint32_t col)
{
    const uint32_t u = hvs_u24(seed, row, col);
    if (col == 0u)
        return profile != HVS_GEN_V0 ? (float)(u % ncat) : hvs_affine(hvs_u01(u), 2.0f, -1.0f);
    if (col == 1u)
        return profile != HVS_GEN_V0 ? hvs_u01(u) : hvs_affine(hvs_u01(u), 6.0f, -3.0f);
    if (profile >= HVS_GEN_CLUSTER && profile <= HVS_GEN_HEAVY) return hvs_gen_vec_elem(seed, HVS_SEED_DATA, profile, row, col - 2u);
    return hvs_affine(hvs_u01(u), 12.0f, -6.0f);
}

/* one element of a query row; force_type < 0 -> uniform in {0,1,2,3} */
HVS_HD float hvs_gen_query_elem(uint64_t seed, int profile, uint32_t ncat, int force_type, uint64_t row,
                                uint32_t col)
{
    const uint32_t type = force_type >= 0 ? (uint32_t)force_type : (hvs_u24(seed, row, 0u) & 3u);
    const uint32_t u = hvs_u24(seed, row, col);
    if (col == 0u) return (float)type;
    if (col == 1u) {
        if (!(type & 1u)) return -1.0f;
        return profile != HVS_GEN_V0 ? (float)(u % ncat) : hvs_affine(hvs_u01(u), 2.0f, -1.0f);
    }
    if (col == 2u || col == 3u) {
        if (!(type & 2u)) return -1.0f;
        const float ul = hvs_u01(hvs_u24(seed, row, 2u));
        const float l = profile != HVS_GEN_V0 ? ul : hvs_affine(ul, 6.0f, -3.0f);
        if (col == 2u) return l;
        const float hi = profile != HVS_GEN_V0 ? 1.0f : 4.0f;
        const float span = hi - l;
        const float t = hvs_u01(u) * span;
        return l + t;
    }
    if (profile >= HVS_GEN_CLUSTER && profile <= HVS_GEN_HEAVY) {
        /* the law of the data set (cluster centres of HVS_SEED_DATA), the query's own draw; 1 % pushed outside the box */
        const uint32_t k = col - 4u;
        const float x = hvs_gen_vec_elem(seed, HVS_SEED_DATA, profile, row, k);
        if (k >= 3u || !hvs_gen_query_is_outlier(seed, row)) return x;
        /* the first three coordinates 10 % beyond anything the law can produce (7.25 / 23 w_k / 26.5 are above 1.1 x
         * the laws' supports 6.58 / 20.8 w_k / 24), so the query is certainly outside the data's box */
        const float w = (float)(32u - k) * 0.03125f;
        const float far = profile == HVS_GEN_CLUSTER ? 7.25f : (profile == HVS_GEN_PCA ? 23.0f * w : 26.5f);
        return (u & 1u) ? far : -far;
    }
    if (profile == HVS_GEN_V1_OUT && col < 7u && hvs_gen_query_is_outlier(seed, row)) return (u & 1u) ? 6.625f : -6.625f;
    return hvs_affine(hvs_u01(u), 12.0f, -6.0f);
}

#endif /* HVS_GEN_H */
