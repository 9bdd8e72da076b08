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
 * All float arithmetic below is single operations on f32 values (no FMA
 * contraction allowed: build with -ffp-contract=off).
 */
#ifndef HVS_GEN_H
#define HVS_GEN_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
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

/* one element of a data row */
HVS_HD float hvs_gen_data_elem(uint64_t seed, int profile, uint32_t ncat, uint64_t row, uint32_t col)
{
    const uint32_t u = hvs_u24(seed, row, col);
    if (col == 0u)
        return profile == HVS_GEN_V1 ? (float)(u % ncat) : hvs_affine(hvs_u01(u), 2.0f, -1.0f);
    if (col == 1u)
        return profile == HVS_GEN_V1 ? hvs_u01(u) : hvs_affine(hvs_u01(u), 6.0f, -3.0f);
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
        return profile == HVS_GEN_V1 ? (float)(u % ncat) : hvs_affine(hvs_u01(u), 2.0f, -1.0f);
    }
    if (col == 2u || col == 3u) {
        if (!(type & 2u)) return -1.0f;
        const float ul = hvs_u01(hvs_u24(seed, row, 2u));
        const float l = profile == HVS_GEN_V1 ? ul : hvs_affine(ul, 6.0f, -3.0f);
        if (col == 2u) return l;
        const float hi = profile == HVS_GEN_V1 ? 1.0f : 4.0f;
        const float span = hi - l;
        const float t = hvs_u01(u) * span;
        return l + t;
    }
    return hvs_affine(hvs_u01(u), 12.0f, -6.0f);
}

#endif /* HVS_GEN_H */
