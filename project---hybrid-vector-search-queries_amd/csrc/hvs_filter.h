// hvs_filter.h -- the MFMA filter engines: matrix-core bound filter (INT8, FP16 or BF16 tiles) + exact-order re-scoring.
//
// Idea.  The reference evaluates 300 non-fusable f32 operations for every (query,row) pair that
// passes the predicate (optimized_impl.h:96-125).  Only a few hundred of those pairs can ever
// enter a query's top-k.  This engine proves, with a cheap matrix-core product and a rigorous
// error bound, that almost every pair CANNOT enter the top-k and runs the exact-order kernel only
// on the few survivors.  Membership and order of the answer are always decided by exact-order f32
// distances, never by the approximate value, so the output is bit-identical to the exact engine's.
//
// Index (built once per data set, "it is prohibitive to use query vectors during indexing" --
// reference README.md:68 -- is respected: only D is used):
//   * two orderings of the rows: by (C,T) and by T.  Every predicate of the reference
//     (optimized_parallel.hpp:105-138) is then a contiguous POSITION RANGE: type 0 = everything,
//     type 1 = the C==v run, type 3 = the T-window inside that run, type 2 = a T-window of the
//     T ordering.  Ranges come from integer binary searches on order-preserving keys.
//   * each ordering is cut into blocks of 32 consecutive positions, stored as MFMA A-operand fragments in
//     one tile format.  16-bit floats (BF16 or IEEE half): K padded 100 -> 112; k = 100..102 hold -|d|^2/2 split into
//     three pieces so that one chain of 7 v_mfma_f32_32x32x16_{bf16,f16} yields  s = q.d - |d|^2/2.  INT8 ("INT8 filter"
//     below): rows and queries centred and quantised with one scale, v_mfma_i32_16x16x64_i8 (or 32x32x32) started from
//     the rows' integer norm terms, exact integer arithmetic.
//   * blocks are stored LEVEL-INTERLEAVED: level 0 = every S0-th block, level j = the multiples of
//     stride[j] not in an earlier level (stride[j-1] = radix[j] * stride[j]; radices 16, ..., 16, 4 by default).  Any
//     position range meets every level in one contiguous storage run, and level j multiplies the rows a query has seen
//     by its radix.  The threshold of a level is GUESSED from the rows seen so far (see hvs_k_merge) and the answer is
//     verified at the end, so each round hands only a few hundred candidates per query to the exact kernel, whatever
//     the range and the radix.
//
// Per batch of queries:  prep (ranges, B fragments, norms) -> level 0 by the exact kernel
// -> merge -> for each further level { MFMA filter (writes 8-byte survivor entries) -> re-scoring kernel
// (expands the entries, exact-order distances) -> merge (top-k, next threshold) }
// -> verify + pad + sort + write.  A query whose guess fails its check is run again with proven thresholds (retry
// batch); a query without a usable bound, or whose lists overflow there, is re-run by the exact engine.
#pragma once

#include "hvs_device.h"
#include "hvs_kernels.h"  // hvs_exact_dist_pk_lds: the exact engine's row loop, run by hvs_k_seed_exact on its LDS images

#define HVS_KPAD 112          // padded contraction length (7 MFMA k-steps of 16)
#define HVS_KSTEPS 7
#define HVS_TILE_U4 (HVS_KSTEPS * 64)  // uint4 per 32-row tile (7 KiB)
#ifndef HVS_QB
#define HVS_QB 4              // query blocks (of 32) per wave in the filter kernel
#endif
#define HVS_GROUP (32 * HVS_QB)
#define HVS_FCAP 1024         // per-query candidate keys per round (the least: HvsBatch::fcap; small batches get more)
#define HVS_GCAP (HVS_GROUP * 1024) // per-group survivor entries per round (the least: HvsBatch::gcap)
#ifndef HVS_SEG
#define HVS_SEG 512           // row blocks per filter work item (round 3: 256 -> 512 +0.5 % at 2^21 queries, equal at 5 x 10^5; 1024
                              // equal / -1 %; round 2: 128 -1 % mixed, -2.7 % type-0 against 256: the item prologues)
#endif
#ifndef HVS_STAGE
#define HVS_STAGE 4           // tiles per LDS stage (one workgroup barrier per stage), BF16 tiles (7 KiB)
#endif
#ifndef HVS_STAGE_I8
#define HVS_STAGE_I8 8        // the same for INT8 tiles (4 KiB): 2 x 33 KiB of LDS per workgroup
#endif
#ifndef HVS_RADIX_LAST
#define HVS_RADIX_LAST 4u      // the last level multiplies the rows a query has seen by this (a power of two) ...
#endif
#ifndef HVS_RADIX_MID
#define HVS_RADIX_MID 16u      // ... and every level before it by this (see "Guessed thresholds" at hvs_k_merge)
#endif
#ifndef HVS_WG_WAVES
#define HVS_WG_WAVES 4        // waves (= query groups) per filter workgroup sharing one tile stream
#endif
#ifndef HVS_FILTER_OCC
#define HVS_FILTER_OCC 2      // waves per SIMD the filter kernel is compiled for
#endif
#ifndef HVS_RESCORE_WAVES
#define HVS_RESCORE_WAVES 8    // waves per re-scoring block (they share the group's queries in LDS)
#endif
#ifndef HVS_RESCORE_UNROLL
#define HVS_RESCORE_UNROLL 2   // re-scoring: rows in flight per wave = 8 x this (16: one group of 16 pairs per pass; 32 and 64
                               // measured 0.6 % and 8 % slower on the bench: more registers, fewer waves)
#endif

typedef __bf16 hvs_bf16x8 __attribute__((ext_vector_type(8)));
typedef float hvs_f32x16 __attribute__((ext_vector_type(16)));
typedef int hvs_i32x4 __attribute__((ext_vector_type(4)));
typedef int hvs_i32x16 __attribute__((ext_vector_type(16)));

// Tile formats of the filter (see "INT8 filter" below).  The level/slot machinery is shared.
#define HVS_FMT_NONE 0
#define HVS_FMT_BF16 1
#define HVS_FMT_I8 2
#define HVS_FMT_I8X16 3        // INT8 operands laid out for v_mfma_i32_16x16x64_i8 (same quantisation and bound as HVS_FMT_I8)
#define HVS_FMT_F16 4          // the BF16 layout and bound with IEEE half operands (v_mfma_f32_32x32x16_f16): 11 instead of 8
                               // significant bits per element -- an 8x tighter band at the BF16 filter's cost, for data whose
                               // neighbour distances are small against its extent (clusters, few dominant dimensions)
#define HVS_IS_I8(fmt) ((fmt) == HVS_FMT_I8 || (fmt) == HVS_FMT_I8X16)
#define HVS_IS_H16(fmt) ((fmt) == HVS_FMT_BF16 || (fmt) == HVS_FMT_F16)
#define HVS_I8_KSTEPS 4        // 4 MFMA k-steps of 32 (K padded 100 -> 128)
#define HVS_I8_KMEM 3          // k-steps stored as full fragments (dims 0..95); the 4th holds 4 real dimensions
#define HVS_I8_TILE_U4 (HVS_I8_KMEM * 64)  // uint4 per 32-row INT8 tile (3 KiB)
#define HVS_I8_NRM_U4 16       // uint4 of side data per tile: [0,8) dims 96..99 of the 32 rows (4 x int8 each),
                               // [8,16) the rows' accumulator inits (32 x int32)
#define HVS_I8_PAD_NORM (-(1 << 30))  // accumulator init of a padding row: can never reach a threshold
// HVS_FMT_I8X16: a 32-row tile = 4 fragments of 64 lanes x 16 B (row block rb = 0,1 x k-step ks = 0,1; fragment
// f = 2 rb + ks; lane l: row 16 rb + (l & 15), k = 64 ks + 16 (l >> 4) + 0..15; K padded 100 -> 128 with zeros)
// = 4 KiB, plus the 32 accumulator inits (8 uint4) as side data
#define HVS_I8X16_FRAGS 4
#define HVS_I8X16_TILE_U4 (HVS_I8X16_FRAGS * 64)
#define HVS_I8X16_NRM_U4 8
#define HVS_I8X16_QSUB 16      // queries per B-operand fragment (8 sub-blocks per 128-query group)

// ---------------------------------------------------------------------------------------------
// order-preserving integer keys of f32 attributes.  -0 is folded into +0 (they compare equal in
// every predicate), NaN maps to the largest key (it fails every comparison of the reference and
// must stay outside every non-trivial range).
// ---------------------------------------------------------------------------------------------
__host__ __device__ static inline uint32_t hvs_attr_key(float f)
{
    if (f != f) return 0xFFFFFFFFu;
    if (f == 0.0f) f = 0.0f;  // -0 -> +0
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t u = __float_as_uint(f);
#else
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ uint16_t hvs_bf16_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);  // round to nearest even
}
__device__ __forceinline__ float hvs_bf16_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
// IEEE half, round to nearest even (values beyond 65504 become inf: such data fails the format's finiteness check)
__device__ __forceinline__ uint16_t hvs_f16_bits(float f)
{
    const _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float hvs_f16_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
// element of a 16-bit float tile format
__device__ __forceinline__ uint16_t hvs_h16_bits(int fmt, float f) { return fmt == HVS_FMT_F16 ? hvs_f16_bits(f) : hvs_bf16_bits(f); }
__device__ __forceinline__ float hvs_h16_to_f32(int fmt, uint16_t b) { return fmt == HVS_FMT_F16 ? hvs_f16_to_f32(b) : hvs_bf16_to_f32(b); }
// The matrix pipe may flush half-precision denormals (|x| < 2^-14) to zero; the bound then sees an operand that differs
// from the converted value by up to 2^-14 per element: HVS_F16_FLUSH = sqrt(100) 2^-14 is added to every error norm of the
// F16 format (rows, queries) and 3 x 2^-14 to rho (the three pieces of -|d|^2/2), so the bound holds either way.
#define HVS_F16_FLUSH 6.103515625e-4
#define HVS_F16_FLUSH_RHO 1.8310546875e-4

__device__ __forceinline__ float hvs_round_up_f32(double x)
{
    float f = (float)x;
    if ((double)f < x) f = __uint_as_float(__float_as_uint(f) + 1u);  // x > 0 here
    return f;
}

__device__ __forceinline__ void hvs_atomic_max_pos(float* addr, float v)  // v >= 0
{
    atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// ---------------------------------------------------------------------------------------------
// level arithmetic (see header comment).  g4(t) = #{u in [0,t) : u % 4 != 0}.
// ---------------------------------------------------------------------------------------------
struct HvsLevels {
    uint32_t K;             // last level index
    uint32_t nblk;          // blocks in the ordering
    uint32_t off[16];       // storage offset of each level; off[K+1] = nblk
    uint32_t stride[16];    // level j holds the multiples of stride[j] ...
    uint32_t radix[16];     // ... that are not multiples of stride[j-1] = stride[j] * radix[j]  (j >= 1)
    uint32_t shift[16];     // log2(stride[j]) and
    uint32_t rshift[16];    // log2(radix[j]) when strides and radices are powers of two (pow2 != 0): the runs need no division
    uint32_t pow2;
};

__host__ __device__ static inline uint32_t hvs_ceil_div(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }
// #{u in [0,t) : u % r != 0}
__host__ __device__ static inline uint32_t hvs_gr(uint32_t t, uint32_t r) { return t - hvs_ceil_div(t, r); }

// storage run [lo,hi) (absolute storage indices) of level j inside block range [blo,bhi)
__host__ __device__ static inline void hvs_level_run(const HvsLevels& L, uint32_t j, uint32_t blo, uint32_t bhi,
                                                    uint32_t& lo, uint32_t& hi)
{
    if (bhi <= blo) {
        lo = hi = 0;
        return;
    }
    const uint32_t s = L.stride[j];
    if (L.pow2) {  // power-of-two radices (the default tables): shifts; #{u in [0,t) : u % r != 0} = t - ceil(t / r)
        const uint32_t sh = L.shift[j], rs = L.rshift[j], r1 = (1u << rs) - 1u;
        const uint32_t tlo = (blo + s - 1u) >> sh, thi = (bhi + s - 1u) >> sh;
        lo = L.off[j] + (j == 0 ? tlo : tlo - ((tlo + r1) >> rs));
        hi = L.off[j] + (j == 0 ? thi : thi - ((thi + r1) >> rs));
        return;
    }
    const uint32_t tlo = hvs_ceil_div(blo, s), thi = hvs_ceil_div(bhi, s);
    if (j == 0) {
        lo = L.off[0] + tlo;
        hi = L.off[0] + thi;
    } else {
        lo = L.off[j] + hvs_gr(tlo, L.radix[j]);
        hi = L.off[j] + hvs_gr(thi, L.radix[j]);
    }
}

// storage index -> source block
__host__ __device__ static inline uint32_t hvs_storage_to_block(const HvsLevels& L, uint32_t idx)
{
    uint32_t j = 0;
    while (j < L.K && idx >= L.off[j + 1]) ++j;
    const uint32_t i = idx - L.off[j];
    if (j == 0) return i * L.stride[0];
    return (i + i / (L.radix[j] - 1u) + 1u) * L.stride[j];  // the i-th positive integer that is not a multiple of radix
}

// level table of an ordering with n rows (host).  Radices (powers of two) from the last level backwards: r_last, then
// r_mid, ..., each cut down so that level 0 keeps >= 16 blocks (512 rows): n = 10^7 -> level 0 of 19 blocks, then
// radices 16, 16, 16, 4.  r_last = r_mid = 2 gives round 2's doubling levels (up to 14 of them).
// `plan` (optional, A/B runs): radices of the last levels, last level first, 0-terminated; r_mid continues behind it
static inline HvsLevels hvs_make_levels(uint32_t n, uint32_t r_last = HVS_RADIX_LAST, uint32_t r_mid = HVS_RADIX_MID,
                                        const uint32_t* plan = nullptr)
{
    HvsLevels L{};
    L.nblk = (n + 31u) / 32u;
    uint32_t rad[16];
    uint32_t K = 0, S = 1;
    bool in_plan = plan != nullptr;
    for (;;) {
        uint32_t r = K == 0 ? r_last : r_mid;
        if (in_plan && plan[K] == 0u) in_plan = false;
        if (in_plan) r = plan[K];
        while (r >= 2u && (uint64_t)L.nblk / ((uint64_t)S * r) < 16u) r >>= 1;
        if (r < 2u || K >= 14u) break;
        rad[K++] = r;
        S *= r;
    }
    L.K = K;
    L.stride[K] = 1;
    L.radix[0] = 1;
    for (uint32_t j = K; j >= 1; --j) {
        L.radix[j] = rad[K - j];
        L.stride[j - 1] = L.stride[j] * L.radix[j];
    }
    uint32_t off = 0;
    for (uint32_t j = 0; j <= K; ++j) {
        L.off[j] = off;
        const uint32_t t = hvs_ceil_div(L.nblk, L.stride[j]);
        off += (j == 0) ? t : hvs_gr(t, L.radix[j]);
    }
    for (uint32_t j = K + 1; j < 16u; ++j) {
        L.off[j] = off;
        L.stride[j] = 1;
        L.radix[j] = 2;
    }
    L.pow2 = 1u;
    for (uint32_t j = 0; j < 16u; ++j) {
        uint32_t sh = 0, rs = 0;
        while ((1u << sh) < L.stride[j]) ++sh;
        while ((1u << rs) < L.radix[j]) ++rs;
        L.shift[j] = sh;
        L.rshift[j] = rs;
        if ((1u << sh) != L.stride[j] || (1u << rs) != L.radix[j]) L.pow2 = 0u;
    }
    return L;
}
// ---------------------------------------------------------------------------------------------
// Index build

// ---------------------------------------------------------------------------------------------
__global__ void hvs_k_attr_keys(const float* __restrict__ D, uint32_t n, uint64_t* __restrict__ keys_ct,
                                uint64_t* __restrict__ keys_t, uint32_t* __restrict__ ids)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t kc = hvs_attr_key(D[(size_t)i * HVS_DCOLS]);
    const uint32_t kt = hvs_attr_key(D[(size_t)i * HVS_DCOLS + 1]);
    keys_ct[i] = ((uint64_t)kc << 32) | kt;
    keys_t[i] = ((uint64_t)kt << 32) | i;  // full 64-bit sort; the row id only breaks ties
    ids[i] = i;
}

struct HvsBounds {  // global maxima over rows, all rounded up
    float e_d;    // max |d - bf16(d)|_2
    float nb_d;   // max |bf16(d)|_2
    float hmax;   // max |d|^2 / 2
    float rho;    // max | |d|^2/2 + (h0+h1+h2) |
    // INT8 format (d' = d - center, dq = int8 image of d'):
    float e_d8;   // max |d' - sd dq|_2
    float n_d8;   // max |d'|_2
    // planner sample only: the F16 format's counterparts of e_d / nb_d
    float e_df, nb_df;
    // planner sample: sum and sum of squares of |a-b|^2 over sampled pairs of rows, and their number
    double pair_sum, pair_sumsq;
    uint32_t pair_n;
};

// ---------------------------------------------------------------------------------------------
// INT8 filter.  Distances are translation invariant, so rows and queries are taken relative to a
// per-dimension centre c (midrange of D):  d' = d - c, q' = q - c, and quantised with ONE scale sd for
// rows and queries alike:  dq = clamp(rint(d'/sd), +-127) (no row is clipped: sd = max |d'_k| / 127),
// qq likewise (queries outside D's bounding box are clipped; their error e_q then carries it).
//   q'.d' = sd^2 qq.dq + (sd qq).(d' - sd dq) + (q' - sd qq).d'
//   s := q'.d' - |d'|^2/2  <=  sd^2 (qq.dq + nh + 1) + |sd qq| E_D + e_q N_D,   nh = floor(-|d'|^2 / (2 sd^2))
// v_mfma_i32_32x32x32_i8 started from the accumulator init nh yields S = qq.dq + nh EXACTLY (no
// accumulation error term), so a row is discarded iff  S < theta_i := floor((|q'|^2 - tau(1+2g))/(2 sd^2)
// - band/sd^2) - 2  with band = |sd qq| E_D + e_q N_D  (hvs_k_merge).  For data that fills its bounding
// box evenly the band is about the BF16 one (BF16 of uncentred data pays |q| ~ 2x |q'|); heavy-tailed
// data widens it, which only costs re-scoring work, never correctness -- the planner (hvs.hip,
// choose_format) estimates that cost from a sample of row pairs.
// ---------------------------------------------------------------------------------------------
// Rotated INT8 tiles (round 4; HVS_FMT_I8X16 only).  One scale for all dimensions is set by the widest one: vectors whose
// variance sits in a few dimensions (PCA-like spectra) spend the 8 bits of every other dimension on nothing and the band
// |sd qq| E_D + e_q N_D grows with sd sqrt(100).  Distances are invariant under y = R x with R^T R = I, so rows and queries
// are first mapped to 128 dimensions by  y = H S pad(x) / sqrt(128)  (S = fixed signs, H = the 128-point Walsh-Hadamard
// matrix, pad = 28 zeros: the K slots the INT8 fragments pad with anyway): every y_k is a signed average of all 100
// components, the per-dimension ranges come out nearly equal, and the same bound (centre, one scale, exact integer chain,
// clip term) holds verbatim on the y's -- R has orthonormal columns, so |R q - R d| = |q - d| in real arithmetic; the y's are
// evaluated in f64 (relative error ~1e-14, inside the 1e-9 slack hvs_k_merge keeps for its own f64 evaluation).
// Measured (profiles/r04/nonuniform_int8.txt): PCA-like data at k/n = 10^-5, rows a band lets through per 30 wanted:
// 456 plain INT8, 151 rotated, 31 FP16; uniform data 48 / 135 / 31 -- the planner's probe decides per data set.
#define HVS_RDIM 128
// sign of input dimension j (a fixed pseudo-random pattern)
__host__ __device__ static inline uint32_t hvs_rot_sign(uint32_t j) { return ((j + 1u) * 0x9E3779B1u >> 17) & 1u; }
// component k of the rotated image of the 100-vector x
template <typename V>
__device__ __forceinline__ double hvs_rot_elem(const V& x, uint32_t k)
{
    double a = 0.0;
#pragma unroll 4
    for (uint32_t j = 0; j < HVS_NDIM; ++j) {
        const double v = (double)x[j];
        a += ((uint32_t)__popc(k & j) + hvs_rot_sign(j)) & 1u ? -v : v;
    }
    return a * 0.08838834764831844055;  // 1 / sqrt(128)
}
// f32 neighbours of a double, rounded outward (the box of the rotated data must contain every rotated row)
__device__ __forceinline__ float hvs_f32_below(double y)
{
    float f = (float)y;
    if ((double)f > y) f = nextafterf(f, -__builtin_inff());
    return f;
}
__device__ __forceinline__ float hvs_f32_above(double y)
{
    float f = (float)y;
    if ((double)f < y) f = nextafterf(f, __builtin_inff());
    return f;
}

struct HvsQuant {
    float center[HVS_RDIM];   // [0, 100) of the vector components, or [0, 128) of the rotated ones (`rot`)
    double sd;       // scale; 0 or non-finite: format unusable
    double inv_sd;
    uint32_t kmin[HVS_RDIM], kmax[HVS_RDIM];  // per-dimension min / max as order-preserving keys (hvs_attr_key)
    uint32_t rot;    // 1: centre / scale / tiles / fragments live in the rotated space
};

__host__ __device__ static inline float hvs_attr_key_inv(uint32_t k)
{
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}

__global__ void hvs_k_quant_reset(HvsQuant* __restrict__ qz, uint32_t rot)
{
    const uint32_t k = threadIdx.x;
    if (k < HVS_RDIM) {
        qz->kmin[k] = 0xFFFFFFFFu;
        qz->kmax[k] = 0u;
        qz->center[k] = 0.0f;
    }
    if (k == 0u) qz->rot = rot;
}

// the same in the rotated space: blockDim = 128 (thread = rotated dimension), rows strided over blocks and staged in LDS
__global__ __launch_bounds__(128) void hvs_k_minmax_rot(const float* __restrict__ D, uint32_t n, HvsQuant* __restrict__ qz)
{
    __shared__ float srow[HVS_NDIM];
    const uint32_t k = threadIdx.x;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        __syncthreads();
        if (k < HVS_NDIM) srow[k] = D[(size_t)i * HVS_DCOLS + 2 + k];
        __syncthreads();
        const double y = hvs_rot_elem(srow, k);
        const uint32_t klo = hvs_attr_key(hvs_f32_below(y)), khi = hvs_attr_key(hvs_f32_above(y));
        lo = klo < lo ? klo : lo;
        hi = khi > hi ? khi : hi;
    }
    atomicMin(&qz->kmin[k], lo);
    atomicMax(&qz->kmax[k], hi);
}

// per-dimension min / max of the vector components; blockDim = 128 (thread = dimension), rows strided over blocks
__global__ __launch_bounds__(128) void hvs_k_minmax(const float* __restrict__ D, uint32_t n, HvsQuant* __restrict__ qz)
{
    const uint32_t k = threadIdx.x;
    if (k >= HVS_NDIM) return;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t key = hvs_attr_key(D[(size_t)i * HVS_DCOLS + 2 + k]);
        lo = key < lo ? key : lo;
        hi = key > hi ? key : hi;
    }
    atomicMin(&qz->kmin[k], lo);
    atomicMax(&qz->kmax[k], hi);
}

__global__ void hvs_k_quant_params(HvsQuant* __restrict__ qz)
{
    __shared__ double half[HVS_RDIM];
    const uint32_t k = threadIdx.x;
    const uint32_t ndim = qz->rot ? HVS_RDIM : HVS_NDIM;
    if (k < HVS_RDIM) half[k] = 0.0;
    if (k < ndim) {
        const double lo = (double)hvs_attr_key_inv(qz->kmin[k]), hi = (double)hvs_attr_key_inv(qz->kmax[k]);  // NaN key -> NaN
        const float c = (float)(0.5 * (lo + hi));
        qz->center[k] = c;
        const double a = fabs(lo - (double)c), b = fabs(hi - (double)c);
        half[k] = (a != a || b != b) ? (double)__builtin_inff() : (a > b ? a : b);
    }
    __syncthreads();
    if (k == 0u) {
        double m = 0.0;
        for (int i = 0; i < HVS_RDIM; ++i) m = half[i] > m ? half[i] : m;
        // no row is clipped; all rows equal: any scale works
        double sd = m > 0.0 ? m / 127.0 * (1.0 + 1e-9) : 1.0;
        if (!(sd < 1.0e18) || sd < 1.0e-18) sd = 0.0;  // non-finite or extreme ranges: INT8 format unusable
        qz->sd = sd;
        qz->inv_sd = sd > 0.0 ? 1.0 / sd : 0.0;
    }
}

__device__ __forceinline__ int hvs_quant_i8(double x, double inv_sd)
{
    double r = rint(x * inv_sd);
    r = r > 127.0 ? 127.0 : (r < -127.0 ? -127.0 : r);
    return (r == r) ? (int)r : 0;
}

// one wave per storage block: 3 KiB INT8 A-operand tile (lane l of k-step s: row l&31, k = 32 s + 16 (l>>5) + 0..15;
// k-steps 0..2), 256 B of side data (dims 96..99 and the 32 accumulator inits) and the row bounds
__global__ __launch_bounds__(256) void hvs_k_build_tiles_i8(const float* __restrict__ D, uint32_t n,
                                                            const uint32_t* __restrict__ perm, HvsLevels L,
                                                            const HvsQuant* __restrict__ qz, uint4* __restrict__ tiles,
                                                            int* __restrict__ norms, uint32_t* __restrict__ blockpos,
                                                            HvsBounds* __restrict__ bounds)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t idx = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (idx >= L.nblk) return;
    const uint32_t b = hvs_storage_to_block(L, idx);
    if (lane == 0u) blockpos[idx] = b;
    const uint32_t r = lane & 31u, h = lane >> 5;
    const uint32_t pos = b * 32u + r;
    const bool valid = pos < n;
    const float* __restrict__ row = D + (size_t)(valid ? perm[pos] : 0u) * HVS_DCOLS + 2;
    const double sd = qz->sd, inv_sd = qz->inv_sd;
    if (h == 0u) {
        double nd = 0.0, e2 = 0.0;
        for (int k = 0; k < HVS_NDIM; ++k) {
            const double x = (double)row[k] - (double)qz->center[k];
            const double xq = sd * (double)hvs_quant_i8(x, inv_sd);
            nd += x * x;
            e2 += (x - xq) * (x - xq);
        }
        int nh = HVS_I8_PAD_NORM;
        if (valid) {
            const double v = floor(-0.5 * nd * inv_sd * inv_sd);
            nh = v > -1.0e9 ? (int)v : HVS_I8_PAD_NORM;  // (|d'_k| <= 127 sd: v >= -806450)
            hvs_atomic_max_pos(&bounds->e_d8, hvs_round_up_f32(sqrt(e2) * (1.0 + 1e-9) + 1e-30));
            hvs_atomic_max_pos(&bounds->n_d8, hvs_round_up_f32(sqrt(nd) * (1.0 + 1e-9) + 1e-30));
        }
        // side data: the 4 real dimensions of the 4th k-step (the other 28 are zero padding and are not stored:
        // the filter rebuilds the fragment from this word), then the accumulator init
        uint32_t tail = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int v = valid ? hvs_quant_i8((double)row[96 + e] - (double)qz->center[96 + e], inv_sd) : 0;
            tail |= ((uint32_t)v & 0xFFu) << (8 * e);
        }
        norms[(size_t)idx * 64u + r] = (int)tail;
        norms[(size_t)idx * 64u + 32u + r] = nh;
    }
#pragma unroll
    for (int s = 0; s < HVS_I8_KMEM; ++s) {
        uint32_t w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint32_t word = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 32 * s + 16 * (int)h + 4 * p + e;
                int v = 0;
                if (valid && k < HVS_NDIM) v = hvs_quant_i8((double)row[k] - (double)qz->center[k], inv_sd);
                word |= ((uint32_t)v & 0xFFu) << (8 * e);
            }
            w[p] = word;
        }
        tiles[((size_t)idx * HVS_I8_KMEM + s) * 64u + lane] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// HVS_FMT_I8X16: one wave per storage block.  Same quantisation, accumulator inits and bounds as
// hvs_k_build_tiles_i8; only the operand layout differs (v_mfma_i32_16x16x64_i8: lane l of fragment (rb, ks) holds
// row 16 rb + (l & 15), k = 64 ks + 16 (l >> 4) + 0..15).
__global__ __launch_bounds__(256) void hvs_k_build_tiles_i8x16(const float* __restrict__ D, uint32_t n,
                                                               const uint32_t* __restrict__ perm, HvsLevels L,
                                                               const HvsQuant* __restrict__ qz, uint4* __restrict__ tiles,
                                                               int* __restrict__ norms, uint32_t* __restrict__ blockpos,
                                                               HvsBounds* __restrict__ bounds)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t idx = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (idx >= L.nblk) return;
    const uint32_t b = hvs_storage_to_block(L, idx);
    if (lane == 0u) blockpos[idx] = b;
    const double sd = qz->sd, inv_sd = qz->inv_sd;
    if (lane < 32u) {  // lane = row: norm term and row bounds
        const uint32_t pos = b * 32u + lane;
        const bool valid = pos < n;
        const float* __restrict__ row = D + (size_t)(valid ? perm[pos] : 0u) * HVS_DCOLS + 2;
        double nd = 0.0, e2 = 0.0;
        for (int k = 0; k < HVS_NDIM; ++k) {
            const double x = (double)row[k] - (double)qz->center[k];
            const double xq = sd * (double)hvs_quant_i8(x, inv_sd);
            nd += x * x;
            e2 += (x - xq) * (x - xq);
        }
        int nh = HVS_I8_PAD_NORM;
        if (valid) {
            const double v = floor(-0.5 * nd * inv_sd * inv_sd);
            nh = v > -1.0e9 ? (int)v : HVS_I8_PAD_NORM;
            hvs_atomic_max_pos(&bounds->e_d8, hvs_round_up_f32(sqrt(e2) * (1.0 + 1e-9) + 1e-30));
            hvs_atomic_max_pos(&bounds->n_d8, hvs_round_up_f32(sqrt(nd) * (1.0 + 1e-9) + 1e-30));
        }
        norms[(size_t)idx * 32u + lane] = nh;
    }
#pragma unroll
    for (int f = 0; f < HVS_I8X16_FRAGS; ++f) {
        const int rbk = f >> 1, ks = f & 1;
        const uint32_t pos = b * 32u + 16u * (uint32_t)rbk + (lane & 15u);
        const bool valid = pos < n;
        const float* __restrict__ row = D + (size_t)(valid ? perm[pos] : 0u) * HVS_DCOLS + 2;
        uint32_t w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint32_t word = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 64 * ks + 16 * (int)(lane >> 4) + 4 * p + e;
                int v = 0;
                if (valid && k < HVS_NDIM) v = hvs_quant_i8((double)row[k] - (double)qz->center[k], inv_sd);
                word |= ((uint32_t)v & 0xFFu) << (8 * e);
            }
            w[p] = word;
        }
        tiles[((size_t)idx * HVS_I8X16_FRAGS + f) * 64u + lane] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// HVS_FMT_I8X16 tiles of the ROTATED rows (see HvsQuant): one wave per storage block.  The block's 32 rows are staged in LDS,
// every rotated component is evaluated once (lane = 64 of the block's 4096 components per pass) and kept as int8 in an LDS
// image the fragments are then cut from; accumulator inits and row bounds as in hvs_k_build_tiles_i8x16, over 128 dimensions.
__global__ __launch_bounds__(256) void hvs_k_build_tiles_i8x16_rot(const float* __restrict__ D, uint32_t n,
                                                                   const uint32_t* __restrict__ perm, HvsLevels L,
                                                                   const HvsQuant* __restrict__ qz, uint4* __restrict__ tiles,
                                                                   int* __restrict__ norms, uint32_t* __restrict__ blockpos,
                                                                   HvsBounds* __restrict__ bounds)
{
    __shared__ float srow[4][32][HVS_NDIM];
    __shared__ __attribute__((aligned(16))) signed char s8[4][32][HVS_RDIM];
    __shared__ double ssum[4][32][2];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t idx = blockIdx.x * 4u + w;
    const bool live = idx < L.nblk;  // (no early return: the workgroup barriers below)
    const uint32_t b = live ? hvs_storage_to_block(L, idx) : 0u;
    if (live && lane == 0u) blockpos[idx] = b;
    for (uint32_t e = lane; e < 32u * HVS_NDIM; e += 64u) {
        const uint32_t r = e / HVS_NDIM, c = e % HVS_NDIM;
        const uint32_t pos = b * 32u + r;
        srow[w][r][c] = (live && pos < n) ? D[(size_t)perm[pos] * HVS_DCOLS + 2u + c] : 0.0f;
    }
    if (lane < 32u) ssum[w][lane][0] = ssum[w][lane][1] = 0.0;
    __syncthreads();
    const double sd = qz->sd, inv_sd = qz->inv_sd;
    for (uint32_t e = lane; e < 32u * HVS_RDIM; e += 64u) {
        const uint32_t r = e >> 7, k = e & 127u;
        const bool valid = live && b * 32u + r < n;
        const double x = hvs_rot_elem(srow[w][r], k) - (double)qz->center[k];
        const int v = valid ? hvs_quant_i8(x, inv_sd) : 0;
        s8[w][r][k] = (signed char)v;
        if (valid) {
            const double xq = sd * (double)v;
            atomicAdd(&ssum[w][r][0], x * x);
            atomicAdd(&ssum[w][r][1], (x - xq) * (x - xq));
        }
    }
    __syncthreads();
    if (!live) return;
    if (lane < 32u) {
        const bool valid = b * 32u + lane < n;
        int nh = HVS_I8_PAD_NORM;
        if (valid) {
            const double nd = ssum[w][lane][0], e2 = ssum[w][lane][1];
            const double v = floor(-0.5 * nd * inv_sd * inv_sd);
            nh = v > -1.0e9 ? (int)v : HVS_I8_PAD_NORM;
            hvs_atomic_max_pos(&bounds->e_d8, hvs_round_up_f32(sqrt(e2) * (1.0 + 1e-9) + 1e-30));
            hvs_atomic_max_pos(&bounds->n_d8, hvs_round_up_f32(sqrt(nd) * (1.0 + 1e-9) + 1e-30));
        }
        norms[(size_t)idx * 32u + lane] = nh;
    }
#pragma unroll
    for (int f = 0; f < HVS_I8X16_FRAGS; ++f) {
        const int rbk = f >> 1, ks = f & 1;
        const uint4 v = *reinterpret_cast<const uint4*>(&s8[w][16 * rbk + (lane & 15u)][64 * ks + 16 * (int)(lane >> 4)]);
        tiles[((size_t)idx * HVS_I8X16_FRAGS + f) * 64u + lane] = v;
    }
}

// Planner inputs in one pass over a sample of rows (stride `step`): the bounds BOTH formats would get
// (from the sampled rows only -- estimates, the formats' own build kernels compute the real maxima) and
// the spread of squared distances between sampled row pairs.  One lane per sampled row.
__global__ void hvs_k_plan_stats(const float* __restrict__ D, uint32_t n, uint32_t step, const HvsQuant* __restrict__ qz,
                                 HvsBounds* __restrict__ bounds)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i64 = (uint64_t)t * step;
    if (i64 >= n) return;
    const uint32_t i = (uint32_t)i64;
    const uint32_t j = (uint32_t)(((uint64_t)i * 2654435761ull + 40503ull) % n);
    const float* __restrict__ a = D + (size_t)i * HVS_DCOLS + 2;
    const float* __restrict__ bb = D + (size_t)j * HVS_DCOLS + 2;
    const double sd = qz->sd, inv_sd = qz->inv_sd;
    double nd = 0.0, e2 = 0.0, nb2 = 0.0, nd8 = 0.0, e28 = 0.0, dist = 0.0, e2f = 0.0, nb2f = 0.0;
    for (int k = 0; k < HVS_NDIM; ++k) {
        const float x = a[k];
        const float xb = hvs_bf16_to_f32(hvs_bf16_bits(x));
        const float xf = hvs_f16_to_f32(hvs_f16_bits(x));
        nd += (double)x * (double)x;
        e2 += ((double)x - (double)xb) * ((double)x - (double)xb);
        nb2 += (double)xb * (double)xb;
        e2f += ((double)x - (double)xf) * ((double)x - (double)xf);
        nb2f += (double)xf * (double)xf;
        const double x8 = (double)x - (double)qz->center[k];
        const double xq = sd * (double)hvs_quant_i8(x8, inv_sd);
        nd8 += x8 * x8;
        e28 += (x8 - xq) * (x8 - xq);
        const double df = (double)x - (double)bb[k];
        dist += df * df;
    }
    if (!(nd < 1.0e30)) return;
    hvs_atomic_max_pos(&bounds->e_d, hvs_round_up_f32(sqrt(e2)));
    hvs_atomic_max_pos(&bounds->nb_d, hvs_round_up_f32(sqrt(nb2)));
    hvs_atomic_max_pos(&bounds->hmax, hvs_round_up_f32(0.5 * nd));
    hvs_atomic_max_pos(&bounds->e_d8, hvs_round_up_f32(sqrt(e28)));
    hvs_atomic_max_pos(&bounds->n_d8, hvs_round_up_f32(sqrt(nd8)));
    if (nb2f < 1.0e30) {
        hvs_atomic_max_pos(&bounds->e_df, hvs_round_up_f32(sqrt(e2f) + HVS_F16_FLUSH));
        hvs_atomic_max_pos(&bounds->nb_df, hvs_round_up_f32(sqrt(nb2f)));
    } else {
        bounds->nb_df = __builtin_inff();  // a component beyond the half-precision range: F16 unusable
    }
    if (i != j && dist < 1.0e30) {
        atomicAdd(&bounds->pair_sum, dist);
        atomicAdd(&bounds->pair_sumsq, dist * dist);
        atomicAdd(&bounds->pair_n, 1u);
    }
}

// one wave per storage block: builds the 7 KiB A-operand tile and the row bounds
// (`fmt`: HVS_FMT_BF16 or HVS_FMT_F16 -- the element type of the 16-bit float layout)
__global__ __launch_bounds__(256) void hvs_k_build_tiles(const float* __restrict__ D, uint32_t n,
                                                         const uint32_t* __restrict__ perm, HvsLevels L,
                                                         uint4* __restrict__ tiles, uint32_t* __restrict__ blockpos,
                                                         HvsBounds* __restrict__ bounds, int fmt)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t idx = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (idx >= L.nblk) return;
    const uint32_t b = hvs_storage_to_block(L, idx);
    if (lane == 0u) blockpos[idx] = b;
    const uint32_t r = lane & 31u, h = lane >> 5;
    const uint32_t pos = b * 32u + r;
    const bool valid = pos < n;
    const float* __restrict__ row = D + (size_t)(valid ? perm[pos] : 0u) * HVS_DCOLS + 2;

    double nd = 0.0, e2 = 0.0, nb2 = 0.0;
    for (int k = 0; k < HVS_NDIM; ++k) {
        const float x = row[k];
        const float xb = hvs_h16_to_f32(fmt, hvs_h16_bits(fmt, x));
        nd += (double)x * (double)x;
        e2 += ((double)x - (double)xb) * ((double)x - (double)xb);
        nb2 += (double)xb * (double)xb;
    }
    // h = -|d|^2/2 as three 16-bit pieces; invalid (padding) rows get a huge negative bias (the largest finite half in
    // the F16 format, whose thresholds never come near it: test on the row's validity keeps padding rows out anyway)
    float hf = valid ? (float)(-0.5 * nd) : (fmt == HVS_FMT_F16 ? -65504.0f : -1.0e30f);
    const uint16_t h0 = hvs_h16_bits(fmt, hf);
    const float r1 = valid ? hf - hvs_h16_to_f32(fmt, h0) : 0.0f;
    const uint16_t h1 = hvs_h16_bits(fmt, r1);
    const float r2 = r1 - hvs_h16_to_f32(fmt, h1);
    const uint16_t h2 = hvs_h16_bits(fmt, r2);
    if (valid && h == 0u) {
        const double hs = (double)hvs_h16_to_f32(fmt, h0) + (double)hvs_h16_to_f32(fmt, h1) + (double)hvs_h16_to_f32(fmt, h2);
        const double flush = fmt == HVS_FMT_F16 ? HVS_F16_FLUSH : 0.0, flush_rho = fmt == HVS_FMT_F16 ? HVS_F16_FLUSH_RHO : 0.0;
        hvs_atomic_max_pos(&bounds->e_d, hvs_round_up_f32(sqrt(e2) + flush));
        hvs_atomic_max_pos(&bounds->nb_d, hvs_round_up_f32(sqrt(nb2)));
        hvs_atomic_max_pos(&bounds->hmax, hvs_round_up_f32(0.5 * nd));
        hvs_atomic_max_pos(&bounds->rho, hvs_round_up_f32(fabs(0.5 * nd + hs) + flush_rho + 1e-30));
    }
#pragma unroll
    for (int s = 0; s < HVS_KSTEPS; ++s) {
        uint32_t w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint16_t lo16, hi16;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 16 * s + 8 * (int)h + 2 * p + e;
                uint16_t v;
                if (k < HVS_NDIM)
                    v = valid ? hvs_h16_bits(fmt, row[k]) : (uint16_t)0;
                else if (k == 100)
                    v = h0;
                else if (k == 101)
                    v = h1;
                else if (k == 102)
                    v = h2;
                else
                    v = 0;
                if (e == 0)
                    lo16 = v;
                else
                    hi16 = v;
            }
            w[p] = (uint32_t)lo16 | ((uint32_t)hi16 << 16);
        }
        tiles[((size_t)idx * HVS_KSTEPS + s) * 64u + lane] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// Per-batch query state (structure of arrays over PADDED SLOTS; slot = block*32 + i)
// ---------------------------------------------------------------------------------------------
struct HvsBatch {
    // layout of the batch
    uint32_t nslots;            // padded slots (multiple of HVS_GROUP)
    uint32_t ngroups;           // nslots / HVS_GROUP
    uint32_t* qid;              // [nslots] original query index or 0xFFFFFFFF
    uint32_t* rank;             // [nslots] predicate class: 0,1,2(type 3),3(invalid),4(type 2 -> T ordering)
    uint32_t* ra;               // [nslots] own position range [ra, rb)
    uint32_t* rb;
    uint32_t* gua;              // [ngroups] union range of the group
    uint32_t* gub;
    uint32_t* gord;             // [ngroups] ordering used by the group: 0 = (C,T), 1 = T
    // filter operands
    uint4* bfrag;               // [nslots/32][7][64] BF16 B-operand fragments ([nslots/32][4][64] in the INT8 format)
    float* theta;               // [nslots] BF16 format: discard a row when its MFMA value is < theta
    int* thetai;                // same storage, INT8 format: integer threshold
    double* qn;                 // [nslots] |q|^2
    float* normq;               // [nslots] BF16 format: |q| (rounded up); INT8 formats: the band's clip term (hvs_k_prep)
    float* eq;                  // [nslots] |q - bf16(q)| (rounded up)
    float* nqb;                 // [nslots] |bf16(q)| (rounded up)
    // top-k state
    uint64_t* top;              // [nslots][topcap]
    uint32_t topcap;            // stride of the stored top-k lists: 128 (k <= 128) or 256
    uint32_t knn;               // k of this batch (hvs_set_k; the reference's KNN_LIMIT, optimized_impl.h:26)
    uint32_t* topcnt;           // [nslots]
    float* tau;                 // [nslots]
    uint64_t* cand;             // [nslots][fcap]
    uint32_t fcap;              // candidate keys per slot and round: HVS_FCAP, more for batches that leave room (retry batches
                                // run every level with the proven threshold: k (radix - 1) candidates per level)
    uint32_t gcap;              // survivor entries per group and round (HVS_GCAP or more)
    uint32_t* candcnt;          // [nslots]
    uint32_t* overflow;         // [nslots] 0 = answered; HVS_FAIL_RETRY: run again with proven thresholds; HVS_FAIL_EXACT: exact engine
    uint32_t fail_code;         // what a capacity overflow / failed verification of THIS batch writes there (see hvs_flag_fail)
    // filter output
    uint64_t* pairs;            // [ngroups][gcap]  survivor entries (hvs_entry_make)
    uint32_t* paircnt;          // [ngroups]
    uint32_t* goverflow;        // [ngroups]
};

// A query the filter engine cannot answer in this batch.  HVS_FAIL_RETRY (batches with guessed thresholds): the query is
// run again in a batch whose last level uses a proven threshold; HVS_FAIL_EXACT (queries without a usable bound, and any
// failure inside such a retry batch): the exact engine answers it.  The larger code wins.
#define HVS_FAIL_RETRY 1u
#define HVS_FAIL_EXACT 2u
__device__ __forceinline__ void hvs_flag_fail(uint32_t code, uint32_t* __restrict__ overflow, uint32_t slot)
{
    // `code` is a kernel argument (wave-uniform): it stays in its scalar register up to this rare path -- hoisted into a vector
    // register in front of the row loop it cost hvs_k_seed_exact its 169th register, i.e. a spill
    asm volatile("" : "+s"(code));
    if (overflow[slot] < code) overflow[slot] = code;  // (racing writers of one batch all write the same code)
}

// Survivor entry of the filter (8 bytes in the group's pair list): the lanes of one (tile, query block) whose
// accumulators reached the threshold.  bits 0..15 accumulator mask (bit r = row (r & 3) + 8 (r >> 2) + 4 half of
// the block), 16..39 block position (24 bits: orderings of up to 2^29 rows -- more than one GPU's HBM holds; round 2's 22 bits
// stopped at 2^27 rows), 40 row half, 41..63 slot (23 bits: a batch has at most 2^21 queries + padding).
#define HVS_ENTRY_MAX_BLOCKS (1u << 24)
#define HVS_ENTRY_MAX_SLOTS (1u << 23)
__device__ __forceinline__ uint64_t hvs_entry_make(uint32_t slot, uint32_t bp, uint32_t half, uint32_t mask)
{
    const uint32_t lo = (bp << 16) | mask;
    const uint32_t hi = (slot << 9) | (half << 8) | (bp >> 16);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t hvs_entry_slot(uint64_t e) { return (uint32_t)(e >> 41); }
__device__ __forceinline__ uint32_t hvs_entry_mask(uint64_t e) { return (uint32_t)e & 0xFFFFu; }
// position of row bit r of the entry
__device__ __forceinline__ uint32_t hvs_entry_pos(uint64_t e, uint32_t r)
{
    const uint32_t bp = (uint32_t)(e >> 16) & (HVS_ENTRY_MAX_BLOCKS - 1u);
    const uint32_t half = (uint32_t)(e >> 40) & 1u;
    return bp * 32u + (r & 3u) + 8u * (r >> 2) + 4u * half;
}

// HVS_FMT_I8X16 entries: the lane of a 16x16 accumulator block holds ONE query and 4 consecutive rows, two row blocks per
// tile: bits 0..7 accumulator mask (bit b = 4 rb + i: row 16 rb + 4 quad + i of the block), 8..9 quad (lane >> 4),
// 16..39 block position, 41..63 slot.
__device__ __forceinline__ uint64_t hvs_entry16_make(uint32_t slot, uint32_t bp, uint32_t quad, uint32_t mask8)
{
    const uint32_t lo = (bp << 16) | (quad << 8) | mask8;
    const uint32_t hi = (slot << 9) | (bp >> 16);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t hvs_entry16_mask(uint64_t e) { return (uint32_t)e & 0xFFu; }
__device__ __forceinline__ uint32_t hvs_entry16_pos(uint64_t e, uint32_t b)
{
    const uint32_t bp = (uint32_t)(e >> 16) & (HVS_ENTRY_MAX_BLOCKS - 1u);
    const uint32_t quad = (uint32_t)(e >> 8) & 3u;
    return bp * 32u + 16u * (b >> 2) + 4u * quad + (b & 3u);
}

// class rank of a query type: (C,T)-ordering classes first, the T-ordering class (type 2) last
__device__ __forceinline__ uint32_t hvs_type_rank(uint32_t type)
{
    return type == 0u ? 0u : type == 1u ? 1u : type == 3u ? 2u : type == 2u ? 4u : 3u;
}

__device__ __forceinline__ uint32_t hvs_lower_bound64(const uint64_t* __restrict__ a, uint32_t n, uint64_t key)
{
    uint32_t lo = 0, hi = n;  // first index with a[i] >= key
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// position range [a,b) of a query's predicate in its ordering ((C,T) for types 0,1,3; T for type 2)
__device__ __forceinline__ void hvs_query_range(const HvsQParams& p, const uint64_t* __restrict__ keys_ct,
                                                const uint64_t* __restrict__ keys_t, uint32_t n, uint32_t& a, uint32_t& b)
{
    a = 0;
    b = 0;
    const uint32_t kv = hvs_attr_key(p.vf), kl = hvs_attr_key(p.l), kr = hvs_attr_key(p.r);
    const bool lr_ok = (p.l == p.l) && (p.r == p.r);  // NaN bounds match nothing
    if (p.type == 0u) {
        b = n;
    } else if (p.type == 1u) {
        a = hvs_lower_bound64(keys_ct, n, (uint64_t)kv << 32);
        b = hvs_lower_bound64(keys_ct, n, ((uint64_t)kv + 1ull) << 32);  // rows with T = NaN still match C==v
    } else if (p.type == 3u && lr_ok && kl <= kr) {
        a = hvs_lower_bound64(keys_ct, n, ((uint64_t)kv << 32) | kl);
        b = hvs_lower_bound64(keys_ct, n, ((uint64_t)kv << 32) | ((uint64_t)kr + 1ull));
    } else if (p.type == 2u && lr_ok && kl <= kr) {
        a = hvs_lower_bound64(keys_t, n, (uint64_t)kl << 32);
        b = hvs_lower_bound64(keys_t, n, ((uint64_t)kr + 1ull) << 32);
    }
    if (b < a) b = a;
}

// sort key of a query inside a batch: rank:3 | bin of the range start:12 | range end:32.
// Queries that share a wave (128 consecutive slots) then have nearly the same position range, so
// the union range the wave has to stream is close to each query's own range.

// population of each predicate class in the batch (sizes the start-position bins)
// (`list`: the batch is the nq query indices stored there instead of the range [q0, q0 + nq) -- retry batches)
__global__ void hvs_k_count_classes(const float* __restrict__ Q, uint32_t q0, uint32_t nq, const uint32_t* __restrict__ list,
                                    uint32_t* __restrict__ counts)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t rank = 0xFFu;
    if (i < nq) rank = hvs_type_rank(hvs_parse_query(Q + (size_t)(list ? list[i] : q0 + i) * HVS_QCOLS).type);
    // one atomic per wave and class (one per query serialises 262144 atomics on 5 addresses: 1.4 ms)
#pragma unroll
    for (uint32_t k = 0; k < 5u; ++k) {
        const uint32_t c = (uint32_t)__popcll(__ballot(rank == k));
        if (c && (threadIdx.x & 63u) == 0u) atomicAdd(&counts[k], c);
    }
}

#ifndef HVS_HIT_TREE
#define HVS_HIT_TREE 1         // sub-block dispatch of a tile with hits through a tree of scalar ORs (0: linear; A/B: +1.2 % queries/s with the tree)
#endif
#ifndef HVS_FILTER_SETPRIO
#define HVS_FILTER_SETPRIO 0  // hvs_k_filter_i8x16: s_setprio around the matrix block (A/B builds: 1 high while multiplying, 2 high in the
                              // epilogue; measured again in round 3, with few survivors left: -0.4 % either way)
#endif
#ifndef HVS_H16_PREFETCH
#define HVS_H16_PREFETCH 0    // hvs_k_filter_mfma: 1 = the next tile's LDS reads issue under this tile's epilogue, as in hvs_k_filter_i8x16
                              // (measured on FP16 / BF16 tiles: 3 % SLOWER than reading them in front of the tile's own matrix block)
#endif
#ifndef HVS_ORDER_MORTON
#define HVS_ORDER_MORTON 0   // 1: Z-order of (range start, range end) instead of start bins sorted by end (A/B builds; measured in
                             // round 3: type-2 batches 597 vs 592 ms, mixed 675 vs 673 ms, 5 x 10^5-query batches equal: not adopted)
#endif
#ifndef HVS_BIN_QUERIES
#define HVS_BIN_QUERIES 0u   // 0: sqrt rule below; otherwise a fixed number of queries per start-position bin (A/B builds)
#endif
// `counts`: the batch's class populations (hvs_k_count_classes) -- read on the device, so that forming a batch needs
// no host round trip
__global__ void hvs_k_query_keys2(const float* __restrict__ Q, uint32_t q0, uint32_t nq, const uint32_t* __restrict__ list,
                                  const uint64_t* __restrict__ keys_ct, const uint64_t* __restrict__ keys_t, uint32_t n,
                                  const uint32_t* __restrict__ counts, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx,
                                  uint32_t* __restrict__ qa, uint32_t* __restrict__ qb)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const uint32_t qi = list ? list[i] : q0 + i;
    const HvsQParams p = hvs_parse_query(Q + (size_t)qi * HVS_QCOLS);
    const uint32_t rk = hvs_type_rank(p.type);
    uint32_t a, b;
    hvs_query_range(p, keys_ct, keys_t, n, a, b);
    // the range travels with the batch-local index (the sort's payload): hvs_k_layout hands it to the query's slot, so the
    // two binary searches are done once per query and batch
    qa[i] = a;
    qb[i] = b;
#if HVS_BIN_QUERIES
    uint32_t nbins = counts[rk] / HVS_BIN_QUERIES;
#else
    // as many start-position bins as there are quads of 512 queries per bin: a quad's spread of range STARTS (one bin) and of
    // range ENDS (its share of the bin, which is sorted by end) are then about the same fraction of the data
    uint32_t nbins = (uint32_t)__builtin_sqrtf((float)counts[rk] * (1.0f / (float)(HVS_WG_WAVES * HVS_GROUP)));
#endif
    nbins = nbins < 1u ? 1u : (nbins > 4096u ? 4096u : nbins);
    const uint32_t abin = (uint32_t)(((uint64_t)a * nbins) / ((uint64_t)n + 1ull));
#if HVS_ORDER_MORTON
    // A/B: queries of a class along the Z-order curve of (range start, range end), 16 bits each: 128 consecutive queries
    // then fill a compact cell of the (start, end) plane instead of a strip of one start bin
    {
        auto spread = [](uint32_t x) -> uint32_t {  // 16 bits -> every other bit of 32
            x &= 0xFFFFu;
            x = (x | (x << 8)) & 0x00FF00FFu;
            x = (x | (x << 4)) & 0x0F0F0F0Fu;
            x = (x | (x << 2)) & 0x33333333u;
            x = (x | (x << 1)) & 0x55555555u;
            return x;
        };
        const uint32_t a16 = (uint32_t)(((uint64_t)a << 16) / ((uint64_t)n + 1ull)), b16 = (uint32_t)(((uint64_t)b << 16) / ((uint64_t)n + 1ull));
        (void)abin;
        keys[i] = ((uint64_t)rk << 61) | ((uint64_t)((spread(a16) << 1) | spread(b16)) << 16) | (uint64_t)(b & 0xFFFFu);
        idx[i] = i;
        return;
    }
#endif
    keys[i] = ((uint64_t)rk << 61) | ((uint64_t)abin << 32) | (uint64_t)b;
    idx[i] = i;
}

// Slot layout: each class padded to whole 32-slot blocks, the (C,T) part padded to a whole quad of
// groups (a filter workgroup serves one quad and one ordering), the type-2 part to a whole group.  Writes the slot -> query map.
// layout[0..4] = first slot of each class rank, layout[5] = nslots used, layout[6] = first group of T part
// `sorted_idx` holds batch-local indices i: the query is list[i] (retry batches) or q0 + i, its position range qa[i], qb[i].
// Any number of 1024-thread blocks (round 4: ONE block walked all 2^21 slots of a full batch -- 4 ms alone, 14 ms beside the
// previous batch's re-scoring on the other lane): every block derives the class boundaries itself (six binary searches) and
// takes its own 1024 slots; block 0 writes `layout`.
__global__ __launch_bounds__(1024) void hvs_k_layout(const uint64_t* __restrict__ sorted_keys, const uint32_t* __restrict__ sorted_idx,
                                                     uint32_t nq, uint32_t nslots_cap, uint32_t q0, const uint32_t* __restrict__ list,
                                                     const uint32_t* __restrict__ qa, const uint32_t* __restrict__ qb,
                                                     uint32_t* __restrict__ qid, uint32_t* __restrict__ rank, uint32_t* __restrict__ ra,
                                                     uint32_t* __restrict__ rb, uint32_t* __restrict__ layout)
{
    __shared__ uint32_t first[6];   // first sorted index of each rank (first[5] = nq)
    __shared__ uint32_t slot0[6];
    if (threadIdx.x < 6u) {
        const uint32_t rk = threadIdx.x;
        uint32_t lo = 0, hi = nq;  // first index with (key >> 61) >= rk
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint32_t)(sorted_keys[mid] >> 61) < rk) lo = mid + 1; else hi = mid;
        }
        first[rk] = rk == 5u ? nq : lo;
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t s = 0;
        for (uint32_t rk = 0; rk < 5u; ++rk) {
            if (rk == 4u) s = hvs_ceil_div(s, HVS_WG_WAVES * HVS_GROUP) * (HVS_WG_WAVES * HVS_GROUP);  // T-ordering part starts a new filter workgroup
            slot0[rk] = s;
            const uint32_t cnt = first[rk + 1] - first[rk];
            s += hvs_ceil_div(cnt, 32u) * 32u;
        }
        slot0[5] = hvs_ceil_div(s, HVS_GROUP) * HVS_GROUP;
        if (blockIdx.x == 0u) {
            for (uint32_t rk = 0; rk < 5u; ++rk) layout[rk] = slot0[rk];
            layout[5] = slot0[5];
            layout[6] = slot0[4] / HVS_GROUP;
        }
    }
    __syncthreads();
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nslots_cap; s += gridDim.x * blockDim.x) {
        uint32_t rk = 0;
        while (rk < 4u && s >= slot0[rk + 1]) ++rk;
        const uint32_t off = s - slot0[rk];
        const uint32_t cnt = first[rk + 1] - first[rk];
        const bool used = s < slot0[5] && off < cnt;
        const uint32_t li = used ? sorted_idx[first[rk] + off] : 0u;
        qid[s] = used ? (list ? list[li] : q0 + li) : 0xFFFFFFFFu;
        ra[s] = used ? qa[li] : 0u;
        rb[s] = used ? qb[li] : 0u;
        rank[s] = rk;
    }
}

// hvs_k_prep -- one 512-thread block per group of 128 slots (round 4: the per-slot and the per-group preparation were two
// kernels, the first with ONE lane per slot walking the query's 100 dimensions and repeating the binary searches of
// hvs_k_query_keys2: 45 + 25 us of a 10^4-query batch whose whole step is 1.9 ms).
//   per slot (4 lanes, 25 dimensions each, f64 partial sums joined by two shuffles): norms and bound inputs of the query,
//   top-k state reset; the position range comes from hvs_k_layout;
//   per group: union of the slots' ranges, ordering, cleared entry counters;
//   per group: the B-operand fragments of its 128 queries in the tile format `fmt`.
//
// INT8 formats, queries outside the data's bounding box: a coordinate beyond +-127 sd is clipped to +-127, and the part
// that was cut off, c_k = |q'_k| - 127 sd, multiplies a row coordinate |d'_k| <= H_k (the box's half width in that
// dimension), so  |(q' - sd qq).d'| <= e_q N_D + sum_k c_k H_k  with e_q now the rounding error of the UNCLIPPED
// dimensions only.  The sum is kept per slot (`normq`, which the INT8 band does not use otherwise) and hvs_k_merge adds it
// to the band: a query a little outside the box costs a wider band, not the exact engine.  Far outside (the clip term
// above 4x the rest of the band) the filter would let most rows through: the exact engine answers.
__global__ __launch_bounds__(4 * HVS_GROUP) void hvs_k_prep(const float* __restrict__ Q, HvsBatch B, int count_pairs,
                                                            unsigned long long* __restrict__ counters, int fmt,
                                                            const HvsQuant* __restrict__ qz, const HvsBounds* __restrict__ bounds,
                                                            int for_filter)
{
    __shared__ uint32_t smin[HVS_GROUP], smax[HVS_GROUP];
    __shared__ unsigned long long spairs;
    // rotated INT8 tiles (HvsQuant::rot): the int8 image of the group's rotated queries and the per-slot sums over 128 dimensions
    __shared__ __attribute__((aligned(16))) signed char sq8[HVS_GROUP][HVS_RDIM];
    __shared__ double srsum[HVS_GROUP][4];  // |q'|^2, |sd qq|^2, |q' - sd qq|^2 (unclipped dimensions), clip term
    __shared__ int srbad[HVS_GROUP];
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    const uint32_t sl = t >> 2, part = t & 3u;  // slot within the group; quarter of the dimensions
    const uint32_t s = g * HVS_GROUP + sl;
    const uint32_t qi = B.qid[s];
    uint32_t a = B.ra[s], b = B.rb[s];          // the predicate's position range (hvs_k_query_keys2 -> hvs_k_layout)
    if (t == 0u) spairs = 0ull;
    double qn = 0.0, e2 = 0.0, nb2 = 0.0, clipband = 0.0;
    int bad = 0;
    constexpr int kPart = HVS_NDIM / 4;
    const bool rot = fmt == HVS_FMT_I8X16 && qz->rot != 0u;  // uniform over the grid
    if (rot) {
        // every rotated component of every query of the group once: thread -> (slot e >> 7, dimension e & 127)
        if (t < HVS_GROUP) {
            srsum[t][0] = srsum[t][1] = srsum[t][2] = srsum[t][3] = 0.0;
            srbad[t] = 0;
        }
        __syncthreads();
        const double sd = qz->sd, inv_sd = qz->inv_sd;
        for (uint32_t e = t; e < HVS_GROUP * HVS_RDIM; e += blockDim.x) {
            const uint32_t rs = e >> 7, k = e & 127u;
            const uint32_t qj = B.qid[g * HVS_GROUP + rs];
            int v = 0;
            if (qj != 0xFFFFFFFFu) {
                const float* __restrict__ q = Q + (size_t)qj * HVS_QCOLS + 4;
                const double x = hvs_rot_elem(q, k) - (double)qz->center[k];
                v = hvs_quant_i8(x, inv_sd);
                const double xq = sd * (double)v;
                atomicAdd(&srsum[rs][0], x * x);
                atomicAdd(&srsum[rs][1], xq * xq);
                if (fabs(x * inv_sd) <= 127.5) {
                    atomicAdd(&srsum[rs][2], (x - xq) * (x - xq));
                } else if (x == x) {
                    const double c = (double)qz->center[k];
                    const double lo = fabs((double)hvs_attr_key_inv(qz->kmin[k]) - c), hi = fabs((double)hvs_attr_key_inv(qz->kmax[k]) - c);
                    atomicAdd(&srsum[rs][3], fabs(x - xq) * (lo > hi ? lo : hi) * (1.0 + 1e-9));
                } else {
                    srbad[rs] = 1;  // NaN component
                }
            }
            sq8[rs][k] = (signed char)v;
        }
        __syncthreads();
        if (part == 0u) {  // (the shuffles below add three zeros to these)
            qn = srsum[sl][0];
            nb2 = srsum[sl][1];
            e2 = srsum[sl][2];
            clipband = srsum[sl][3];
            bad = srbad[sl];
        }
    } else if (qi != 0xFFFFFFFFu) {
        const float* __restrict__ q = Q + (size_t)qi * HVS_QCOLS + 4;
        if (HVS_IS_I8(fmt)) {
            // relative to the centre: qn = |q'|^2, nb2 = |sd qq|^2, e2 = |q' - sd qq|^2 over the unclipped dimensions
            const double sd = qz->sd, inv_sd = qz->inv_sd;
            for (int k = (int)part * kPart; k < (int)(part + 1u) * kPart; ++k) {
                const double x = (double)q[k] - (double)qz->center[k];
                const double xq = sd * (double)hvs_quant_i8(x, inv_sd);
                qn += x * x;
                nb2 += xq * xq;
                if (fabs(x * inv_sd) <= 127.5) {
                    e2 += (x - xq) * (x - xq);
                } else if (x == x) {
                    const double c = (double)qz->center[k];
                    const double lo = fabs((double)hvs_attr_key_inv(qz->kmin[k]) - c), hi = fabs((double)hvs_attr_key_inv(qz->kmax[k]) - c);
                    clipband += fabs(x - xq) * (lo > hi ? lo : hi) * (1.0 + 1e-9);
                } else {
                    bad = 1;  // NaN component
                }
            }
        } else {
            for (int k = (int)part * kPart; k < (int)(part + 1u) * kPart; ++k) {
                const float x = q[k];
                const float xb = hvs_h16_to_f32(fmt, hvs_h16_bits(fmt, x));
                qn += (double)x * (double)x;
                e2 += ((double)x - (double)xb) * ((double)x - (double)xb);
                nb2 += (double)xb * (double)xb;
            }
        }
    }
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {  // the slot's 4 lanes are neighbours in the wave
        qn += __shfl_xor(qn, o);
        e2 += __shfl_xor(e2, o);
        nb2 += __shfl_xor(nb2, o);
        clipband += __shfl_xor(clipband, o);
        bad |= __shfl_xor(bad, o);
    }
    __syncthreads();  // (spairs is cleared)
    if (part == 0u) {
        bool hopeless = bad != 0;
        if (qi != 0xFFFFFFFFu) {
            if (HVS_IS_I8(fmt)) {
                const double rest = sqrt(nb2) * (double)bounds->e_d8 + sqrt(e2) * (double)bounds->n_d8;
                if (!(clipband <= 4.0 * rest)) hopeless = true;
            } else if (fmt == HVS_FMT_F16) {
                e2 = (sqrt(e2) + HVS_F16_FLUSH) * (sqrt(e2) + HVS_F16_FLUSH);  // (possible denormal flush, see HVS_F16_FLUSH)
                if (!(nb2 < 1.0e9)) hopeless = true;                           // a component beyond the half-precision range
            }
            if (count_pairs && b > a) atomicAdd(&spairs, (unsigned long long)(b - a));
        }
        // a query without a usable bound (non-finite components, far outside the data's box) is answered by the exact engine
        // and takes no part in the filter (empty range)
        if (!(qn < 1.0e30)) hopeless = true;
        if (qi == 0xFFFFFFFFu || !for_filter) hopeless = false;  // (the exact engine's range scans use the ranges only)
        if (hopeless) a = b = 0u;
        B.ra[s] = a;
        B.rb[s] = b;
        B.qn[s] = qn;
        B.normq[s] = HVS_IS_I8(fmt) ? (clipband > 0.0 ? hvs_round_up_f32(clipband) : 0.0f) : hvs_round_up_f32(sqrt(qn) * (1.0 + 1e-9) + 1e-30);
        B.eq[s] = hvs_round_up_f32(sqrt(e2) * (1.0 + 1e-9) + 1e-30);
        B.nqb[s] = hvs_round_up_f32(sqrt(nb2) * (1.0 + 1e-9) + 1e-30);
        B.topcnt[s] = 0;
        B.candcnt[s] = 0;
        B.overflow[s] = hopeless ? HVS_FAIL_EXACT : 0u;
        B.tau[s] = __builtin_inff();
        // -inf: everything in range is a candidate until a threshold is set; +inf: nothing can ever match
        if (HVS_IS_I8(fmt))
            B.thetai[s] = b > a ? (int)0x80000000 : 0x7FFFFFFF;
        else
            B.theta[s] = b > a ? -__builtin_inff() : __builtin_inff();
        smin[sl] = a < b ? a : 0xFFFFFFFFu;
        smax[sl] = a < b ? b : 0u;
    }
    __syncthreads();
    for (uint32_t w = HVS_GROUP / 2; w > 0; w >>= 1) {
        if (t < w) {
            smin[t] = smin[t] < smin[t + w] ? smin[t] : smin[t + w];
            smax[t] = smax[t] > smax[t + w] ? smax[t] : smax[t + w];
        }
        __syncthreads();
    }
    if (t == 0u) {
        const bool any = smin[0] < smax[0];
        B.gua[g] = any ? smin[0] : 0u;
        B.gub[g] = any ? smax[0] : 0u;
        B.gord[g] = B.rank[g * HVS_GROUP] == 4u ? 1u : 0u;
        B.paircnt[g] = 0;
        B.goverflow[g] = 0;
        if (count_pairs && spairs) atomicAdd(&counters[0], spairs);
    }
    if (rot) {
        // B fragments cut from the int8 image of the rotated queries (all 128 K slots carry data)
        constexpr uint32_t kSub = HVS_GROUP / HVS_I8X16_QSUB;
        for (uint32_t e = t; e < kSub * 2u * 64u; e += blockDim.x) {
            const uint32_t j = e / 128u, ks = (e / 64u) & 1u, l = e & 63u;
            B.bfrag[((size_t)(g * kSub + j) * 2u + ks) * 64u + l] =
                *reinterpret_cast<const uint4*>(&sq8[j * HVS_I8X16_QSUB + (l & 15u)][64u * ks + 16u * (l >> 4)]);
        }
        return;
    }
    if (fmt == HVS_FMT_I8X16) {
        // B fragments of v_mfma_i32_16x16x64_i8: fragment (sub-block j of 16 queries, k-step ks): lane l holds query
        // column 16 j + (l & 15), k = 64 ks + 16 (l >> 4) + 0..15; stored [group][j][ks][lane]
        const double inv_sd = qz->inv_sd;
        constexpr uint32_t kSub = HVS_GROUP / HVS_I8X16_QSUB;
        for (uint32_t e = t; e < kSub * 2u * 64u; e += blockDim.x) {
            const uint32_t j = e / 128u, ks = (e / 64u) & 1u, l = e & 63u;
            const uint32_t slot = g * HVS_GROUP + j * HVS_I8X16_QSUB + (l & 15u);
            const uint32_t qj = B.qid[slot];
            uint32_t w[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint32_t word = 0;
#pragma unroll
                for (int e2i = 0; e2i < 4; ++e2i) {
                    const int k = 64 * (int)ks + 16 * (int)(l >> 4) + 4 * p + e2i;
                    int v = 0;
                    if (qj != 0xFFFFFFFFu && k < HVS_NDIM)
                        v = hvs_quant_i8((double)Q[(size_t)qj * HVS_QCOLS + 4 + k] - (double)qz->center[k], inv_sd);
                    word |= ((uint32_t)v & 0xFFu) << (8 * e2i);
                }
                w[p] = word;
            }
            B.bfrag[((size_t)(g * kSub + j) * 2u + ks) * 64u + l] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        return;
    }
    if (fmt == HVS_FMT_I8) {
        // INT8 B fragments: lane l of k-step ks holds query column (l & 31), k = 32 ks + 16 (l >> 5) + 0..15
        const double inv_sd = qz->inv_sd;
        for (uint32_t e = t; e < HVS_QB * HVS_I8_KSTEPS * 64u; e += blockDim.x) {
            const uint32_t qb = e / (HVS_I8_KSTEPS * 64u);
            const uint32_t ks = (e / 64u) % HVS_I8_KSTEPS;
            const uint32_t l = e & 63u;
            const uint32_t slot = g * HVS_GROUP + qb * 32u + (l & 31u);
            const uint32_t qj = B.qid[slot];
            uint32_t w[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint32_t word = 0;
#pragma unroll
                for (int e2i = 0; e2i < 4; ++e2i) {
                    const int k = 32 * (int)ks + 16 * (int)(l >> 5) + 4 * p + e2i;
                    int v = 0;
                    if (qj != 0xFFFFFFFFu && k < HVS_NDIM)
                        v = hvs_quant_i8((double)Q[(size_t)qj * HVS_QCOLS + 4 + k] - (double)qz->center[k], inv_sd);
                    word |= ((uint32_t)v & 0xFFu) << (8 * e2i);
                }
                w[p] = word;
            }
            B.bfrag[((size_t)(g * HVS_QB + qb) * HVS_I8_KSTEPS + ks) * 64u + l] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        return;
    }
    // B fragments: lane l of k-step ks holds query column (l & 31), k = 16 ks + 8 (l >> 5) + 0..7
    for (uint32_t e = t; e < HVS_QB * HVS_KSTEPS * 64u; e += blockDim.x) {
        const uint32_t qb = e / (HVS_KSTEPS * 64u);
        const uint32_t ks = (e / 64u) % HVS_KSTEPS;
        const uint32_t l = e & 63u;
        const uint32_t slot = g * HVS_GROUP + qb * 32u + (l & 31u);
        const uint32_t qj = B.qid[slot];
        uint32_t w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint16_t v2[2];
#pragma unroll
            for (int e2i = 0; e2i < 2; ++e2i) {
                const int k = 16 * (int)ks + 8 * (int)(l >> 5) + 2 * p + e2i;
                uint16_t v = 0;
                if (qj != 0xFFFFFFFFu) {
                    if (k < HVS_NDIM)
                        v = hvs_h16_bits(fmt, Q[(size_t)qj * HVS_QCOLS + 4 + k]);
                    else if (k < 103)
                        v = fmt == HVS_FMT_F16 ? 0x3C00 : 0x3F80;  // 1.0: multiplies the three -|d|^2/2 pieces
                }
                v2[e2i] = v;
            }
            w[p] = (uint32_t)v2[0] | ((uint32_t)v2[1] << 16);
        }
        B.bfrag[((size_t)(g * HVS_QB + qb) * HVS_KSTEPS + ks) * 64u + l] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// hvs_k_seed_exact -- level 0 by the exact kernel (lane = query, wave-uniform rows through the
// scalar cache, exactly like hvs_k_scan_exact) over the level-0 blocks of the group's range.
// The per-lane predicate is the slot's own position range.  Survivors go to cand[slot].
// One wave per 64 slots.
// ---------------------------------------------------------------------------------------------
#ifndef HVS_SEED_LDS
#define HVS_SEED_LDS 1      // level-0 rows through a wave-private LDS image (0: one by one through the scalar cache; A/B builds)
#endif
#ifndef HVS_SEED_STAGE
#define HVS_SEED_STAGE 16u  // rows per image (a power of two dividing 32)
#endif
#ifndef HVS_SEED_UNROLL
#define HVS_SEED_UNROLL 7   // gather loads of a lane in flight at a time (13 = all of a 16-row image: 50 registers spilled)
#endif
#define HVS_PRAGMA_(x) _Pragma(#x)
#define HVS_PRAGMA(x) HVS_PRAGMA_(x)
struct HvsUniformRowF2 {
    const hvs_f2* __restrict__ p;
    __device__ __forceinline__ hvs_f2 operator[](int i) const { return p[i]; }
};

template <int CAP>
__global__ __launch_bounds__(256, 3) void hvs_k_seed_exact(const float* __restrict__ D, uint32_t n, uint32_t sn,
                                                           const float* __restrict__ Q, HvsBatch B,
                                                           const uint32_t* __restrict__ perm_ct,
                                                           const uint32_t* __restrict__ perm_t,
                                                           const uint32_t* __restrict__ bpos_ct,
                                                           const uint32_t* __restrict__ bpos_t, HvsLevels L,
                                                           unsigned long long* __restrict__ counters, uint32_t nchunks)
{
#if HVS_SEED_LDS
    __shared__ float4 srow[4][HVS_SEED_STAGE][26];  // per wave: 16 data rows as 16-byte aligned images x0..x99 (+ pad)
#endif
    // nchunks > 1 (small batches, level 0 of at most 1024 rows): grid.y waves share one 64-slot group, each
    // takes every nchunks-th level-0 block and appends to the slots' lists with atomics -- a single wave per
    // group walks ~1000 randomly placed rows one after the other and is latency-bound (2 ms at 10^4 queries)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t slot = w * 64u + lane;
    if (w * 64u >= B.nslots) return;
    const uint32_t g = (w * 64u) / HVS_GROUP;
    const uint32_t ord = B.gord[g];
    const uint32_t* __restrict__ perm = ord ? perm_t : perm_ct;
    const uint32_t* __restrict__ bpos = ord ? bpos_t : bpos_ct;
    const uint32_t ua = B.gua[g], ub = B.gub[g];
    uint32_t lo, hi;
    hvs_level_run(L, 0, ua / 32u, hvs_ceil_div(ub, 32u), lo, hi);
    if (lo >= hi) return;

    const uint32_t qi = B.qid[slot];
    const bool have_q = qi != 0xFFFFFFFFu;
    const float* __restrict__ qrow = Q + (size_t)(have_q ? qi : 0u) * HVS_QCOLS;
    const uint32_t ra = have_q ? B.ra[slot] : 0u, rb = have_q ? B.rb[slot] : 0u;
    hvs_f2 q2[HVS_NDIM / 2];
#pragma unroll
    for (int i = 0; i < HVS_NDIM / 4; ++i) {
        const float4 v4 = *reinterpret_cast<const float4*>(qrow + 4 + 4 * i);
        q2[2 * i] = hvs_f2{v4.x, v4.y};
        q2[2 * i + 1] = hvs_f2{v4.z, v4.w};
    }
    uint64_t* __restrict__ mylist = B.cand + (size_t)slot * B.fcap;
    float tau = __builtin_inff();
    uint32_t cnt = 0, nscan = 0;
    for (uint32_t i = lo + blockIdx.y; i < hi; i += nchunks) {
        const uint32_t b = bpos[i];
        uint32_t kk = 0, kend = 0;  // chunked form: this lane's places [kk, kend) in its slot's list for the block's rows
        if (nchunks > 1u) {         // wave-uniform
            // ONE atomic per lane and block instead of one per row and lane: the rows of the block a lane takes are its
            // range cut to the block, less those outside the sampled prefix
            const uint32_t p0 = b * 32u, p1 = (p0 + 32u < n) ? p0 + 32u : n;
            const uint32_t a = ra > p0 ? ra : p0, e = rb < p1 ? rb : p1;
            uint32_t mine = a < e ? e - a : 0u;
            if (sn < n) {
                for (uint32_t pos = p0; pos < p1; ++pos)
                    if (perm[pos] >= sn && pos >= ra && pos < rb) --mine;  // perm[pos]: wave-uniform (scalar) load
            }
            if (mine != 0u) {
                kk = atomicAdd(&B.candcnt[slot], mine);
                kend = kk + mine;
                if (kend > B.fcap) {
                    hvs_flag_fail(B.fail_code, B.overflow, slot);
                    kend = kk < B.fcap ? B.fcap : kk;
                }
            }
        }
        for (uint32_t r = 0; r < 32u; ++r) {
            const uint32_t pos = b * 32u + r;
            if (pos >= n) break;
#if HVS_SEED_LDS
            // The block's rows sit anywhere in D (perm): fetched one by one through the scalar cache, every row is a dependent
            // round trip to HBM (0.23 ms for 960 rows x 10^4 queries, a quarter of the exact engine's rate).  16 rows at a
            // time are gathered by the whole wave instead (800 8-byte loads in flight) into a wave-private LDS image and
            // read back as broadcast ds_read_b128 by the exact engine's row loop (hvs_exact_dist_pk_lds).
            if ((r & (HVS_SEED_STAGE - 1u)) == 0u) {
                const uint32_t p0 = pos, p1 = p0 + HVS_SEED_STAGE;
                if (__ballot(ra < p1 && rb > p0) == 0ull) {  // no lane's range meets these rows
                    r += HVS_SEED_STAGE - 1u;
                    continue;
                }
                hvs_f2* img = reinterpret_cast<hvs_f2*>(&srow[threadIdx.x >> 6][0][0]);
HVS_PRAGMA(unroll HVS_SEED_UNROLL)
                for (uint32_t t = 0; t < (HVS_SEED_STAGE * 50u + 63u) / 64u; ++t) {
                    const uint32_t e = lane + 64u * t, rr = e / 50u, c = e % 50u;
                    if (e < HVS_SEED_STAGE * 50u && p0 + rr < n) {
                        const uint32_t idr = perm[p0 + rr];
                        img[rr * 52u + c] = *reinterpret_cast<const hvs_f2*>(D + (size_t)idr * HVS_DCOLS + 2u + 2u * c);
                    }
                }
                // the image is written and read by different lanes of ONE wave: LDS operations of a wave execute in order,
                // the fence keeps the compiler from moving the row loop's reads in front of the writes
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#endif
            const bool pass = pos >= ra && pos < rb;
            if (__ballot(pass) == 0ull) continue;
            const uint32_t id = perm[pos];
            if (id >= sn) continue;  // sampled prefix: rows [0, sn) of the original order only
            nscan += 64u;
#if HVS_SEED_LDS
            const float dist = hvs_exact_dist_pk_lds(&srow[threadIdx.x >> 6][r & (HVS_SEED_STAGE - 1u)][0], q2);
#else
            HvsUniformRowF2 dv{reinterpret_cast<const hvs_f2*>(D + (size_t)id * HVS_DCOLS + 2)};
            const float dist = hvs_exact_dist_pk(dv, q2);
#endif
            if (nchunks > 1u) {  // wave-uniform
                if (pass && kk < kend) mylist[kk] = hvs_make_key(dist, id);
                kk += pass ? 1u : 0u;
                continue;
            }
            if (pass && dist <= tau) {
                mylist[cnt] = hvs_make_key(dist, id);
                ++cnt;
            }
            uint64_t full = __ballot(cnt == (uint32_t)CAP);
            if (full != 0ull) {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                while (full != 0ull) {
                    const uint32_t l = (uint32_t)__builtin_ctzll(full);
                    full &= full - 1ull;
                    uint64_t* lst = B.cand + (size_t)(w * 64u + l) * B.fcap;
                    const uint64_t kth = hvs_wave_select_prune<CAP / 64>(lst, (uint32_t)CAP, B.knn, lane);
                    if (lane == l) {
                        cnt = B.knn;
                        tau = hvs_key_dist(kth);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
        }
    }
    if (nchunks == 1u) B.candcnt[slot] = cnt;
    if (lane == 0u) atomicAdd(&counters[1], (unsigned long long)nscan);
}

// ---------------------------------------------------------------------------------------------
// hvs_k_scan_ranges -- the exact engine on top of the index: the same lane-per-query, scalar-row
// kernel as hvs_k_scan_exact, but a wave walks only the POSITION RANGE of its 64 queries (the
// union of their predicate ranges in the (C,T) or T ordering) instead of all of D, and the per-lane
// predicate is a position-range test.  Type-1/3 queries then touch ~1 % / 0.25 % of the rows and a
// type-2 wave touches only rows that some of its queries want.  grid.y cuts the range into chunks.
// With sn < n (sample_proportion < 1) rows whose original id is >= sn are skipped.
// ---------------------------------------------------------------------------------------------
template <bool SCALAR_ORDER, int CAP>
__global__ __launch_bounds__(256, 3) void hvs_k_scan_ranges(const float* __restrict__ D, uint32_t sn,
                                                            const float* __restrict__ Q, HvsBatch B,
                                                            const uint32_t* __restrict__ perm_ct,
                                                            const uint32_t* __restrict__ perm_t, uint32_t nchunks,
                                                            uint32_t slot_begin, uint32_t slot_end, uint64_t* __restrict__ cand,
                                                            uint32_t* __restrict__ cand_cnt,
                                                            unsigned long long* __restrict__ counters)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t slot = w * 64u + lane;
    if (w * 64u >= slot_end || w * 64u + 64u <= slot_begin) return;
    const uint32_t chunk = blockIdx.y;
    const uint32_t* __restrict__ perm = perm_ct;  // only (C,T)-ordering classes (types 1, 3) are routed here
    (void)perm_t;

    // slots outside [slot_begin, slot_end) are type-0 / type-2 queries: answered by the sequential full scan
    const uint32_t qi = (slot >= slot_begin && slot < slot_end) ? B.qid[slot] : 0xFFFFFFFFu;
    const bool have_q = qi != 0xFFFFFFFFu;
    const float* __restrict__ qrow = Q + (size_t)(have_q ? qi : 0u) * HVS_QCOLS;
    const uint32_t ra = have_q ? B.ra[slot] : 0u, rb = have_q ? B.rb[slot] : 0u;
    // union of this wave's ranges (64 slots; the group's union would also cover the other half)
    uint32_t ua = ra < rb ? ra : 0xFFFFFFFFu, ub = ra < rb ? rb : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t oa = __shfl_xor(ua, o), ob = __shfl_xor(ub, o);
        ua = oa < ua ? oa : ua;
        ub = ob > ub ? ob : ub;
    }
    ua = __builtin_amdgcn_readfirstlane(ua);
    ub = __builtin_amdgcn_readfirstlane(ub);
    if (ua >= ub) return;
    const uint32_t len = hvs_ceil_div(ub - ua, nchunks);
    const uint32_t p0 = ua + chunk * len;
    const uint32_t p1 = (p0 + len) < ub ? (p0 + len) : ub;
    if (p0 >= p1) return;

    hvs_f2 q2[HVS_NDIM / 2];
#pragma unroll
    for (int i = 0; i < HVS_NDIM / 4; ++i) {
        const float4 v4 = *reinterpret_cast<const float4*>(qrow + 4 + 4 * i);
        q2[2 * i] = hvs_f2{v4.x, v4.y};
        q2[2 * i + 1] = hvs_f2{v4.z, v4.w};
    }
    uint64_t* __restrict__ mylist = cand + ((size_t)chunk * B.nslots + slot) * (uint32_t)CAP;
    float tau = __builtin_nanf("");  // admits every passing row until the first cut (see hvs_k_scan_exact)
    uint32_t cnt = 0, nscan = 0;
    for (uint32_t pos = p0; pos < p1; ++pos) {
        const bool pass = pos >= ra && pos < rb;
        if (__ballot(pass) == 0ull) continue;
        const uint32_t id = perm[pos];
        if (id >= sn) continue;  // sampled prefix: rows [0, sn) of the original order only
        nscan += 64u;
        const float* __restrict__ row = D + (size_t)id * HVS_DCOLS + 2;
        float dist;
        if (SCALAR_ORDER) {
            struct R1 { const float* __restrict__ p; __device__ __forceinline__ float operator[](int i) const { return p[i]; } } d1{row};
            struct Q1 { const hvs_f2* q; __device__ __forceinline__ float operator[](int i) const { return (i & 1) ? q[i >> 1].y : q[i >> 1].x; } } q1{q2};
            dist = hvs_scalar_order_dist(d1, q1);
        } else {
            HvsUniformRowF2 dv{reinterpret_cast<const hvs_f2*>(row)};
            dist = hvs_exact_dist_pk(dv, q2);
        }
        if (pass && !(dist > tau)) {  // ids are not monotone along a position range: keep ties, the keys sort them out
            mylist[cnt] = hvs_make_key(dist, id);
            ++cnt;
        }
        uint64_t full = __ballot(cnt == (uint32_t)CAP);
        if (full != 0ull) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            while (full != 0ull) {
                const uint32_t l = (uint32_t)__builtin_ctzll(full);
                full &= full - 1ull;
                uint64_t* lst = cand + ((size_t)chunk * B.nslots + (w * 64u + l)) * (uint32_t)CAP;
                const uint64_t kth = hvs_wave_select_prune<CAP / 64>(lst, (uint32_t)CAP, B.knn, lane);
                if (lane == l) {
                    cnt = B.knn;
                    tau = hvs_key_dist(kth);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
    }
    if (have_q) cand_cnt[(size_t)chunk * B.nslots + slot] = cnt;
    if (lane == 0u) atomicAdd(&counters[1], (unsigned long long)nscan);
}

// passing pairs of a sampled prefix (sn < n): the position range over-counts, so count ids < sn exactly
__global__ void hvs_k_count_prefix_pairs(HvsBatch B, const uint32_t* __restrict__ perm_ct, const uint32_t* __restrict__ perm_t,
                                         uint32_t sn, unsigned long long* __restrict__ counters)
{
    const uint32_t slot = blockIdx.x;
    if (B.qid[slot] == 0xFFFFFFFFu) return;
    const uint32_t* __restrict__ perm = B.gord[slot / HVS_GROUP] ? perm_t : perm_ct;
    uint32_t c = 0;
    for (uint32_t pos = B.ra[slot] + threadIdx.x; pos < B.rb[slot]; pos += blockDim.x) c += perm[pos] < sn ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63u) == 0u && c) atomicAdd(&counters[0], (unsigned long long)c);
}

// ---------------------------------------------------------------------------------------------
// hvs_k_filter_mfma<FMT> -- the dominant kernel (FMT = tile format: BF16 or INT8).
//
// Work item of a wave = (group of 4 query blocks = 128 queries, one level, one segment of <= HVS_SEG
// row blocks of the level's storage).  A 256-thread workgroup = 4 waves = 4 consecutive groups on the
// SAME segment: each tile (BF16 7 KiB, INT8 4 KiB + 128 B of accumulator inits) is brought into LDS once per
// workgroup by LDS-DMA (stages of 4 tiles, double-buffered, one barrier per stage) and read from there by
// all four waves.
//
// Per tile and wave: ds_read_b128 fetch the A fragments (7 / 4); against each of the wave's 4 resident
// query blocks (B fragments in 112 / 64 VGPRs) a chain of 7 v_mfma_f32_32x32x16_bf16 (resp. 4
// v_mfma_i32_32x32x32_i8 started from the rows' integer norm terms) yields s[row][query] = q.d - |d|^2/2 (in the
// format's units); the chains are interleaved k-step by k-step.  The 16 accumulators of a lane belong to ONE
// query (the lane's column), so the test "could this row still enter the query's top-100"  s >= theta[query]
// needs one v_max3 chain and one compare against a per-lane constant; blocks inside every lane's own position
// range (wave-uniform test) skip the per-lane range test.  Survivors (about 0.8 per wave and tile) are found from
// a per-lane bit mask of the accumulators (v_cmp + v_addc_co), range-checked, packed as (slot << 32 | position)
// into a wave-private LDS buffer and flushed to the group's pair list with one returning atomic per ~200 pairs.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ hvs_bf16x8 hvs_as_bf16x8(const uint4& u)
{
    union {
        uint4 u4;
        hvs_bf16x8 b8;
    } c;
    c.u4 = u;
    return c.b8;
}

__device__ __forceinline__ hvs_i32x4 hvs_as_i32x4(const uint4& u)
{
    union {
        uint4 u4;
        hvs_i32x4 i4;
    } c;
    c.u4 = u;
    return c.i4;
}

// Bit r of the result = (a_r >= th), for the 16 accumulators of a lane: v_cmp into a scalar pair + v_addc_co
// (mask = 2 mask + carry) per accumulator, 33 VALU instructions.  (The compiler's own rendering of the same
// expression is compare / select / or / shift plus hazard nops, about twice that.)  gfx950 wants two wait
// states between a VALU writing a scalar pair and a VALU reading it: three pairs rotate, so every v_addc_co
// reads a pair written at least four instructions earlier.  CMP is "v_cmp_ge_i32_e64" or "v_cmp_ge_f32_e64".
#define HVS_HITMASK_ASM(CMP)                                                                                  \
    "v_mov_b32 %0, 0\n\t" CMP " %1, %20, %21\n\t" CMP " %2, %19, %21\n\t" CMP " %3, %18, %21\n\t"                 \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\t" CMP " %1, %17, %21\n\t"                                         \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\t" CMP " %2, %16, %21\n\t"                                         \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\t" CMP " %3, %15, %21\n\t"                                         \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\t" CMP " %1, %14, %21\n\t"                                         \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\t" CMP " %2, %13, %21\n\t"                                         \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\t" CMP " %3, %12, %21\n\t"                                         \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\t" CMP " %1, %11, %21\n\t"                                         \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\t" CMP " %2, %10, %21\n\t"                                         \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\t" CMP " %3, %9, %21\n\t"                                          \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\t" CMP " %1, %8, %21\n\t"                                          \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\t" CMP " %2, %7, %21\n\t"                                          \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\t" CMP " %3, %6, %21\n\t"                                          \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\t" CMP " %1, %5, %21\n\t"                                          \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\t"                                                                 \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\t"                                                                 \
    "s_nop 1\n\t"                                                                                             \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %1"
template <typename ACC, typename THR>
__device__ __forceinline__ uint32_t hvs_hit_mask(const ACC& a, THR th)
{
    uint32_t m;
    uint64_t s0, s1, s2, sc;
    if constexpr (sizeof(THR) == 4 && ((THR)0.5f != (THR)0))  // float thresholds
        asm volatile(HVS_HITMASK_ASM("v_cmp_ge_f32_e64")
                     : "=&v"(m), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(sc)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]),
                       "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]), "v"(th));
    else
        asm volatile(HVS_HITMASK_ASM("v_cmp_ge_i32_e64")
                     : "=&v"(m), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(sc)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]),
                       "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]), "v"(th));
    return m;
}

#define HVS_ITEM_QUAD_BITS 13  // item code: (segment within the level << 13) | quad of groups (2^13 quads = 2^22 query slots)
// Work items.  A launch of one level is a fixed number of workgroups that pull (quad of groups, segment) items from
// the level's list (HvsItems, built per batch by hvs_k_item_*: only pairs whose ranges meet, ordered segment-major so
// that concurrent workgroups stream the same tiles) with one atomic per item.  A 2-D grid over all (quad, segment)
// pairs is mostly empty for windowed predicates -- 75 % of the workgroups of a 25 % timestamp window, > 98 % for
// categorical ones -- and the empties cost ~7 ns each of dispatch time (18 % of a type-2 batch's filter time).
struct HvsItems {
    const uint32_t* list;    // items of all levels: (segment within the level << HVS_ITEM_QUAD_BITS) | quad
    const uint32_t* lvloff;  // [K + 2] first item of each level; lvloff[level + 1] - lvloff[level] = the level's items
    uint32_t* cursor;        // [16] next item of each level (zeroed per batch)
    uint32_t segsize;        // row blocks per item at the level being launched (HvsSegs::seg)
};

// operand / accumulator types and the MFMA step of the two tile formats
template <int FMT>
struct HvsFmt;
template <>
struct HvsFmt<HVS_FMT_BF16> {
    static constexpr int KSTEPS = HVS_KSTEPS;
    static constexpr int KMEM = HVS_KSTEPS;  // k-steps stored in the tile
    typedef hvs_bf16x8 frag_t;
    typedef hvs_f32x16 acc_t;
    typedef float thr_t;
    static __device__ __forceinline__ frag_t frag(const uint4& u) { return hvs_as_bf16x8(u); }
    static __device__ __forceinline__ acc_t mfma(const frag_t& a, const frag_t& b, const acc_t& c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ thr_t max2(thr_t a, thr_t b) { return fmaxf(a, b); }
};
typedef _Float16 hvs_f16x8 __attribute__((ext_vector_type(8)));
template <>
struct HvsFmt<HVS_FMT_F16> {
    static constexpr int KSTEPS = HVS_KSTEPS;
    static constexpr int KMEM = HVS_KSTEPS;
    typedef hvs_f16x8 frag_t;
    typedef hvs_f32x16 acc_t;
    typedef float thr_t;
    static __device__ __forceinline__ frag_t frag(const uint4& u)
    {
        union {
            uint4 u4;
            hvs_f16x8 h8;
        } c;
        c.u4 = u;
        return c.h8;
    }
    static __device__ __forceinline__ acc_t mfma(const frag_t& a, const frag_t& b, const acc_t& c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ thr_t max2(thr_t a, thr_t b) { return fmaxf(a, b); }
};
template <>
struct HvsFmt<HVS_FMT_I8> {
    static constexpr int KSTEPS = HVS_I8_KSTEPS;
    static constexpr int KMEM = HVS_I8_KMEM;
    typedef hvs_i32x4 frag_t;
    typedef hvs_i32x16 acc_t;
    typedef int thr_t;
    static __device__ __forceinline__ frag_t frag(const uint4& u) { return hvs_as_i32x4(u); }
    static __device__ __forceinline__ acc_t mfma(const frag_t& a, const frag_t& b, const acc_t& c)
    {
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ thr_t max2(thr_t a, thr_t b) { return a > b ? a : b; }
};

template <int FMT>
__global__ __launch_bounds__(64 * HVS_WG_WAVES, HVS_FILTER_OCC) void hvs_k_filter_mfma(const uint4* __restrict__ tiles_ct,
                                                            const uint4* __restrict__ tiles_t,
                                                            const uint4* __restrict__ nrm_ct,
                                                            const uint4* __restrict__ nrm_t,
                                                            const uint32_t* __restrict__ bpos_ct,
                                                            const uint32_t* __restrict__ bpos_t, HvsLevels L,
                                                            uint32_t level, HvsBatch B, HvsItems W,
                                                            unsigned long long* __restrict__ counters)
{
    typedef HvsFmt<FMT> F;
    constexpr int KS = F::KSTEPS;
    constexpr int KM = F::KMEM;  // k-steps held by the tile in memory (INT8: 3 of the 4)
    constexpr int TILE_U4 = KM * 64;
    constexpr bool kI8 = FMT == HVS_FMT_I8;
    constexpr int STG = kI8 ? HVS_STAGE_I8 : HVS_STAGE;  // tiles per LDS stage
    // two stages of 4 A tiles shared by the 4 waves (BF16: 2 x 28 KiB, INT8: 2 x 16 KiB + the rows' accumulator inits)
    __shared__ uint4 stile[2][STG * TILE_U4];
    __shared__ uint4 snrm[2][kI8 ? STG * HVS_I8_NRM_U4 : 1];
    __shared__ uint64_t sbuf[HVS_WG_WAVES][256];         // wave-private survivor buffers
    __shared__ uint32_t srange[HVS_WG_WAVES][2];
    __shared__ uint32_t sitem;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = threadIdx.x >> 6;
    // Work items (HvsItems): a fixed crew of workgroups pulls (quad of groups, segment) pairs of this level, ordered
    // segment-major: consecutive items stream the same run of tiles for different queries, so a segment
    // is fetched from HBM about once and then served by the XCDs' L2s / the Infinity Cache.
    // The quad's 4 waves walk the segment together: every tile is fetched once per workgroup
    // into LDS and read from there by all 4 waves (ds_read_b128).  All groups of a quad use the
    // same ordering (the T-ordering part of a batch starts at a quad boundary).
    const uint32_t item0 = W.lvloff[level], nitems = W.lvloff[level + 1u] - item0;
  for (;;) {
    if (threadIdx.x == 0u) sitem = atomicAdd(&W.cursor[level], 1u);
    __syncthreads();
    const uint32_t item = __builtin_amdgcn_readfirstlane(sitem);
    if (item >= nitems) break;  // uniform over the workgroup
    const uint32_t code = __builtin_amdgcn_readfirstlane(W.list[item0 + item]);
    const uint32_t quad = code & ((1u << HVS_ITEM_QUAD_BITS) - 1u), segment = code >> HVS_ITEM_QUAD_BITS;
    const uint32_t g = quad * HVS_WG_WAVES + wv;
    const uint32_t gq = quad * HVS_WG_WAVES;  // first group of the workgroup decides the ordering
    const uint32_t ord = B.gord[gq];
    const uint4* __restrict__ tiles = ord ? tiles_t : tiles_ct;
    const uint32_t* __restrict__ bpos = ord ? bpos_t : bpos_ct;
    const uint4* __restrict__ nrm = ord ? nrm_t : nrm_ct;
    const uint32_t seg_lo = L.off[level] + segment * W.segsize;
    uint32_t i0 = 0, i1 = 0;  // this wave's tiles [i0,i1) inside the segment (empty when i0 >= i1)
    if (g < B.ngroups && B.gord[g] == ord) {
        uint32_t lo, hi;
        hvs_level_run(L, level, B.gua[g] / 32u, hvs_ceil_div(B.gub[g], 32u), lo, hi);
        if (seg_lo < hi && seg_lo + W.segsize > lo) {
            i0 = seg_lo > lo ? seg_lo : lo;
            i1 = (seg_lo + W.segsize) < hi ? (seg_lo + W.segsize) : hi;
        }
    }
    // (computing all four groups' runs from scalars in every wave, without this LDS exchange and barrier,
    // measured 1 % slower)
    if (lane == 0u) {
        srange[wv][0] = i0 < i1 ? i0 : 0xFFFFFFFFu;
        srange[wv][1] = i0 < i1 ? i1 : 0u;
    }
    __syncthreads();
    uint32_t I0 = srange[0][0], I1 = srange[0][1];
#pragma unroll
    for (int w = 1; w < HVS_WG_WAVES; ++w) {
        I0 = srange[w][0] < I0 ? srange[w][0] : I0;
        I1 = srange[w][1] > I1 ? srange[w][1] : I1;
    }
    if (I0 >= I1) continue;  // uniform over the workgroup (cannot happen with a well-formed list)
    I0 = __builtin_amdgcn_readfirstlane(I0);  // (read from LDS: tell the compiler they are scalars)
    I1 = __builtin_amdgcn_readfirstlane(I1);
    i0 = __builtin_amdgcn_readfirstlane(i0);  // wave-uniform by construction: keep the tile test scalar
    i1 = __builtin_amdgcn_readfirstlane(i1);
    const bool active = i0 < i1;

    // resident query operands
    typename F::frag_t bq[HVS_QB][KS];
    typename F::thr_t theta[HVS_QB];
    uint32_t ra[HVS_QB], rb[HVS_QB];
#pragma unroll
    for (int qb = 0; qb < HVS_QB; ++qb) {
        const uint32_t slot = (active ? g : gq) * HVS_GROUP + qb * 32u + (lane & 31u);
        if constexpr (kI8)
            theta[qb] = B.thetai[slot];
        else
            theta[qb] = B.theta[slot];
        ra[qb] = B.ra[slot];
        rb[qb] = B.rb[slot];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            bq[qb][ks] = F::frag(B.bfrag[((size_t)((active ? g : gq) * HVS_QB + qb) * KS + ks) * 64u + lane]);
    }
    uint64_t* __restrict__ lbuf = sbuf[wv];
    uint32_t wcnt = 0;  // wave-uniform fill of lbuf
    uint32_t nblocks = 0;

    auto flush = [&]() {
        if (wcnt == 0u) return;
        uint32_t base = 0;
        if (lane == 0u) base = atomicAdd(&B.paircnt[g], wcnt);
        base = __builtin_amdgcn_readfirstlane(base);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (base + wcnt <= B.gcap) {
            for (uint32_t e = lane; e < wcnt; e += 64u) B.pairs[(size_t)g * B.gcap + base + e] = lbuf[e];
        } else if (lane == 0u) {
            B.goverflow[g] = 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        wcnt = 0;
    };

    // Tiles reach LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs): a
    // stage = 4 tiles = 28 chunks, 7 per wave.  Stage s+1 is in flight while stage s is multiplied;
    // s_waitcnt vmcnt(0) + barrier at the end of a stage both lands the next stage (every wave has
    // waited for its own chunks before any wave passes the barrier) and frees the current buffer
    // (every wave has finished its ds_reads of it).  One barrier per 4 tiles keeps the waves loosely coupled: a
    // wave that spends time on survivors of one tile catches up inside the stage.
    constexpr int kChunksPerWave = (STG * KM + HVS_WG_WAVES - 1) / HVS_WG_WAVES;
    auto issue_chunks = [&](uint32_t buf, uint32_t first_tile, int k0, int k1) {
        for (int k = k0; k < k1; ++k) {
            const uint32_t c = __builtin_amdgcn_readfirstlane(wv) + (uint32_t)HVS_WG_WAVES * (uint32_t)k;  // chunk of the stage
            if (c >= STG * KM) break;
            uint32_t tile = first_tile + c / KM;
            if (tile >= I1) tile = I1 - 1u;  // tail of the last stage: re-read a valid tile, never used
            const uint4* src = tiles + (size_t)tile * TILE_U4 + (c % KM) * 64u + lane;
            const uint4* dst = &stile[buf][(c / KM) * TILE_U4 + (c % KM) * 64u];
            // LDS byte address of the chunk (wave-uniform) goes to M0; the instruction adds lane*16.
            // Issued as inline asm on purpose: hipcc orders every later ds_read behind a
            // compiler-visible LDS-DMA with s_waitcnt vmcnt(0), which would serialise the prefetch
            // with the multiply; here the only wait is the explicit one before the stage barrier.
            const uint32_t lds_addr = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)dst);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src), "s"(lds_addr)
                         : "memory");
        }
    };
    auto issue_stage = [&](uint32_t buf, uint32_t first_tile) {
        issue_chunks(buf, first_tile, 0, kChunksPerWave);
        if constexpr (kI8) {
            // the stage's side data (per tile 128 B of tail dimensions + 128 B of accumulator inits, contiguous in
            // storage order): STG * 16 uint4, 64 per wave-instruction, issued by the last waves
            constexpr uint32_t kAuxInstr = (STG * HVS_I8_NRM_U4 + 63u) / 64u;
            const uint32_t wvs = __builtin_amdgcn_readfirstlane(wv);
            if (wvs + kAuxInstr >= HVS_WG_WAVES) {
                const uint32_t a = HVS_WG_WAVES - 1u - wvs;  // which 64-uint4 piece
                const uint32_t u = a * 64u + lane;
                uint32_t tile = first_tile + u / HVS_I8_NRM_U4;
                if (tile >= I1) tile = I1 - 1u;
                const uint4* src = nrm + (size_t)tile * HVS_I8_NRM_U4 + (u % HVS_I8_NRM_U4);
                const uint32_t lds_addr = __builtin_amdgcn_readfirstlane(
                    (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)&snrm[buf][a * 64u]);
                if (u < STG * HVS_I8_NRM_U4) {
                    uint32_t keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep)
                                 : "v"(src), "s"(lds_addr)
                                 : "memory");
                }
            }
        }
    };
    auto stage_barrier = [&]() {
#ifndef HVS_EXPERIMENT_NOSYNC
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA chunks have landed
        __syncthreads();
#endif
    };
    // ---- the tile loop ---------------------------------------------------------------------------------------
    // Per tile: A fragments (+ INT8 accumulator inits) from LDS, the 4 chains (two pairs, each interleaved
    // k-step by k-step), the 4 epilogues in the same scheduling region (their VALU work interleaves with the
    // MFMA issue), then -- rarely -- the survivors.  (A hand-pipelined variant that overlapped one pair's
    // epilogue with the other pair's chains across tiles measured 2-5 % slower: the second wave of the SIMD
    // already fills those gaps.)
    constexpr int HQ = HVS_QB / 2;
    static_assert(HVS_QB % 2 == 0, "query blocks are processed in two halves");
    typename F::frag_t af[KS];
    typename F::acc_t acc0;
    typename F::acc_t acc[HVS_QB];
    uint64_t hm[HVS_QB];
    uint32_t bp = 0;

    // A fragments and accumulator start of tile i from its stage buffer.  Start: 0 (BF16: the norm term sits in
    // k = 100..102) or the rows' nh (INT8): accumulator r of a lane is row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    // -> 4 broadcast b128 reads
    // `slot` = the tile's place in the two stage buffers (buffer * STG + tile within the stage): the caller steps it by one
    // per tile instead of deriving it from i (as hvs_k_filter_i8x16)
    const uint4* stile_flat = &stile[0][0];
    auto load_tile = [&](uint32_t slot) {
    #pragma unroll
        for (int ks = 0; ks < KM; ++ks) af[ks] = F::frag(stile_flat[slot * TILE_U4 + ks * 64 + lane]);
        if constexpr (kI8) {
            // 4th k-step: lanes 0..31 hold k = 96..111 of their row (4 real dimensions from the side data, then
            // zeros), lanes 32..63 hold k = 112..127 (zeros)
            const uint32_t buf = slot / STG, tt = slot % STG;
            const uint32_t* tailw = reinterpret_cast<const uint32_t*>(&snrm[buf][tt * HVS_I8_NRM_U4]);
            const int t4 = lane < 32u ? (int)tailw[lane & 31u] : 0;
            af[KM] = hvs_i32x4{t4, 0, 0, 0};
    #pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const hvs_i32x4 v = hvs_as_i32x4(snrm[buf][tt * HVS_I8_NRM_U4 + 8 + 2 * g4 + (lane >> 5)]);
                acc0[4 * g4 + 0] = v[0];
                acc0[4 * g4 + 1] = v[1];
                acc0[4 * g4 + 2] = v[2];
                acc0[4 * g4 + 3] = v[3];
            }
        } else {
            acc0 = typename F::acc_t{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    // the tile's block position: a scalar load (computing it -- runtime division by radix-1 -- measured 4 % slower)
    auto load_bp = [&](uint32_t i) { bp = __builtin_amdgcn_readfirstlane(bpos[i]); };
    // the chains of one pair, interleaved k-step by k-step
    auto chains = [&](int pair) {
    #pragma unroll
        for (int h = 0; h < HQ; ++h) acc[pair * HQ + h] = F::mfma(af[0], bq[pair * HQ + h][0], acc0);
    #pragma unroll
        for (int ks = 1; ks < KS; ++ks) {
    #pragma unroll
            for (int h = 0; h < HQ; ++h) acc[pair * HQ + h] = F::mfma(af[ks], bq[pair * HQ + h][ks], acc[pair * HQ + h]);
        }
    };
    // max chain + threshold (+ range test) of one pair on the tile at block position bpx.  hm[qb] = lanes
    // (queries) of the block with an accumulator at or above their threshold -- kept as SCALAR lane masks so
    // that every later "anything to do?" test is a scalar compare, not a vector compare + branch.
    auto epilogue = [&](int pair, uint32_t bpx, bool inner) {
    #pragma unroll
        for (int h = 0; h < HQ; ++h) {
            const int qb = pair * HQ + h;
            typename F::thr_t m = F::max2(F::max2(acc[qb][0], acc[qb][1]), acc[qb][2]);  // v_max3 chain
    #pragma unroll
            for (int r = 3; r < 15; r += 2) m = F::max2(F::max2(m, acc[qb][r]), acc[qb][r + 1]);
            m = F::max2(m, acc[qb][15]);
            hm[qb] = __ballot(m >= theta[qb]);
    #ifdef HVS_EXPERIMENT_NOHIT
            hm[qb] = __ballot(m == (typename F::thr_t)12345678);  // keeps the max chain alive, (almost) never true: ceiling experiment
    #endif
        }
    };
    // Survivors of one query block (about one wave-tile in ten at the top level, every tile at the low ones --
    // each level hands a query ~100 new candidates whatever its size).  The matrix waves do the minimum: a
    // per-lane bit mask of the 16 accumulators (straight-line v_cmp + v_addc_co) and ONE compaction of the lanes
    // with a non-zero mask into 8-byte SURVIVOR ENTRIES (hvs_entry_*): slot, block position, row half, mask.
    // Turning entries into (slot, row position) pairs, the per-row range test and everything else happens in
    // the re-scoring kernel, which waits on HBM anyway.
    auto survivors = [&](int qb, uint32_t bpx, bool inner) {
        uint32_t mask = hvs_hit_mask(acc[qb], theta[qb]);
        // (a lane still collecting its first 100 rows passes everything: keep it to the blocks of its own range)
        if (!inner) mask = ((bpx * 32u + 32u > ra[qb]) & (bpx * 32u < rb[qb])) ? mask : 0u;
        const uint32_t slot = g * HVS_GROUP + qb * 32u + (lane & 31u);
        const uint64_t nz = __ballot(mask != 0u);
        if (mask != 0u) lbuf[wcnt + hvs_prefix_count(nz)] = hvs_entry_make(slot, bpx, lane >> 5, mask);
        wcnt += (uint32_t)__popcll(nz);
        if (wcnt > 192u) flush();
    };

    // the positions every lane's range covers: [ra_max, rb_min) (lanes that can never hit do not count)
    uint32_t ra_max = 0u, rb_min = 0xFFFFFFFFu;
    #pragma unroll
    for (int qb = 0; qb < HVS_QB; ++qb) {
        const bool live = rb[qb] > ra[qb];
        ra_max = max(ra_max, live ? ra[qb] : 0u);
        rb_min = min(rb_min, live ? rb[qb] : 0xFFFFFFFFu);
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ra_max = max(ra_max, (uint32_t)__shfl_xor((int)ra_max, o));
        rb_min = min(rb_min, (uint32_t)__shfl_xor((int)rb_min, o));
    }
    ra_max = __builtin_amdgcn_readfirstlane(ra_max);
    rb_min = __builtin_amdgcn_readfirstlane(rb_min);

    // tile at a time: four chains, four epilogues, survivors
    const uint32_t nstage = hvs_ceil_div(I1 - I0, STG);
    issue_stage(0u, I0);
    stage_barrier();
    // Everything loaded so far (B fragments, thresholds, ranges) has landed -- the stage barrier waited for
    // vmcnt(0) -- but the compiler cannot see through that inline asm and would re-wait for those loads at
    // their first uses INSIDE the loop (`s_waitcnt vmcnt(0)` in the middle of every tile's MFMA block, which
    // with the LDS-DMA prefetch of the next stage in flight exposes the DMA latency once per stage).  A wait
    // the compiler does understand, once, here, clears its scoreboard.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    // Tile t: [LDS reads of its fragments] [28 / 16 matrix instructions] [epilogues] [survivors, rarely].  Round 3 took over from
    // hvs_k_filter_i8x16: the wave's tile range per stage computed once, the stage slot stepped instead of derived, the range
    // tests of edge blocks behind one branch, the hit tests as a tree (+2.7 % on FP16 / BF16 tiles, +3.7 % on 32x32x32 INT8
    // tiles); NOT its read order (HVS_H16_PREFETCH: the next tile's fragment reads under this tile's epilogue), which costs
    // 3 % here -- 7 KiB of fragments per tile and wave instead of 4, and the compiler's staggered chains already start the first
    // epilogue under the last matrix instructions.
    auto tile_body = [&](uint32_t bpx, bool inner, uint32_t inext, uint32_t slot_next) {
        chains(0);
        chains(1);
#if HVS_H16_PREFETCH
        __builtin_amdgcn_sched_barrier(0);
        load_tile(slot_next);  // (always: a conditional load makes the compiler copy the fragment registers per tile)
        load_bp(inext);
        __builtin_amdgcn_sched_barrier(0);
#endif
        epilogue(0, bpx, inner);
        epilogue(1, bpx, inner);
        if (!inner) {  // blocks at the edge of some lane's own range: one small branch for the 8 per-lane range compares
    #pragma unroll
            for (int qb = 0; qb < HVS_QB; ++qb) hm[qb] &= __ballot(bpx * 32u + 32u > ra[qb]) & __ballot(bpx * 32u < rb[qb]);  // (each compare IS a lane mask)
        }
        const uint64_t h01 = hm[0] | hm[1], h23 = hm[2] | hm[3];
        static_assert(HVS_QB == 4, "the hit tests are written for 4 query blocks");
        if ((h01 | h23) != 0ull) {
            if (h01 != 0ull) {
                if (hm[0] != 0ull) survivors(0, bpx, inner);
                if (hm[1] != 0ull) survivors(1, bpx, inner);
            }
            if (h23 != 0ull) {
                if (hm[2] != 0ull) survivors(2, bpx, inner);
                if (hm[3] != 0ull) survivors(3, bpx, inner);
            }
        }
    };
    for (uint32_t st = 0; st < nstage; ++st) {
#ifndef HVS_EXPERIMENT_NODMA
        if (st + 1u < nstage) issue_stage((st & 1u) ^ 1u, I0 + (st + 1u) * STG);
#endif
        // this wave's tiles of the stage: [t0, t1)
        const uint32_t s0 = I0 + st * STG;
        const uint32_t t0 = i0 > s0 ? i0 : s0;
        const uint32_t t1 = i1 < s0 + STG ? i1 : s0 + STG;  // (i1 <= I1)
        if (active && t0 < t1) {
            uint32_t slot = (st & 1u) * STG + (t0 - s0);
#if HVS_H16_PREFETCH
            load_tile(slot);
            load_bp(t0);
#endif
#pragma unroll 1
            for (uint32_t i = t0; i < t1; ++i) {
                ++nblocks;
#if !HVS_H16_PREFETCH
                // fragments and block position in front of the tile's own matrix block.  (Measured on FP16 tiles, against
                // this order: the reads under the previous tile's epilogue -3 %; only the block position's scalar load
                // under it -1 %, and -2.5 % more with a scheduling barrier between the matrix block and the epilogue -- the
                // compiler starts the first chain's epilogue between the last matrix instructions of the other three.)
                load_tile(slot);
                load_bp(i);
#endif
                wcnt = __builtin_amdgcn_readfirstlane(wcnt);
                const uint32_t bpx = bp;
                const bool inner = bpx * 32u >= ra_max && bpx * 32u + 32u <= rb_min;
                const bool more = i + 1u < t1;  // (last tile of the stage: re-read this one, unused)
#if HVS_H16_PREFETCH
                slot += more ? 1u : 0u;
#else
                slot += 1u;
#endif
                asm volatile("" : "+s"(slot));  // (otherwise the compiler re-derives it from i)
                tile_body(bpx, inner, more ? i + 1u : i, slot);
            }
        }
        stage_barrier();
    }
    if (active) {
        flush();
        if (lane == 0u) atomicAdd(&counters[1], (unsigned long long)nblocks * 32ull * HVS_GROUP);
    }
  }  // next work item (the last stage barrier has released the stage buffers)
}

// ---------------------------------------------------------------------------------------------
// hvs_k_filter_i8x16 -- the INT8 filter on v_mfma_i32_16x16x64_i8 (tile format HVS_FMT_I8X16).
//
// Same work decomposition, LDS-DMA staging, survivor buffers and bound as hvs_k_filter_mfma<HVS_FMT_I8>; the matrix
// shape differs.  In the filter-shaped loop (A fragments + accumulator inits re-read from LDS per tile, max/threshold
// epilogue, two waves per SIMD, random operands) the 16x16x64 shape sustains 1.16-1.17x the pair rate of 32x32x32 on
// this chip (scripts/mfma_shape_lab.hip: 12.8 vs 10.9 G 32x32 pair blocks/s) -- the chip holds a higher clock under
// the 16x16 instruction stream (MI355X_MICROARCH.md, DVFS give-back (7), reports the same for BF16).
//
// Per tile and wave: 4 ds_read_b128 fetch the A fragments (row block rb = 0,1 x k-step ks = 0,1), 2 ds_read_b128 the
// accumulator inits (a lane's 4 accumulators are 4 consecutive rows); against the wave's 8 resident sub-blocks of 16
// queries (B fragments in 64 VGPRs) 32 v_mfma_i32_16x16x64_i8 (2 row blocks x 8 sub-blocks x 2 k-steps) yield
// S[row][query] = qq.dq + nh exactly.  A lane's 8 accumulators per sub-block belong to ONE query: max chain, one
// compare against the lane's threshold, scalar hit masks; survivors leave as 8-byte entries (hvs_entry16_*).
// ---------------------------------------------------------------------------------------------
#define HVS_HITMASK8_ASM                                                                                                 \
    "v_cmp_ge_i32_e64 %1, %12, %13\n\tv_cmp_ge_i32_e64 %2, %11, %13\n\tv_cmp_ge_i32_e64 %3, %10, %13\n\t"                    \
    "v_addc_co_u32_e64 %0, %4, 0, 0, %1\n\tv_cmp_ge_i32_e64 %1, %9, %13\n\t"                                                \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\tv_cmp_ge_i32_e64 %2, %8, %13\n\t"                                            \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\tv_cmp_ge_i32_e64 %3, %7, %13\n\t"                                            \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\tv_cmp_ge_i32_e64 %1, %6, %13\n\t"                                            \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\tv_cmp_ge_i32_e64 %2, %5, %13\n\t"                                            \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\t"                                                                           \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\t"                                                                           \
    "s_nop 1\n\t"                                                                                                        \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %2"
// bit b = 4 rb + i of the result = (acc[rb][i] >= th)
__device__ __forceinline__ uint32_t hvs_hit_mask8(const hvs_i32x4& a0, const hvs_i32x4& a1, int th)
{
    uint32_t m;
    uint64_t s0, s1, s2, sc;
    asm volatile(HVS_HITMASK8_ASM
                 : "=&v"(m), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(sc)
                 : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "v"(th));
    return m;
}

// ---------------------------------------------------------------------------------------------
// Work-item lists of a batch (see HvsItems).  Segments of all levels are numbered consecutively (HvsSegs); per quad
// of groups the union block range and ordering; per (level, segment) the quads whose range meets it.
//   hvs_k_quad_ranges : per quad [block lo, block hi) over its groups (those that share the first group's ordering)
//   hvs_k_item_count  : one workgroup per (level, segment): how many quads cover it
//   hvs_k_item_scan   : exclusive prefix over all segments (one workgroup) + per-level offsets
//   hvs_k_item_fill   : the same sweep as the count, writing (segment << HVS_ITEM_QUAD_BITS | quad) in quad order
// ---------------------------------------------------------------------------------------------
struct HvsSegs {
    uint32_t first[17];  // first global segment number of each level; first[K + 1] = total
    uint32_t seg[17];    // row blocks per work item of each level (a power of two, 8 .. HVS_SEG)
    uint32_t K;
};
// Segment size per level: HVS_SEG blocks where the batch has enough quads to fill the chip, smaller (down to one LDS
// stage of 8 tiles) for small batches, so that a level of T blocks still makes ~4 work items per workgroup slot:
// with 10^4 queries (20 quads) a fixed 256-block segment left the lower levels of D = 10^6 with 20 items for 512 slots.
static inline HvsSegs hvs_make_segs(const HvsLevels& L, uint32_t nquads, uint32_t wg_slots, uint32_t items_per_slot = 4u)
{
    HvsSegs S{};
    S.K = L.K;
    uint32_t t = 0;
    for (uint32_t j = 0; j <= L.K; ++j) {
        const uint64_t T = L.off[j + 1] - L.off[j];
        uint64_t want = T * (uint64_t)(nquads ? nquads : 1u) / ((uint64_t)(items_per_slot ? items_per_slot : 1u) * (wg_slots ? wg_slots : 1u));
        uint32_t seg = 8u;
        while (seg < HVS_SEG && (uint64_t)seg * 2u <= want) seg *= 2u;
        S.seg[j] = seg;
        S.first[j] = t;
        t += hvs_ceil_div((uint32_t)T, seg);
    }
    for (uint32_t j = L.K + 1; j < 17u; ++j) {
        S.first[j] = t;
        S.seg[j] = HVS_SEG;
    }
    return S;
}

__global__ void hvs_k_quad_ranges(HvsBatch B, uint32_t nquads, uint32_t* __restrict__ qlo, uint32_t* __restrict__ qhi)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nquads) return;
    const uint32_t g0 = q * HVS_WG_WAVES, ord = B.gord[g0];
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t w = 0; w < HVS_WG_WAVES; ++w) {
        const uint32_t g = g0 + w;
        if (g >= B.ngroups || B.gord[g] != ord || B.gub[g] <= B.gua[g]) continue;
        const uint32_t a = B.gua[g] / 32u, b = hvs_ceil_div(B.gub[g], 32u);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    qlo[q] = lo < hi ? lo : 0u;
    qhi[q] = lo < hi ? hi : 0u;
}

// does quad range [blo, bhi) (blocks) meet segment `seg` of `level`?
__device__ __forceinline__ bool hvs_quad_meets(const HvsLevels& L, uint32_t level, uint32_t seg, uint32_t segsize, uint32_t blo,
                                               uint32_t bhi)
{
    if (bhi <= blo) return false;
    uint32_t lo, hi;
    hvs_level_run(L, level, blo, bhi, lo, hi);
    const uint32_t seg_lo = L.off[level] + seg * segsize;
    return seg_lo < hi && seg_lo + segsize > lo;
}

template <bool FILL>
__global__ __launch_bounds__(256) void hvs_k_item_sweep(HvsLevels L, HvsSegs S, uint32_t nquads, const uint32_t* __restrict__ qlo,
                                                        const uint32_t* __restrict__ qhi, uint32_t* __restrict__ segcnt,
                                                        const uint32_t* __restrict__ segoff, uint32_t* __restrict__ list)
{
    __shared__ uint32_t swave[4];
    __shared__ uint32_t sbase;
    const uint32_t gs = blockIdx.x;  // global segment number
    uint32_t level = 1;              // (level 0 is the exact seed's: no filter launch)
    while (level < S.K && gs >= S.first[level + 1]) ++level;
    if (gs < S.first[1]) return;
    const uint32_t seg = gs - S.first[level];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0u) sbase = FILL ? segoff[gs] : 0u;
    __syncthreads();
    for (uint32_t q0 = 0; q0 < nquads; q0 += 256u) {
        const uint32_t q = q0 + threadIdx.x;
        const bool hit = q < nquads && hvs_quad_meets(L, level, seg, S.seg[level], qlo[q], qhi[q]);
        const uint64_t m = __ballot(hit);
        if (lane == 0u) swave[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4u; ++w) {
            before += w < wv ? swave[w] : 0u;
            total += swave[w];
        }
        if (FILL && hit) list[sbase + before + hvs_prefix_count(m)] = (seg << HVS_ITEM_QUAD_BITS) | q;
        __syncthreads();
        if (threadIdx.x == 0u) sbase += total;
        __syncthreads();
    }
    if (!FILL && threadIdx.x == 0u) segcnt[gs] = sbase;
}

// exclusive prefix of segcnt[0 .. nseg) -> segoff[0 .. nseg]; lvloff[j] = segoff[S.first[j]].  One workgroup of 1024.
__global__ __launch_bounds__(1024) void hvs_k_item_scan(HvsSegs S, const uint32_t* __restrict__ segcnt, uint32_t* __restrict__ segoff,
                                                        uint32_t* __restrict__ lvloff)
{
    __shared__ uint32_t spart[1024];
    const uint32_t nseg = S.first[S.K + 1];
    const uint32_t per = hvs_ceil_div(nseg, 1024u);
    const uint32_t a = threadIdx.x * per, b = (a + per) < nseg ? (a + per) : nseg;
    uint32_t sum = 0;
    for (uint32_t i = a; i < b; ++i) sum += (i >= S.first[1]) ? segcnt[i] : 0u;
    spart[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024u; o <<= 1) {  // inclusive scan of the per-thread sums
        const uint32_t v = threadIdx.x >= o ? spart[threadIdx.x - o] : 0u;
        __syncthreads();
        spart[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? spart[threadIdx.x - 1u] : 0u;
    for (uint32_t i = a; i < b; ++i) {
        segoff[i] = run;
        run += (i >= S.first[1]) ? segcnt[i] : 0u;
    }
    if (threadIdx.x == 1023u) segoff[nseg] = spart[1023];
    __syncthreads();
    if (threadIdx.x <= S.K + 1u) lvloff[threadIdx.x] = threadIdx.x == S.K + 1u ? spart[1023] : 0u;
    __syncthreads();
    // (segoff is complete only after every thread's loop: read it back through global memory after a fence)
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x <= S.K) lvloff[threadIdx.x] = segoff[S.first[threadIdx.x]];
}

__global__ __launch_bounds__(64 * HVS_WG_WAVES, HVS_FILTER_OCC) void hvs_k_filter_i8x16(
    const uint4* __restrict__ tiles_ct, const uint4* __restrict__ tiles_t, const uint4* __restrict__ nrm_ct,
    const uint4* __restrict__ nrm_t, const uint32_t* __restrict__ bpos_ct, const uint32_t* __restrict__ bpos_t, HvsLevels L,
    uint32_t level, HvsBatch B, HvsItems W, unsigned long long* __restrict__ counters)
{
    constexpr int STG = HVS_STAGE_I8;
    constexpr int TILE_U4 = HVS_I8X16_TILE_U4;
    constexpr int NRM_U4 = HVS_I8X16_NRM_U4;
    constexpr int NSUB = HVS_GROUP / HVS_I8X16_QSUB;  // 8 sub-blocks of 16 queries per wave
    static_assert(STG * NRM_U4 == 64, "the side data of a stage is one LDS-DMA wave-instruction");
    static_assert((STG * HVS_I8X16_FRAGS) % HVS_WG_WAVES == 0, "chunks of a stage divide over the waves");
    __shared__ uint4 stile[2][STG * TILE_U4];  // 2 x 32 KiB
    __shared__ uint4 snrm[2][STG * NRM_U4];    // 2 x 1 KiB
    __shared__ uint64_t sbuf[HVS_WG_WAVES][256];
    __shared__ uint32_t srange[HVS_WG_WAVES][2];
    __shared__ uint32_t sitem;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t item0 = W.lvloff[level], nitems = W.lvloff[level + 1u] - item0;
  for (;;) {  // one work item per turn; every workgroup leaves when the level's list is exhausted
    if (threadIdx.x == 0u) sitem = atomicAdd(&W.cursor[level], 1u);
    __syncthreads();
    const uint32_t item = __builtin_amdgcn_readfirstlane(sitem);
    if (item >= nitems) break;  // uniform over the workgroup
    const uint32_t code = __builtin_amdgcn_readfirstlane(W.list[item0 + item]);
    const uint32_t quad = code & ((1u << HVS_ITEM_QUAD_BITS) - 1u), segment = code >> HVS_ITEM_QUAD_BITS;
    const uint32_t g = quad * HVS_WG_WAVES + wv;
    const uint32_t gq = quad * HVS_WG_WAVES;
    const uint32_t ord = B.gord[gq];
    const uint4* __restrict__ tiles = ord ? tiles_t : tiles_ct;
    const uint32_t* __restrict__ bpos = ord ? bpos_t : bpos_ct;
    const uint4* __restrict__ nrm = ord ? nrm_t : nrm_ct;
    const uint32_t seg_lo = L.off[level] + segment * W.segsize;
    uint32_t i0 = 0, i1 = 0;
    if (g < B.ngroups && B.gord[g] == ord) {
        uint32_t lo, hi;
        hvs_level_run(L, level, B.gua[g] / 32u, hvs_ceil_div(B.gub[g], 32u), lo, hi);
        if (seg_lo < hi && seg_lo + W.segsize > lo) {
            i0 = seg_lo > lo ? seg_lo : lo;
            i1 = (seg_lo + W.segsize) < hi ? (seg_lo + W.segsize) : hi;
        }
    }
    if (lane == 0u) {
        srange[wv][0] = i0 < i1 ? i0 : 0xFFFFFFFFu;
        srange[wv][1] = i0 < i1 ? i1 : 0u;
    }
    __syncthreads();
    uint32_t I0 = srange[0][0], I1 = srange[0][1];
#pragma unroll
    for (int w = 1; w < HVS_WG_WAVES; ++w) {
        I0 = srange[w][0] < I0 ? srange[w][0] : I0;
        I1 = srange[w][1] > I1 ? srange[w][1] : I1;
    }
    if (I0 >= I1) continue;  // uniform over the workgroup (cannot happen with a well-formed list)
    I0 = __builtin_amdgcn_readfirstlane(I0);
    I1 = __builtin_amdgcn_readfirstlane(I1);
    i0 = __builtin_amdgcn_readfirstlane(i0);
    i1 = __builtin_amdgcn_readfirstlane(i1);
    const bool active = i0 < i1;
    const uint32_t gg = active ? g : gq;

    // resident query operands: sub-block j, lane l <-> query slot 16 j + (l & 15)
    hvs_i32x4 bq[NSUB][2];
    int theta[NSUB];
    uint32_t ra[NSUB], rb[NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
        const uint32_t slot = gg * HVS_GROUP + j * HVS_I8X16_QSUB + (lane & 15u);
        theta[j] = B.thetai[slot];
        ra[j] = B.ra[slot];
        rb[j] = B.rb[slot];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bq[j][ks] = hvs_as_i32x4(B.bfrag[((size_t)(gg * NSUB + j) * 2u + ks) * 64u + lane]);
    }
    uint64_t* __restrict__ lbuf = sbuf[wv];
    uint32_t wcnt = 0;
    uint32_t nblocks = 0;

    auto flush = [&]() {
        if (wcnt == 0u) return;
        uint32_t base = 0;
        if (lane == 0u) base = atomicAdd(&B.paircnt[g], wcnt);
        base = __builtin_amdgcn_readfirstlane(base);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (base + wcnt <= B.gcap) {
            for (uint32_t e = lane; e < wcnt; e += 64u) B.pairs[(size_t)g * B.gcap + base + e] = lbuf[e];
        } else if (lane == 0u) {
            B.goverflow[g] = 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        wcnt = 0;
    };

    // LDS-DMA: a stage = 8 tiles x 4 fragments = 32 chunks of 1 KiB (8 per wave) + 1 chunk of accumulator inits
    constexpr int kChunksPerWave = STG * HVS_I8X16_FRAGS / HVS_WG_WAVES;
    auto dma = [&](const uint4* src, const uint4* dst) {
        const uint32_t lds_addr =
            __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)dst);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(lds_addr)
                     : "memory");
    };
    auto issue_stage = [&](uint32_t buf, uint32_t first_tile) {
        const uint32_t wvs = __builtin_amdgcn_readfirstlane(wv);
#pragma unroll
        for (int k = 0; k < kChunksPerWave; ++k) {
            const uint32_t c = wvs + (uint32_t)HVS_WG_WAVES * (uint32_t)k;
            uint32_t tile = first_tile + c / HVS_I8X16_FRAGS;
            if (tile >= I1) tile = I1 - 1u;  // tail of the last stage: re-read a valid tile, never used
            dma(tiles + (size_t)tile * TILE_U4 + (c % HVS_I8X16_FRAGS) * 64u + lane, &stile[buf][c * 64u]);
        }
        if (wvs == HVS_WG_WAVES - 1u) {
            uint32_t tile = first_tile + lane / NRM_U4;
            if (tile >= I1) tile = I1 - 1u;
            dma(nrm + (size_t)tile * NRM_U4 + (lane % NRM_U4), &snrm[buf][0]);
        }
    };
    auto stage_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    hvs_i32x4 af[HVS_I8X16_FRAGS];
    hvs_i32x4 nh[2];
    hvs_i32x4 acc[2][NSUB];
    uint64_t hm[NSUB];
    uint32_t bp = 0;
    // tile i of the staged run; `slot` = its place in the two stage buffers (buffer * STG + tile within the stage): the
    // caller steps it by one per tile instead of deriving it from i (scalar instructions are not free in this loop)
    const uint4* stile_flat = &stile[0][0];
    const uint4* snrm_flat = &snrm[0][0];
    auto load_tile = [&](uint32_t i, uint32_t slot) {
#pragma unroll
        for (int f = 0; f < HVS_I8X16_FRAGS; ++f) af[f] = hvs_as_i32x4(stile_flat[slot * TILE_U4 + f * 64 + lane]);
        const uint32_t nslot = (slot / STG) * 64u + (slot % STG) * NRM_U4;  // (snrm rows are 64 uint4 per stage buffer)
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) nh[r2] = hvs_as_i32x4(snrm_flat[nslot + 4 * r2 + (lane >> 4)]);
        bp = __builtin_amdgcn_readfirstlane(bpos[i]);
    };
    // The 32 matrix instructions of a tile in two groups of 16 (k-step 0 of every accumulator block, then k-step 1,
    // accumulating IN PLACE): each instruction of the second group depends on one issued 16 instructions (>= 256
    // cycles) earlier, so none waits.  Written as inline asm on purpose: left to the compiler, each block's two k-steps
    // are paired back to back with the block's max chain right behind them (2-7 wait states per instruction, ~40 s_nop
    // per tile), or -- with scheduling barriers -- the second group is not accumulated in place, which costs 64 more
    // registers and 12 v_mov_b64 per tile to shuffle the next tile's fragments.  The compiler does not see matrix
    // instructions inside asm, so the wait states it would insert are guaranteed by construction instead:
    //   * operands come from LDS / global loads (s_waitcnt is operand-based and still inserted), never from a VALU
    //     instruction right in front of the block;
    //   * a k-step-1 instruction reads an accumulator written 16 matrix instructions earlier;
    //   * the epilogue reads the accumulators in issue order (block j = 0 first); the youngest one (j = 7) is read
    //     after >= 6 LDS reads and >= 28 vector instructions, far beyond the 4-pass result latency;
    //   * the fragment / init registers are overwritten only by LDS reads issued after the last matrix instruction.
    auto chains = [&]() {
#if HVS_FILTER_SETPRIO == 1
        __builtin_amdgcn_s_setprio(3);  // A/B: the multiplying wave keeps the matrix pipe
#elif HVS_FILTER_SETPRIO == 2
        __builtin_amdgcn_s_setprio(0);  // A/B: the wave in its epilogue goes first
#endif
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=&v"(acc[r2][j]) : "v"(af[2 * r2]), "v"(bq[j][0]), "v"(nh[r2]));
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[r2][j]) : "v"(af[2 * r2 + 1]), "v"(bq[j][1]));
#if HVS_FILTER_SETPRIO == 1
        __builtin_amdgcn_s_setprio(0);
#elif HVS_FILTER_SETPRIO == 2
        __builtin_amdgcn_s_setprio(3);
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    auto imax = [](int a, int b) { return a > b ? a : b; };
    // ONE copy of the tile body: blocks inside every lane's own range (`inner`, wave-uniform, the common case) skip
    // the 16 per-lane range compares through a small branch instead of a second copy of the whole body (with two
    // copies the register allocator shuffled the 24 fragment registers at every loop back-edge)
    auto epilogue = [&](uint32_t bpx, bool inner) {
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
#ifdef HVS_EXPERIMENT_NOEPI
            int m = imax(acc[0][j][0], acc[1][j][3]);  // ceiling experiment: 1 instead of 4 instructions per sub-block
#else
            int m = imax(imax(acc[0][j][0], acc[0][j][1]), acc[0][j][2]);  // v_max3 chain
            m = imax(imax(m, acc[0][j][3]), acc[1][j][0]);
            m = imax(imax(m, acc[1][j][1]), acc[1][j][2]);
            m = imax(m, acc[1][j][3]);
#endif
            hm[j] = __ballot(m >= theta[j]);
#ifdef HVS_EXPERIMENT_NOHIT
            hm[j] = __ballot(m == 0x7fffff37);  // keeps the max chain alive, (almost) never true: ceiling experiment
#endif
            if (j == NSUB / 2 - 1) __builtin_amdgcn_sched_barrier(0);  // (the younger accumulators are read last)
        }
        if (!inner) {
#pragma unroll
            for (int j = 0; j < NSUB; ++j) hm[j] &= __ballot(bpx * 32u + 32u > ra[j]) & __ballot(bpx * 32u < rb[j]);
        }
    };
    // hm[j] is EXACTLY the set of lanes that have an entry to write: a lane's max reached its threshold iff one of its 8
    // accumulators did (mask != 0), and outside `inner` tiles hm[j] already carries the lane's range test -- so the
    // lane set needs no second compare, ballot or range test: the scalar mask becomes the execution mask directly
    // (s_and_saveexec) and the prefix count / popcount come from it
    auto survivors = [&](int j, uint32_t bpx) {
        const uint32_t mask = hvs_hit_mask8(acc[0][j], acc[1][j], theta[j]);
        const uint64_t nz = hm[j];
        if (__builtin_amdgcn_inverse_ballot_w64(nz)) {
            const uint32_t slot = g * HVS_GROUP + j * HVS_I8X16_QSUB + (lane & 15u);
            lbuf[wcnt + hvs_prefix_count(nz)] = hvs_entry16_make(slot, bpx, lane >> 4, mask);
        }
        wcnt += (uint32_t)__popcll(nz);
        if (wcnt > 192u) flush();
    };

    uint32_t ra_max = 0u, rb_min = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
        const bool live = rb[j] > ra[j];
        ra_max = max(ra_max, live ? ra[j] : 0u);
        rb_min = min(rb_min, live ? rb[j] : 0xFFFFFFFFu);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ra_max = max(ra_max, (uint32_t)__shfl_xor((int)ra_max, o));
        rb_min = min(rb_min, (uint32_t)__shfl_xor((int)rb_min, o));
    }
    ra_max = __builtin_amdgcn_readfirstlane(ra_max);
    rb_min = __builtin_amdgcn_readfirstlane(rb_min);

    const uint32_t nstage = hvs_ceil_div(I1 - I0, STG);
    issue_stage(0u, I0);
    stage_barrier();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): clears the compiler's scoreboard (see hvs_k_filter_mfma)
    // Tile t: [32 matrix instructions] [LDS reads of tile t+1's fragments, when it sits in the same stage] [epilogue of
    // tile t] [survivors, rarely].  The fragment registers are dead once the last matrix instruction has issued, so the
    // next tile's reads travel under the epilogue's ~50 vector instructions instead of in front of the next matrix block.
    auto tile_body = [&](uint32_t bpx, bool inner, uint32_t inext, uint32_t slot_next) {
        chains();
        load_tile(inext, slot_next);  // (always: a conditional load would make the compiler copy the 24 fragment registers per tile)
        __builtin_amdgcn_sched_barrier(0);
        epilogue(bpx, inner);
#if HVS_HIT_TREE
        // most tiles with a hit have it in ONE sub-block: find it through a tree of scalar ORs (3 tests) instead of 8
        static_assert(NSUB == 8, "the hit tree is written for 8 sub-blocks");
        const uint64_t h01 = hm[0] | hm[1], h23 = hm[2] | hm[3], h45 = hm[4] | hm[5], h67 = hm[6] | hm[7];
        const uint64_t lo = h01 | h23, hi = h45 | h67;
        if ((lo | hi) != 0ull) {
            if (lo != 0ull) {
                if (h01 != 0ull) {
                    if (hm[0] != 0ull) survivors(0, bpx);
                    if (hm[1] != 0ull) survivors(1, bpx);
                }
                if (h23 != 0ull) {
                    if (hm[2] != 0ull) survivors(2, bpx);
                    if (hm[3] != 0ull) survivors(3, bpx);
                }
            }
            if (hi != 0ull) {
                if (h45 != 0ull) {
                    if (hm[4] != 0ull) survivors(4, bpx);
                    if (hm[5] != 0ull) survivors(5, bpx);
                }
                if (h67 != 0ull) {
                    if (hm[6] != 0ull) survivors(6, bpx);
                    if (hm[7] != 0ull) survivors(7, bpx);
                }
            }
        }
#else
        uint64_t any = 0;
#pragma unroll
        for (int j = 0; j < NSUB; ++j) any |= hm[j];
        if (any != 0ull) {
#pragma unroll
            for (int j = 0; j < NSUB; ++j)
                if (hm[j] != 0ull) survivors(j, bpx);
        }
#endif
    };
    for (uint32_t st = 0; st < nstage; ++st) {
        if (st + 1u < nstage) issue_stage((st & 1u) ^ 1u, I0 + (st + 1u) * STG);
        // this wave's tiles of the stage: [t0, t1)
        const uint32_t s0 = I0 + st * STG;
        const uint32_t t0 = i0 > s0 ? i0 : s0;
        const uint32_t t1 = i1 < s0 + STG ? i1 : s0 + STG;  // (i1 <= I1)
        if (active && t0 < t1) {
            uint32_t slot = (st & 1u) * STG + (t0 - s0);
            load_tile(t0, slot);
#pragma unroll 1
            for (uint32_t i = t0; i < t1; ++i) {
                ++nblocks;
                wcnt = __builtin_amdgcn_readfirstlane(wcnt);
                const uint32_t bpx = bp;
                const bool inner = bpx * 32u >= ra_max && bpx * 32u + 32u <= rb_min;
                const bool more = i + 1u < t1;  // (last tile of the stage: re-read this one, unused)
                slot += more ? 1u : 0u;
                asm volatile("" : "+s"(slot));  // (otherwise the compiler re-derives it from i: ~8 scalar instructions per tile)
                tile_body(bpx, inner, more ? i + 1u : i, slot);
            }
        }
        stage_barrier();
    }
    if (active) {
        flush();
        if (lane == 0u) atomicAdd(&counters[1], (unsigned long long)nblocks * 32ull * HVS_GROUP);
    }
  }  // next work item (the last stage barrier has released the stage buffers)
}

// ---------------------------------------------------------------------------------------------
// hvs_k_rescore -- exact-order distances of the filter's survivors, key appended to the slot's list.
//
// Front end (per wave, 64 survivor entries at a time, lane = entry): the k-th set bit of every entry's mask is
// turned into a (slot, row position) pair, range-tested against the slot's own position range and packed into a
// wave-private list in LDS; rounds repeat while any entry has bits left (usually one round).
// Back end: FOUR lanes per pair, lane t playing AVX lanes 2t and 2t+1 of the reference (optimized_impl.h:96-125) as one
// packed f32 pair: it accumulates dims 8b+2t, 8b+2t+1 for b = 0..11 (and 96..99 into accumulators 4..7: the masked
// tail) in that order, then the horizontal sum ((a0+a4)+(a1+a5))+((a2+a6)+(a3+a7)) runs across the 4 lanes (xor 2,
// x + y, xor 1; f32 addition is commutative, so lane 0 ends with the same bits as the sequential hvs_exact_dist).
// The point is the memory access: one load instruction of a wave reads 16 rows x 32 contiguous bytes, and the
// four instructions that walk one 128-byte line follow each other directly.  (One lane per pair -- 64 rows
// x 8 bytes per instruction, each line revisited by 16 instructions spread over the whole row walk --
// re-fetched lines from L2/HBM many times; eight lanes per pair with scalar math, round 1's form, issued twice
// the vector and load instructions per pair.)
// ---------------------------------------------------------------------------------------------
// E16: entries of the 16x16 tile format (hvs_entry16_*) instead of the 32x32 formats' (hvs_entry_*)
template <bool E16>
__global__ __launch_bounds__(64 * HVS_RESCORE_WAVES) void hvs_k_rescore(const float* __restrict__ D, uint32_t n, uint32_t sn, const float* __restrict__ Q,
                                                     HvsBatch B, const uint32_t* __restrict__ perm_ct,
                                                     const uint32_t* __restrict__ perm_t,
                                                     unsigned long long* __restrict__ counters)
{
    // the group's 128 query vectors are staged in LDS once per block (51 KB)
    __shared__ float sq[HVS_GROUP][HVS_NDIM];
    __shared__ uint64_t slist[HVS_RESCORE_WAVES][64];  // wave-private (slot << 32 | position) pairs of one round
    const uint32_t g = blockIdx.y;
    uint32_t np = B.paircnt[g];
    // a group whose entry list overflowed holds unwritten entries past the failed flush: none of
    // them are used, all of its queries are re-run by the exact engine
    if (np > B.gcap || B.goverflow[g]) np = 0;
    if (blockIdx.x == 0u && threadIdx.x == 0u && B.goverflow[g]) {
        for (uint32_t s = 0; s < HVS_GROUP; ++s) hvs_flag_fail(B.fail_code, B.overflow, g * HVS_GROUP + s);
    }
    if (blockIdx.x * (64u * HVS_RESCORE_WAVES) >= np) return;  // uniform over the block
    for (uint32_t e = threadIdx.x; e < HVS_GROUP * (HVS_NDIM / 4); e += blockDim.x) {
        const uint32_t ql = e / (HVS_NDIM / 4), c4 = e % (HVS_NDIM / 4);
        const uint32_t qi = B.qid[g * HVS_GROUP + ql];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qi != 0xFFFFFFFFu) v = *reinterpret_cast<const float4*>(Q + (size_t)qi * HVS_QCOLS + 4 + 4 * c4);
        *reinterpret_cast<float4*>(&sq[ql][4 * c4]) = v;
    }
    __syncthreads();
    const uint32_t* __restrict__ perm = B.gord[g] ? perm_t : perm_ct;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint64_t* list = slist[w];
    uint32_t npairs = 0;  // wave-uniform
    auto emask = [](uint64_t e) -> uint32_t { return E16 ? hvs_entry16_mask(e) : hvs_entry_mask(e); };
    auto epos = [](uint64_t e, uint32_t r) -> uint32_t { return E16 ? hvs_entry16_pos(e, r) : hvs_entry_pos(e, r); };

    // exact distances of list[0..cnt) = (slot << 32 | row id).  FOUR lanes per pair, lane t holding the reference's AVX
    // accumulators (2t, 2t+1) as one packed f32 pair: per 8-wide step it loads dims 8b+2t, 8b+2t+1 of the row (one
    // 8-byte load; a wave instruction still reads whole 32-byte row segments) and of the query (LDS), then v_pk_add
    // (d - q), v_pk_mul, v_pk_add -- each element one IEEE operation, the same bits as the scalar form
    // (optimized_impl.h:96-125).  The masked tail puts dims 96..99 into accumulators 4..7 = lanes 2 and 3.  hsum
    // ((a0+a4)+(a1+a5)) + ((a2+a6)+(a3+a7)) (optimized_impl.h:37-47): pair + pair of lane t^2, x + y, lane 0 + lane 1.
    // Half the vector and load instructions per pair of the 8-lane form (the low levels, whose rows sit in L2, are
    // instruction-bound).  HVS_RESCORE_UNROLL groups of 16 pairs per pass keep 32 rows in flight per wave.
    constexpr int kUn = HVS_RESCORE_UNROLL / 2 > 0 ? HVS_RESCORE_UNROLL / 2 : 1;
    const uint32_t t4 = lane & 3u;
    // Accepted candidates take a place in their slot's list with a RETURNING atomic (~1-2 us).  Waiting for it inside
    // the pass made every pass of 16-32 pairs cost one atomic round trip (the floor of the low levels, whose rows come
    // from L2: 3.2 ms per level).  The key is therefore stored one pass later: the atomic travels under the next pass's
    // row loads.  `retire` is called at the start of the next pass and once more at the end of the kernel.
    uint32_t pend_k[kUn], pend_slot[kUn];
    uint64_t pend_key[kUn];
    bool pend[kUn];
    #pragma unroll
    for (int u = 0; u < kUn; ++u) pend[u] = false;
    auto retire = [&]() {
    #pragma unroll
        for (int u = 0; u < kUn; ++u) {
            if (pend[u]) {
                if (pend_k[u] < B.fcap)
                    B.cand[(size_t)pend_slot[u] * B.fcap + pend_k[u]] = pend_key[u];
                else
                    hvs_flag_fail(B.fail_code, B.overflow, pend_slot[u]);
            }
            pend[u] = false;
        }
    };
    auto score_list = [&](uint32_t cnt) {
        for (uint32_t p0 = 0; p0 < cnt; p0 += 16u * kUn) {  // wave-uniform
            uint32_t slot[kUn], id[kUn];
            bool ok[kUn];
            hvs_f2 dk[kUn][13];
    #pragma unroll
            for (int u = 0; u < kUn; ++u) {
                const uint32_t pi = p0 + 16u * u + (lane >> 2);
                const uint64_t pr = list[pi < cnt ? pi : 0u];
                slot[u] = (uint32_t)(pr >> 32);
                id[u] = (uint32_t)pr;
                // sampled prefix (sample_proportion < 1): the filter does not know about it
                ok[u] = pi < cnt && id[u] < sn;
            }
    #pragma unroll
            for (int u = 0; u < kUn; ++u) {
                const hvs_f2* __restrict__ dv = reinterpret_cast<const hvs_f2*>(D + (size_t)(ok[u] ? id[u] : 0u) * HVS_DCOLS + 2);
    #pragma unroll
                for (int b = 0; b < 12; ++b) dk[u][b] = dv[4 * b + t4];
                dk[u][12] = dv[46u + (t4 | 2u)];  // dims 96,97 (lanes 0, 2) / 98,99 (lanes 1, 3); used by lanes 2, 3
            }
            retire();  // the previous pass's candidates: their atomics have long returned
    #pragma unroll
            for (int u = 0; u < kUn; ++u) {
                const hvs_f2* qv = reinterpret_cast<const hvs_f2*>(&sq[slot[u] - g * HVS_GROUP][0]);
                hvs_f2 acc = hvs_f2{0.0f, 0.0f};
    #pragma unroll
                for (int b = 0; b < 12; ++b) {
                    hvs_f2 t = dk[u][b] - qv[4 * b + t4];
                    t = t * t;
                    acc = acc + t;
                }
                {
                    hvs_f2 t = dk[u][12] - qv[46u + (t4 | 2u)];
                    t = t * t;
                    const hvs_f2 with_tail = acc + t;
                    acc = t4 >= 2u ? with_tail : acc;  // the masked tail feeds accumulators 4..7 only
                }
                // lanes t and t^2 hold (a_2t, a_2t+1) and (a_2t+4, a_2t+5) resp. the other way round
                const hvs_f2 sm = acc + hvs_f2{__shfl_xor(acc.x, 2), __shfl_xor(acc.y, 2)};
                const float am = sm.x + sm.y;                 // lane 0: (a0+a4)+(a1+a5), lane 1: (a2+a6)+(a3+a7)
                const float dist = am + __shfl_xor(am, 1);
                pend[u] = ok[u] && t4 == 0u && dist <= B.tau[slot[u]];
                pend_slot[u] = slot[u];
                pend_key[u] = hvs_make_key(dist, id[u]);
                if (pend[u]) pend_k[u] = atomicAdd(&B.candcnt[slot[u]], 1u);  // (result used by the next retire)
            }
        }
    };
    // one round: the lanes with `c` put (slot, id) into the wave's list, then the list is scored
    auto round = [&](bool c, uint32_t slot, uint32_t id) {
        const uint64_t cm = __ballot(c);
        const uint32_t cnt = (uint32_t)__popcll(cm);
        if (c) list[hvs_prefix_count(cm)] = ((uint64_t)slot << 32) | id;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        npairs += cnt;
        score_list(cnt);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the list is rewritten by the next round
    };
    // Software pipeline over chunks of 64 entries (lane = entry): the entry of chunk c+2 and, for chunk c+1, the
    // slot's range and the row id of its FIRST set bit (a dependent random load: entry -> perm[pos]) are fetched
    // while chunk c is scored, so that the chain entry -> row id -> row is not paid serially per chunk.
    const uint32_t stride = gridDim.x * (64u * HVS_RESCORE_WAVES);
    const uint32_t first = (blockIdx.x * HVS_RESCORE_WAVES + w) * 64u;
    auto load_entry = [&](uint32_t base) -> uint64_t {
        const uint32_t ei = base + lane;
        return (base < np && ei < np) ? B.pairs[(size_t)g * B.gcap + ei] : 0ull;  // (mask 0: nothing)
    };
    struct Staged {
        uint64_t ent;
        uint32_t ra, rb, pos0, id0;
    };
    auto stage = [&](uint64_t ent) -> Staged {
        Staged t;
        t.ent = ent;
        t.ra = t.rb = t.id0 = 0u;
        const uint32_t m = emask(ent);
        t.pos0 = epos(ent, (uint32_t)__builtin_ctz(m | 0x10000u) & 15u);
        if (m != 0u) {
            const uint32_t es = hvs_entry_slot(ent);
            t.ra = B.ra[es];
            t.rb = B.rb[es];
            t.id0 = t.pos0 < n ? perm[t.pos0] : 0u;  // (bits of padding rows past the end of the ordering)
        }
        return t;
    };
    Staged cur = stage(load_entry(first));
    uint64_t ent_next = load_entry(first + stride);
    for (uint32_t base = first; base < np; base += stride) {  // wave-uniform
        const Staged nxt = stage(ent_next);                    // loads in flight while `cur` is scored
        ent_next = load_entry(base + 2u * stride);
        const uint32_t eslot = hvs_entry_slot(cur.ent);
        uint32_t mask = emask(cur.ent);
        round((mask != 0u) & (cur.pos0 >= cur.ra) & (cur.pos0 < cur.rb), eslot, cur.id0);
        mask &= mask - 1u;
        while (__ballot(mask != 0u) != 0ull) {  // entries with more than one bit (few)
            const uint32_t r = (uint32_t)__builtin_ctz(mask | 0x10000u);
            mask &= mask - 1u;
            const uint32_t pos = epos(cur.ent, r & 15u);
            const bool c = (r < 16u) & (pos >= cur.ra) & (pos < cur.rb);
            round(c, eslot, c ? perm[pos] : 0u);
        }
        cur = nxt;
    }
    retire();
    if (lane == 0u && npairs) atomicAdd(&counters[2], (unsigned long long)npairs);
}

// Order statistic of a guessed threshold as a function of the fraction F of the query's rows seen so far (host:
// plan_guess in hvs.hip): m[i] belongs to F = 2^(-i/8); a query looks up the next smaller grid value of its own F.
#define HVS_GUESS_STEPS 168  // F down to 2^-20.9
struct HvsGuessTable {
    uint16_t m[HVS_GUESS_STEPS];
    uint16_t floor_m;   // smallest order statistic used at all
    uint16_t last_m;    // != 0: the order statistic of the LAST level, whatever F is (retry batches: k, the proven threshold)
};

// level of a block: the first level whose stride divides it
__host__ __device__ static inline uint32_t hvs_block_level(const HvsLevels& L, uint32_t b)
{
    uint32_t j = 0;
    while (j < L.K && (b & (L.stride[j] - 1u)) != 0u) ++j;  // (strides are powers of two)
    return j;
}

// rows of position range [a, b) that lie in levels < `level` (exact: whole blocks of the level runs, less the parts of the
// range's first and last block that are outside it)
__host__ __device__ static inline uint32_t hvs_rows_seen_before(const HvsLevels& L, uint32_t level, uint32_t a, uint32_t b)
{
    if (b <= a) return 0u;
    const uint32_t fb = a / 32u, lb = (b - 1u) / 32u;
    uint32_t rows = 0;
    for (uint32_t j = 0; j < level; ++j) {
        uint32_t lo, hi;
        hvs_level_run(L, j, fb, lb + 1u, lo, hi);
        rows += (hi - lo) * 32u;
    }
    if (hvs_block_level(L, fb) < level) rows -= a - fb * 32u;
    if (hvs_block_level(L, lb) < level) rows -= (lb + 1u) * 32u - b;
    return rows;
}

// order statistic for the threshold of `level` for a query with position range [a, b)
__device__ __forceinline__ uint32_t hvs_guess_m(const HvsLevels& L, const HvsGuessTable& G, uint32_t level, uint32_t a, uint32_t b,
                                                uint32_t knn)
{
    if (level == L.K && G.last_m) return G.last_m < knn ? G.last_m : knn;
    // a range of k rows or fewer can never hold k rows below any threshold: a guess there buys nothing (every row of the range
    // is kept anyway) and the final check `k keys held` would send the query to a retry batch
    if (b - a <= knn) return knn;
    const uint32_t seen = hvs_rows_seen_before(L, level, a, b);
    if (seen == 0u || b <= a) return knn;
    const float lf = __log2f((float)(b - a) / (float)seen);  // -log2 F >= 0
    int idx = (int)ceilf(8.0f * lf - 0.02f);
    idx = idx < 0 ? 0 : (idx >= HVS_GUESS_STEPS ? HVS_GUESS_STEPS - 1 : idx);
    uint32_t m = G.m[idx];
    m = m < G.floor_m ? G.floor_m : m;
    return m < knn ? m : knn;
}

// ---------------------------------------------------------------------------------------------
// hvs_k_merge -- per slot (one wave): top-k := k smallest keys of (top U cand); the threshold tau of the NEXT level
// and the filter threshold theta that belongs to it.  With `final` it verifies the answer, pads from the end of D
// (optimized_parallel.hpp:149-157), rank-sorts and writes the answer at the query's own index.
//
// Guessed thresholds.  A level multiplies the rows a query has seen by its radix r.  With the PROVEN threshold (tau =
// the k-th smallest distance seen so far, what the reference's Knn::check_add compares against, optimized_impl.h:301)
// the level hands k (r - 1) rows to the exact kernel -- which is why round 2 doubled (r = 2, 14 levels, ~k rows per
// query and level).  The rows seen so far are a systematic sample (every r-th block) of the rows seen after the level,
// so the k-th smallest distance AFTER the level is close to the (k / r)-th smallest BEFORE it.  The merge in front of a
// level therefore sets
//     tau_next = min(tau_now, m-th smallest distance held)            (m = m_next <= k; unchanged while fewer are held)
// with m the smallest order statistic whose chance of leaving fewer than k rows of the WHOLE range below tau is under a
// target (hvs.hip, plan_guess): if a fraction F of the query's rows has been seen, the number of rows of the whole
// range below the m-th smallest seen distance is m + NegBin(m, F).  k = 100, target 10^-5: F = 1/4 (in front of the
// last level of radix 4) -> m = 45, the level hands ~135 rows to the exact kernel instead of 300; F = 1/64 -> m = 10;
// F = 1/1024 -> m = 4 (10^-3, the target of large batches: 40 / 7 / 3).  F is the query's own (hvs_guess_m: a narrow predicate range sees a different share of its rows
// than the level radices say).  Nothing is taken on trust:
//   * tau only ever decreases, every row with exact distance <= tau of its level reaches the exact kernel (the filter's
//     bound, see theta below) and the top-k truncation only drops keys above k kept ones -- so after the last level
//     the list holds EVERY row of the query's range with distance <= tau_last;
//   * the final merge checks that the k-th smallest key it holds is <= tau_last (or that tau never became finite:
//     nothing was ever discarded).  Then the k smallest keys held are the k smallest of the whole range, bit for bit
//     what the proven threshold gives.  Otherwise the query is flagged (B.fail_code) and run again in a batch whose
//     last level uses m = k.
//
// theta: a row can be discarded when its exact-order distance R is certainly > tau.
//   R >= T (1 - g), T = |q-d|^2 (real), g = 20 * 2^-24 (f32 roundings of the reference order)
//   T >= |q|^2 - 2 s~ - 2 (mu + rho + |q| E_D + e_q NB_D)
//        s~  = MFMA value of  bf16(q).bf16(d) + h0+h1+h2,     h ~ -|d|^2/2
//        mu  = bound on the MFMA accumulation error, rho = |  |d|^2/2 + h0+h1+h2 |
//        |q.d - bf16(q).bf16(d)| <= |q| E_D + e_q NB_D   (Cauchy-Schwarz; E_D, NB_D row maxima)
//   => discard iff  s~ < theta := (|q|^2 - tau (1 + 2g)) / 2 - (mu + rho + |q| E_D + e_q NB_D) - slack,
//      slack = 1e-9 (|q|^2 + tau + band) covering the f64 evaluation of theta itself
// ---------------------------------------------------------------------------------------------
// (FINAL as a template parameter: the padding path's exact-order distance costs 70 VGPRs that would halve the
// occupancy of the latency-bound merges before it)
template <bool FINAL, int CAP>
__global__ __launch_bounds__(256) void hvs_k_merge(const float* __restrict__ D, uint32_t n, const float* __restrict__ Q,
                                                   HvsBatch B, const HvsBounds* __restrict__ bounds, int pad,
                                                   uint32_t* __restrict__ out_ids, float* __restrict__ out_dists, int fmt,
                                                   const HvsQuant* __restrict__ qz, HvsLevels L, uint32_t level_next,
                                                   HvsGuessTable G)
{
    __shared__ uint64_t sbuf[4][CAP];
    __shared__ uint32_t shist[4][256];  // digit histograms of the radix select
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t slot = blockIdx.x * 4u + w;
    const uint32_t knn = B.knn;
    if (slot >= B.nslots) return;
    // the group's survivor-entry list starts empty at the next level (this level's re-scoring is complete: stream order)
    if (!FINAL && lane == 0u && (slot % HVS_GROUP) == 0u) B.paircnt[slot / HVS_GROUP] = 0;
    const uint32_t qi = B.qid[slot];
    if (qi == 0xFFFFFFFFu) return;
    uint64_t* buf = sbuf[w];
    uint32_t m = B.candcnt[slot];
    if (m > B.fcap) m = B.fcap;
    // nothing new at this level: top-k, tau and theta stand.  (tau is not re-derived for `level_next` from the keys held: the
    // order statistic only grows from level to level, so the m_next-th smallest of an unchanged list is >= the tau already
    // set, and tau never increases -- min(tau, .) would return tau.)
    if (m == 0u && !FINAL) return;
    uint32_t cnt = B.topcnt[slot];
    for (uint32_t e = lane; e < cnt; e += 64u) buf[e] = B.top[(size_t)slot * B.topcap + e];
    const uint64_t* __restrict__ lst = B.cand + (size_t)slot * B.fcap;
    for (uint32_t off = 0; off < m; off += 64u) {
        if (cnt + 64u > (uint32_t)CAP) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            hvs_wave_select_prune<CAP / 64>(buf, cnt, knn, lane, shist[w]);
            cnt = knn;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        const uint32_t take = (m - off) < 64u ? (m - off) : 64u;
        if (lane < take) buf[cnt + lane] = lst[off + lane];
        cnt += take;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (cnt > knn) {
        hvs_wave_select_prune<CAP / 64>(buf, cnt, knn, lane, shist[w]);
        cnt = knn;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    // largest distance held, as key bits (finite < +inf < NaN, the order of the keys)
    uint32_t dmax_bits = 0u;
    for (uint32_t e = lane; e < cnt; e += 64u) {
        const uint32_t bits = (uint32_t)(buf[e] >> 32);
        dmax_bits = bits > dmax_bits ? bits : dmax_bits;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)dmax_bits, o);
        dmax_bits = other > dmax_bits ? other : dmax_bits;
    }
    if constexpr (!FINAL) {
        for (uint32_t e = lane; e < cnt; e += 64u) B.top[(size_t)slot * B.topcap + e] = buf[e];
        // the m_next-th smallest distance held (the largest one when exactly m_next are held)
        uint32_t mth_bits = dmax_bits;
        const uint32_t mn = __builtin_amdgcn_readfirstlane(hvs_guess_m(L, G, level_next, B.ra[slot], B.rb[slot], knn));
        if (cnt > mn) mth_bits = (uint32_t)(hvs_wave_select_prune<CAP / 64, false>(buf, cnt, mn, lane, shist[w]) >> 32);
        if (lane == 0u) {
            B.topcnt[slot] = cnt;
            B.candcnt[slot] = 0;
            float tau = B.tau[slot];  // (+inf until the first cut)
            if (cnt >= mn) tau = fminf(tau, __uint_as_float(mth_bits));  // (a NaN key leaves tau as it is)
            const bool cut = tau < __builtin_inff();
            float theta = B.rb[slot] > B.ra[slot] ? -__builtin_inff() : __builtin_inff();
            if (HVS_IS_I8(fmt)) {
                // INT8 formats (see "INT8 filter" above): S = qq.dq + nh is exact, the band has no accumulation term
                int ti = B.rb[slot] > B.ra[slot] ? (int)0x80000000 : 0x7FFFFFFF;
                if (cut) {
                    const double g = 20.0 * 5.9604644775390625e-08;
                    const double iu = qz->inv_sd * qz->inv_sd;  // 1 / sd^2
                    // (normq: the clip term of a query outside the data's box, see hvs_k_prep)
                    const double band = ((double)B.nqb[slot] * (double)bounds->e_d8 + (double)B.eq[slot] * (double)bounds->n_d8 +
                                         (double)B.normq[slot]) * (1.0 + 1e-6);
                    // -2: one unit for nh = floor(.), one for the f64 evaluation of this expression (relative
                    // 1e-9 of the magnitudes on top)
                    double th = (0.5 * (B.qn[slot] * (1.0 - 1e-12) - (double)tau * (1.0 + 2.0 * g)) - band) * iu;
                    th -= 2.0 + 1e-9 * (B.qn[slot] + (double)tau + band) * iu;
                    th = floor(th);
                    ti = th >= 2147483647.0 ? 0x7FFFFFFF : (th <= -2147483647.0 ? (int)0x80000001 : (int)th);
                    if (!(th == th)) ti = (int)0x80000001;  // NaN cannot happen with finite inputs; keep everything
                }
                B.tau[slot] = tau;
                B.thetai[slot] = ti;
                return;
            }
            if (cut) {
                const double g = 20.0 * 5.9604644775390625e-08;
                const double sabs = (double)B.nqb[slot] * (double)bounds->nb_d + 1.02 * (double)bounds->hmax;
                const double mu = 256.0 * 5.9604644775390625e-08 * sabs;
                const double band = mu + (double)bounds->rho + (double)B.normq[slot] * (double)bounds->e_d +
                                    (double)B.eq[slot] * (double)bounds->nb_d;
                // slack for the f64 evaluation itself: relative to the magnitudes involved (an absolute constant
                // would swamp data whose distances are tiny, e.g. vectors scaled by 1e-3)
                const double slack = 1e-9 * (B.qn[slot] + (double)tau + band);
                const double th = 0.5 * (B.qn[slot] * (1.0 - 1e-12) - (double)tau * (1.0 + 2.0 * g)) - band * (1.0 + 1e-6) - slack;
                float tf = (float)th;
                if ((double)tf > th) {  // round down
                    if (tf == 0.0f) tf = -1.0e-30f;
                    else tf = __uint_as_float(__float_as_uint(tf) + (tf < 0.0f ? 1u : 0xFFFFFFFFu));
                }
                theta = tf;
            }
            B.tau[slot] = tau;
            B.theta[slot] = theta;
        }
        return;
    }
    // ---- final: verify.  Every row of the range with distance <= tau_last is held (see "Guessed thresholds"): the k
    // smallest keys held are the answer iff the k-th of them is <= tau_last, or nothing was ever discarded (tau = +inf).
    if (lane == 0u) {
        const uint32_t tau_bits = __float_as_uint(B.tau[slot]);
        const bool verified = tau_bits == 0x7F800000u || (cnt >= knn && dmax_bits <= tau_bits);
        if (!verified) hvs_flag_fail(B.fail_code, B.overflow, slot);
    }
    // ---- final: pad, rank-sort, write
    const float* __restrict__ qv = Q + (size_t)qi * HVS_QCOLS + 4;
    for (uint32_t base = cnt; base < knn; base += 64u) {
        const uint32_t e = base + lane;
        if (e < knn) {
            const uint32_t id = n - 1u - (e - cnt);
            const float* __restrict__ dv = D + (size_t)id * HVS_DCOLS + 2;
            // padding off (partial answers of a data shard): empty slots hold the largest key
            buf[e] = pad ? hvs_make_key(hvs_exact_dist(dv, qv), id) : ~0ull;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (uint32_t e = lane; e < knn; e += 64u) {
        const uint64_t ke = buf[e];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < knn; ++j) {
            const uint64_t kj = buf[j];
            rank += (kj < ke || (kj == ke && j < e)) ? 1u : 0u;
        }
        out_ids[(size_t)qi * knn + rank] = hvs_key_id(ke);
        if (out_dists) out_dists[(size_t)qi * knn + rank] = ke == ~0ull ? __builtin_inff() : hvs_key_dist(ke);
    }
}

// result rows of the queries in `list`, packed: dst[i][0..k) = src[list[i]][0..k)  (ids, or distances as bit patterns)
__global__ void hvs_k_gather_rows(const uint32_t* __restrict__ list, uint32_t count, const uint32_t* __restrict__ src, uint32_t k,
                                  uint32_t* __restrict__ dst)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count * k) return;
    const uint32_t i = e / k, j = e - i * k;
    dst[e] = src[(size_t)list[i] * k + j];
}

// planner probe: rows of D as type-0 queries (every `step`-th row from step / 2)
__global__ void hvs_k_probe_queries(const float* __restrict__ D, uint32_t n, uint32_t step, uint32_t count, float* __restrict__ Q)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count * HVS_QCOLS) return;
    const uint32_t i = e / HVS_QCOLS, col = e % HVS_QCOLS;
    const uint64_t row = ((uint64_t)i * step + step / 2u) % n;
    Q[e] = col == 0u ? 0.0f : (col < 4u ? -1.0f : D[row * HVS_DCOLS + (col - 2u)]);
}

// queries this batch did not answer -> appended to the call's lists: HVS_FAIL_EXACT for the exact engine, HVS_FAIL_RETRY
// for a filter batch with a proven last threshold (the counts run over all batches of a call)
__global__ void hvs_k_collect_overflow(HvsBatch B, uint32_t* __restrict__ list_exact, uint32_t* __restrict__ count_exact,
                                       uint32_t* __restrict__ list_retry, uint32_t* __restrict__ count_retry)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B.nslots) return;
    if (B.qid[s] == 0xFFFFFFFFu) return;
    const uint32_t code = B.overflow[s];
    if (code >= HVS_FAIL_EXACT)
        list_exact[atomicAdd(count_exact, 1u)] = B.qid[s];
    else if (code == HVS_FAIL_RETRY)
        list_retry[atomicAdd(count_retry, 1u)] = B.qid[s];
}
