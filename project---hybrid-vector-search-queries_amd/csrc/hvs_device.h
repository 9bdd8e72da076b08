// hvs_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// Everything here must reproduce the reference's float32 results bit for bit, so this
// translation unit is built with -ffp-contract=off (hipcc contracts a + b*b into v_fmac_f32
// by default; the reference is built without FMA, CMakeLists.txt:8).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define HVS_DCOLS 102
#define HVS_QCOLS 104
#define HVS_NDIM 100
#define HVS_KNN 100     // default k (the reference's KNN_LIMIT, optimized_impl.h:26); hvs_set_k changes it per context
#define HVS_KMAX 256    // largest k: lists of 64 NK keys with NK = 4 (k <= 128) or 8 (k <= 256)
#define HVS_WAVE 64

// ---------------------------------------------------------------------------------------------
// Query parameters: reference include/optimized_parallel.hpp:93-96
//   query_type = uint32(q[0]); v = int32(q[1]) (truncation); l = q[2]; r = q[3]
// `nodes[j][0] == v` compares the row's float with float(v).  uint32(t) truncates toward zero, so t in (-1, 0) is
// type 0, and int32(-2^31) is INT_MIN (both conversions are defined).  Types outside 0..3 (and values whose
// conversion is undefined behaviour in the reference) match no row: type 4.
// ---------------------------------------------------------------------------------------------
struct HvsQParams {
    uint32_t type;
    float vf, l, r;
};

__device__ __forceinline__ HvsQParams hvs_parse_query(const float* __restrict__ q)
{
    HvsQParams p;
    const float t = q[0];
    p.type = (t > -1.0f && t < 4.0f) ? (uint32_t)(int32_t)t : 4u;
    const float v = q[1];
    if (v >= -2147483648.0f && v < 2147483648.0f) {
        p.vf = (float)(int32_t)v;
    } else {
        p.vf = 0.0f;
        if (p.type == 1u || p.type == 3u) p.type = 4u;
    }
    p.l = q[2];
    p.r = q[3];
    return p;
}

// reference include/optimized_parallel.hpp:105-138
__device__ __forceinline__ bool hvs_row_passes(const HvsQParams& p, float C, float T)
{
    const bool ceq = (C == p.vf);
    const bool tin = (T >= p.l) && (T <= p.r);
    return p.type == 0u || (p.type == 1u && ceq) || (p.type == 2u && tin) || (p.type == 3u && ceq && tin);
}

// ---------------------------------------------------------------------------------------------
// Exact-order squared L2 (reference include/optimized_impl.h:96-125 + hsum :37-47):
// 8 strided accumulators, 12 full 8-wide steps over dims 0..95, masked tail adding dims
// 96..99 into accumulators 4..7 (accumulators 0..3 add +0.0f), reduction tree
// ((a0+a4)+(a1+a5)) + ((a2+a6)+(a3+a7)).  sub, mul, add are separate f32 roundings.
// `d` and `q` are indexable by a compile-time constant (register arrays or pointers).
// ---------------------------------------------------------------------------------------------
template <typename DV, typename QV>
__device__ __forceinline__ float hvs_exact_dist(const DV& d, const QV& q)
{
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int b = 0; b < 12; ++b) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = d[8 * b + j] - q[8 * b + j];
            t = t * t;
            acc[j] = acc[j] + t;
        }
    }
#pragma unroll
    for (int j = 4; j < 8; ++j) {
        float t = d[92 + j] - q[92 + j];
        t = t * t;
        acc[j] = acc[j] + t;
    }
    // accumulators 0..3 add +0.0f in the reference's masked step: x + 0.0f == x for x >= +0
    const float s0 = acc[0] + acc[4];
    const float s1 = acc[1] + acc[5];
    const float s2 = acc[2] + acc[6];
    const float s3 = acc[3] + acc[7];
    const float a = s0 + s1;
    const float b2 = s2 + s3;
    return a + b2;
}

// Packed form of the same arithmetic for gfx950's v_pk_add_f32 / v_pk_mul_f32 (two f32 lanes per
// VGPR pair, each lane an independent IEEE operation, so the bits are those of the scalar form).
// acc2[k] holds accumulators (2k, 2k+1); dims (8b+2k, 8b+2k+1) feed acc2[k]; the masked tail puts
// dims 96..99 into accumulators 4..7 = acc2[2], acc2[3]; the hsum tree
// ((a0+a4)+(a1+a5)) + ((a2+a6)+(a3+a7)) starts with two packed adds.
// `d2` / `q2` are indexable sequences of 50 float pairs (dims 2i, 2i+1).
typedef float hvs_f2 __attribute__((ext_vector_type(2)));

template <typename DV2, typename QV2>
__device__ __forceinline__ float hvs_exact_dist_pk(const DV2& d2, const QV2& q2)
{
    hvs_f2 acc2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc2[k] = hvs_f2{0.0f, 0.0f};
#pragma unroll
    for (int b = 0; b < 12; ++b) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hvs_f2 t = d2[4 * b + k] - q2[4 * b + k];
            t = t * t;
            acc2[k] = acc2[k] + t;
        }
    }
#pragma unroll
    for (int k = 2; k < 4; ++k) {
        hvs_f2 t = d2[46 + k] - q2[46 + k];
        t = t * t;
        acc2[k] = acc2[k] + t;
    }
    const hvs_f2 s01 = acc2[0] + acc2[2];  // (a0+a4, a1+a5)
    const hvs_f2 s23 = acc2[1] + acc2[3];  // (a2+a6, a3+a7)
    const float a = s01.x + s01.y;
    const float b2 = s23.x + s23.y;
    return a + b2;
}

// Sequential-order squared L2 of the reference's BASELINE engine (include/baseline.hpp:53-64,
// compare_with_id) and of the .dist side file (include/io.h:38-48): sum = sum + diff*diff for
// dims 0..99 in order.  A different f32 value than the SIMD order above (the reference documents
// the discrepancy, optimized.hpp:34-42 and src/fp_inaccuracy_test.cpp).
template <typename DV, typename QV>
__device__ __forceinline__ float hvs_scalar_order_dist(const DV& d, const QV& q)
{
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < HVS_NDIM; ++i) {
        float t = d[i] - q[i];
        t = t * t;
        sum = sum + t;
    }
    return sum;
}

// ---------------------------------------------------------------------------------------------
// Candidate keys: (distance bits << 32) | row id.  Distances are sums of squares (>= +0), so
// their IEEE bit patterns order like unsigned integers and ascending u64 order is exactly the
// canonical result order (dist asc, id asc) of SURVEY.md 8c.  Non-finite distances (inf/NaN components or
// overflow) get a defined place: +inf after every finite value, NaN (one canonical bit pattern) after +inf.  The
// reference admits any row while its list is not full (optimized_impl.h:301-304: add_new_vec = not_full || better)
// and then sorts with std::sort, whose result for NaN keys is unspecified -- this order is the build's own choice there.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t hvs_make_key(float dist, uint32_t id)
{
    const uint32_t bits = (dist != dist) ? 0x7FC00000u : __float_as_uint(dist);
    return ((uint64_t)bits << 32) | (uint64_t)id;
}
__device__ __forceinline__ float hvs_key_dist(uint64_t key) { return __uint_as_float((uint32_t)(key >> 32)); }
__device__ __forceinline__ uint32_t hvs_key_id(uint64_t key) { return (uint32_t)key; }

// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t hvs_prefix_count(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---------------------------------------------------------------------------------------------
// wave_select_prune: the wave-level counterpart of the reference's Knn container bookkeeping
// (include/optimized_impl.h:284-311 check_add/find_worst, :337-385 merge): instead of evicting
// the worst slot on every insertion, candidates are appended to a list and the list is cut back
// to its KEEP smallest keys when it fills up.  MSB-first radix select over the 64-bit keys
// (ballot + popcount per bit), then an in-place ordered compaction.  Keys must be distinct.
// All 64 lanes must call it (wave-uniform control flow).  `list` holds `cnt` <= 256 keys in
// memory visible to the whole wave.  Returns the KEEP-th smallest key; the list then holds
// exactly KEEP keys.  Precondition: cnt > KEEP.
// ---------------------------------------------------------------------------------------------
// Same contract, 8 bits per step instead of 1: a 256-bucket histogram of the undecided digit in wave-private LDS
// (`hist`, 256 x u32), one wave-wide prefix scan to find the bucket that holds the KEEP-th key, repeat inside that
// bucket.  Distances of one query's candidates separate within ~3 digits below their common prefix, against
// ~25 single-bit steps of 4 ballots each: the merge kernels spend 5-6x fewer instructions here.
// NK = keys per lane: the list holds at most 64 NK keys (256 for k <= 128, 512 for k <= 256); `keep` = k at run time.
// COMPACT = false: only the keep-th smallest key is returned, the list is left as it is.
template <int NK, bool COMPACT = true>
__device__ __forceinline__ uint64_t hvs_wave_select_prune(uint64_t* list, uint32_t cnt, uint32_t keep, uint32_t lane, uint32_t* hist)
{
    uint64_t k[NK];
    bool act[NK];
    bool valid[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const uint32_t idx = lane + 64u * i;
        act[i] = idx < cnt;
        valid[i] = act[i];
        k[i] = act[i] ? list[idx] : ~0ull;
    }
    // common prefix of all keys (see the bit-serial version)
    int hi;
    uint64_t prefix;
    {
        const uint64_t k0 = list[0];
        uint64_t d = 0;
#pragma unroll
        for (int i = 0; i < NK; ++i) d |= act[i] ? (k[i] ^ k0) : 0ull;
        uint32_t dlo = (uint32_t)d, dhi = (uint32_t)(d >> 32);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            dlo |= (uint32_t)__shfl_xor((int)dlo, o);
            dhi |= (uint32_t)__shfl_xor((int)dhi, o);
        }
        d = ((uint64_t)__builtin_amdgcn_readfirstlane(dhi) << 32) | __builtin_amdgcn_readfirstlane(dlo);
        hi = d ? 63 - (int)__builtin_clzll(d) : 0;
        prefix = hi < 63 ? (k0 & (~0ull << (hi + 1))) : 0ull;
    }
    uint32_t r = keep;  // rank of the wanted key inside the active set
    for (;;) {          // wave-uniform
        const int lo = hi >= 7 ? hi - 7 : 0;
        const uint32_t dmask = (1u << (hi - lo + 1)) - 1u;
#pragma unroll
        for (int i = 0; i < 4; ++i) hist[lane + 64u * i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
        for (int i = 0; i < NK; ++i)
            if (act[i]) atomicAdd(&hist[(uint32_t)(k[i] >> lo) & dmask], 1u);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // lane l owns buckets 4l .. 4l+3
        const uint32_t c0 = hist[4u * lane], c1 = hist[4u * lane + 1u], c2 = hist[4u * lane + 2u], c3 = hist[4u * lane + 3u];
        const uint32_t mine = c0 + c1 + c2 + c3;
        uint32_t incl = mine;  // inclusive scan over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
            incl += lane >= (uint32_t)o ? up : 0u;
        }
        const uint32_t b0 = incl - mine, b1 = b0 + c0, b2 = b1 + c1, b3 = b2 + c2;  // keys in buckets before mine
        // exactly one bucket has  before < r <= before + count
        const bool h0 = b0 < r && r <= b0 + c0, h1 = b1 < r && r <= b1 + c1, h2 = b2 < r && r <= b2 + c2,
                   h3 = b3 < r && r <= b3 + c3;
        const uint32_t jd = h0 ? 0u : h1 ? 1u : h2 ? 2u : 3u;
        const uint32_t jb = h0 ? b0 : h1 ? b1 : h2 ? b2 : b3;
        const uint32_t jc = h0 ? c0 : h1 ? c1 : h2 ? c2 : c3;
        const uint64_t hm = __ballot(h0 | h1 | h2 | h3);
        const int src = (int)__builtin_ctzll(hm | (1ull << 63));
        const uint32_t digit = 4u * (uint32_t)src + __builtin_amdgcn_readlane(jd, src);
        const uint32_t before = __builtin_amdgcn_readlane(jb, src);
        const uint32_t rem = __builtin_amdgcn_readlane(jc, src);
        r -= before;
        prefix |= (uint64_t)digit << lo;
#pragma unroll
        for (int i = 0; i < NK; ++i) act[i] = act[i] && (((uint32_t)(k[i] >> lo) & dmask) == digit);
        if (rem == 1u) {
            // one key left under this prefix: it is the answer (keys are distinct)
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const uint64_t mm = __ballot(act[i]);
                if (mm != 0ull) {
                    const int s2 = (int)__builtin_ctzll(mm);
                    const uint32_t klo = __builtin_amdgcn_readlane((uint32_t)k[i], s2);
                    const uint32_t khi = __builtin_amdgcn_readlane((uint32_t)(k[i] >> 32), s2);
                    prefix = ((uint64_t)khi << 32) | klo;
                }
            }
            break;
        }
        if (lo == 0) break;  // every bit decided: prefix is the key
        hi = lo - 1;
    }
    if constexpr (!COMPACT) return prefix;
    uint32_t base = 0;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const bool kp = valid[i] && k[i] <= prefix;
        const uint64_t mask = __ballot(kp);
        if (kp) list[base + hvs_prefix_count(mask)] = k[i];
        base += (uint32_t)__popcll(mask);
    }
    return prefix;
}

template <int NK>
__device__ __forceinline__ uint64_t hvs_wave_select_prune(uint64_t* list, uint32_t cnt, uint32_t keep, uint32_t lane)
{
    uint64_t k[NK];
    bool valid[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const uint32_t idx = lane + 64u * i;
        valid[i] = idx < cnt;
        k[i] = valid[i] ? list[idx] : ~0ull;
    }
    uint64_t prefix = 0;
    uint32_t r = keep;
    uint32_t rem = cnt;  // keys that still match the decided prefix bits
    // Bits above the highest bit in which any two keys differ are common to all keys (distances of one query's
    // candidates share sign, exponent and often the first mantissa bits): they go into the prefix without a
    // counting step each.
    int top = 63;
    {
        const uint64_t k0 = list[0];  // cnt > KEEP >= 1: a valid key
        uint64_t d = 0;
#pragma unroll
        for (int i = 0; i < NK; ++i) d |= valid[i] ? (k[i] ^ k0) : 0ull;
        uint32_t dlo = (uint32_t)d, dhi = (uint32_t)(d >> 32);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            dlo |= (uint32_t)__shfl_xor((int)dlo, o);
            dhi |= (uint32_t)__shfl_xor((int)dhi, o);
        }
        d = ((uint64_t)dhi << 32) | dlo;
        d = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(d >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)d);
        top = d ? 63 - (int)__builtin_clzll(d) : 0;  // (d == 0 cannot happen: keys are distinct and cnt > 1)
        prefix = top < 63 ? (k0 & (~0ull << (top + 1))) : 0ull;
    }
    for (int bit = top; bit >= 0; --bit) {
        const uint64_t himask = (bit == 63) ? 0ull : (~0ull << (bit + 1));
        uint32_t c0 = 0;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const bool m = valid[i] && (((k[i] ^ prefix) & himask) == 0ull) && (((k[i] >> bit) & 1ull) == 0ull);
            c0 += (uint32_t)__popcll(__ballot(m));
        }
        if (r > c0) {
            r -= c0;
            rem -= c0;
            prefix |= (1ull << bit);
        } else {
            rem = c0;
        }
        if (rem == 1u) {
            // one key left under this prefix: it is the answer (keys are distinct), no need to walk
            // the remaining bits (distances usually separate within the first ~25 bits)
            const uint64_t lomask = ~0ull << bit;
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const uint64_t mm = __ballot(valid[i] && (((k[i] ^ prefix) & lomask) == 0ull));
                if (mm != 0ull) {
                    const int src = (int)__builtin_ctzll(mm);
                    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)k[i], src);
                    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(k[i] >> 32), src);
                    prefix = ((uint64_t)hi << 32) | lo;
                }
            }
            break;
        }
    }
    uint32_t base = 0;
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const bool kp = valid[i] && k[i] <= prefix;
        const uint64_t mask = __ballot(kp);
        if (kp) list[base + hvs_prefix_count(mask)] = k[i];
        base += (uint32_t)__popcll(mask);
    }
    return prefix;
}
