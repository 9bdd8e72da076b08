// hvs_kernels.h -- gfx950 kernels of the exact engine (FP32 exact-order scan + top-100 select).
#pragma once

#include "hvs_device.h"
#include "../../include/hvs_gen.h"

// per (query, row-chunk) candidate list capacity (keys): kernels are instantiated for CAP = 256 (k <= 128) and
// CAP = 512 (k <= 256); k itself is a run-time argument

// ---------------------------------------------------------------------------------------------
// Synthetic inputs generated in HBM (include/hvs_gen.h), one thread per element.
// ---------------------------------------------------------------------------------------------
__global__ void hvs_k_gen_data(float* __restrict__ out, uint64_t nelem, uint64_t seed, int profile, uint32_t ncat)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nelem; e += stride) {
        const uint64_t row = e / HVS_DCOLS;
        const uint32_t col = (uint32_t)(e - row * HVS_DCOLS);
        out[e] = hvs_gen_data_elem(seed, profile, ncat, row, col);
    }
}

__global__ void hvs_k_gen_queries(float* __restrict__ out, uint64_t nelem, uint64_t seed, int profile, uint32_t ncat,
                                  int force_type, uint64_t first_row)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nelem; e += stride) {
        const uint64_t row = e / HVS_QCOLS;
        const uint32_t col = (uint32_t)(e - row * HVS_QCOLS);
        out[e] = hvs_gen_query_elem(seed, profile, ncat, force_type, first_row + row, col);
    }
}

// ---------------------------------------------------------------------------------------------
// Query scheduling keys.  Queries are independent (optimized_parallel.hpp:91 carries no state
// between iterations), so the engine may answer them in any order: they are grouped by
// (type, v, l) so that the 64 queries of a wavefront share one predicate shape and a row that
// no lane wants is skipped by the whole wave.
// key = type:3 | float(v) as ordered u32:32 | top 29 bits of ordered(l)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hvs_ordered_u32(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void hvs_k_query_keys(const float* __restrict__ Q, uint32_t q0, uint32_t nq, uint64_t* __restrict__ keys,
                                 uint32_t* __restrict__ idx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const HvsQParams p = hvs_parse_query(Q + (size_t)(q0 + i) * HVS_QCOLS);
    const uint32_t vk = (p.type == 1u || p.type == 3u) ? hvs_ordered_u32(p.vf) : 0u;
    const uint32_t lk = (p.type == 2u || p.type == 3u) ? hvs_ordered_u32(p.l) : 0u;
    keys[i] = ((uint64_t)p.type << 61) | ((uint64_t)vk << 29) | (uint64_t)(lk >> 3);
    idx[i] = q0 + i;
}

// ---------------------------------------------------------------------------------------------
// hvs_k_scan_exact -- the reference's inner hot loop (optimized_parallel.hpp:100-139 +
// optimized_impl.h:54-125,284-311) re-shaped for a 64-wide wavefront:
//
//   * one LANE per QUERY: the query's 100 dims live in 100 VGPRs for the whole kernel, the
//     running threshold tau and the list fill are per-lane registers;
//   * the DATA ROW is wave-uniform: it is fetched through the scalar cache into SGPRs and fed
//     to v_sub_f32 as the scalar operand, so one 408-byte row fetch serves 64 (query,row) pairs
//     and costs no VGPRs, no LDS and no vector-memory instruction;
//   * predicates (C == v, l <= T <= r) are evaluated per lane on the scalar C,T; a row no lane
//     accepts is skipped by the whole wave (one ballot + scalar branch);
//   * the distance is the reference's exact order (8 accumulators, no FMA), 300 VALU ops/pair;
//   * admission is the reference's strict `dist < worst` (optimized_impl.h:301): accepted pairs
//     are appended to the lane's private list in global memory; a full list is cut back to its
//     100 smallest (dist,id) keys by the whole wave (hvs_wave_select_prune), which also gives
//     the new tau.  Rows are scanned in ascending id, so a later row with dist == tau can never
//     displace a kept one under the canonical (dist asc, id asc) order.
//
// Grid: x = blocks of 4 query-waves (256 queries), y = row chunk.  Waves never synchronise
// with each other.  Output: per (chunk, query slot) a list of <= HVS_CAND_CAP keys + its fill.
// ---------------------------------------------------------------------------------------------
struct HvsUniformRow2 {
    const hvs_f2* __restrict__ p;  // row + 2 floats: 8-byte aligned (row stride 408 B)
    __device__ __forceinline__ hvs_f2 operator[](int i) const { return p[i]; }
};

struct HvsUniformRow1 {
    const float* __restrict__ p;
    __device__ __forceinline__ float operator[](int i) const { return p[i]; }
};
struct HvsPairAsScalar {  // view the lane's 50 query pairs as 100 floats
    const hvs_f2* q2;
    __device__ __forceinline__ float operator[](int i) const { return (i & 1) ? q2[i >> 1].y : q2[i >> 1].x; }
};

// SCALAR_ORDER = false: the hot path's SIMD summation order (optimized_impl.h:96-125);
// SCALAR_ORDER = true : the baseline engine's sequential order (baseline.hpp:53-64), BASELINE.json configs[0].
// (three workgroups per CU: at four the 128-register budget spilled 10-22 registers of the 100 query components)
template <bool SCALAR_ORDER, int CAP>
__global__ __launch_bounds__(256, 3) void hvs_k_scan_exact(
    const float* __restrict__ D, const float* __restrict__ Q, const uint32_t* __restrict__ qorder, uint32_t nq,
    uint32_t nq_pad, uint32_t sn, uint32_t rows_per_chunk, uint64_t* __restrict__ cand, uint32_t* __restrict__ cand_cnt,
    unsigned long long* __restrict__ counters, uint32_t knn)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t qwave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t slot = qwave * 64u + lane;
    const uint32_t chunk = blockIdx.y;
    if (qwave * 64u >= nq) return;  // wave-uniform

    const bool have_q = slot < nq;
    const uint32_t qi = qorder[have_q ? slot : nq - 1u];
    const float* __restrict__ qrow = Q + (size_t)qi * HVS_QCOLS;
    HvsQParams p = hvs_parse_query(qrow);
    if (!have_q) p.type = 4u;

    hvs_f2 q2[HVS_NDIM / 2];
#pragma unroll
    for (int i = 0; i < HVS_NDIM / 4; ++i) {
        const float4 v4 = *reinterpret_cast<const float4*>(qrow + 4 + 4 * i);
        q2[2 * i] = hvs_f2{v4.x, v4.y};
        q2[2 * i + 1] = hvs_f2{v4.z, v4.w};
    }

    const uint32_t r0 = chunk * rows_per_chunk;
    uint32_t r1 = r0 + rows_per_chunk;
    if (r1 > sn || r1 < r0) r1 = sn;

    uint64_t* __restrict__ mylist = cand + ((size_t)chunk * nq_pad + slot) * CAP;
    // tau = NaN until the list has been cut back once: `!(dist >= tau)` then admits EVERY passing row, +inf and NaN
    // distances included, as the reference does while its list is not full (optimized_impl.h:301-304)
    float tau = __builtin_nanf("");
    uint32_t cnt = 0;
    uint32_t npass = 0, nscan = 0;  // per-wave statistics (uniform)

    for (uint32_t j = r0; j < r1; ++j) {
        const float* __restrict__ row = D + (size_t)j * HVS_DCOLS;
        const float C = row[0];
        const float T = row[1];
        const bool pass = hvs_row_passes(p, C, T);
        const uint64_t pmask = __ballot(pass);
        if (pmask == 0ull) continue;
        npass += (uint32_t)__popcll(pmask);
        nscan += 64u;

        HvsUniformRow2 dv{reinterpret_cast<const hvs_f2*>(row + 2)};
        float dist;
        if (SCALAR_ORDER) {
            HvsUniformRow1 d1{row + 2};
            HvsPairAsScalar q1{q2};
            dist = hvs_scalar_order_dist(d1, q1);
        } else {
            dist = hvs_exact_dist_pk(dv, q2);
        }

        if (pass && !(dist >= tau)) {
            mylist[cnt] = hvs_make_key(dist, j);
            ++cnt;
        }
        uint64_t full = __ballot(cnt == (uint32_t)CAP);
        if (full != 0ull) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            while (full != 0ull) {
                const uint32_t l = (uint32_t)__builtin_ctzll(full);
                full &= full - 1ull;
                uint64_t* lst = cand + ((size_t)chunk * nq_pad + (qwave * 64u + l)) * CAP;
                const uint64_t kth = hvs_wave_select_prune<CAP / 64>(lst, (uint32_t)CAP, knn, lane);
                if (lane == l) {
                    cnt = knn;
                    tau = hvs_key_dist(kth);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
    }
    if (have_q) cand_cnt[(size_t)chunk * nq_pad + slot] = cnt;
    if (lane == 0u) {
        atomicAdd(&counters[0], (unsigned long long)npass);
        atomicAdd(&counters[1], (unsigned long long)nscan);
    }
}

// ---------------------------------------------------------------------------------------------
// hvs_k_scan_exact_lds -- the same scan with the data rows staged in LDS instead of SGPRs.
// The four waves of a workgroup (256 queries) walk the same row chunk, so a block of 32 rows is
// copied once per workgroup (coalesced 16-B loads, double-buffered, one barrier per 32 rows) into a
// 16-B aligned image [x0..x99, C, T, pad, pad] and every wave reads it back with broadcast
// ds_read_b128.  This removes the scalar-load round trips of hvs_k_scan_exact from the row loop
// (two dependent s_load + s_waitcnt per row there); arithmetic and admission are identical.
// ---------------------------------------------------------------------------------------------
#ifndef HVS_LDS_ROWS
#define HVS_LDS_ROWS 16   // rows per staged block (32: 8 more staging registers, which spill at three workgroups per CU)
#endif
#define HVS_LDS_ROW_F 104  // floats per staged row (416 B, 16-B aligned)

struct HvsLdsRow2 {
    const float4* p;  // 16-B aligned row image in LDS
    __device__ __forceinline__ hvs_f2 operator[](int i) const
    {
        const float4 v = p[i >> 1];
        return (i & 1) ? hvs_f2{v.z, v.w} : hvs_f2{v.x, v.y};
    }
};
// Exact-order distance of an LDS-staged row, software-pipelined by hand: the row's 25 ds_read_b128 are issued two b-steps
// (4 reads, 16 components) ahead of the packed arithmetic that consumes them, alternating between two register sets, so
// that every group of 8 independent v_pk_add / v_pk_mul / v_pk_add triples runs while the next group's reads are in
// flight.  (Left to the compiler the loop kept 2 reads in flight and waited for each: ~35 s_nop and 27 s_waitcnt per row.)
// The arithmetic and its order per accumulator are hvs_exact_dist_pk's: acc2[k] takes dims (8b + 2k, 8b + 2k + 1) for
// b = 0..11 in order, then the masked tail, then the hsum tree (optimized_impl.h:96-125, :37-47).
// (ceiling experiments, never shipped: HVS_EXPERIMENT_EXACT=1 reads every other step's row data from LDS and re-uses the registers
// for the steps between -- half the ds_read_b128 traffic, wrong distances; =2 drops the multiplications -- two thirds of the vector
// arithmetic.  profiles/r04/exact_engine_ceiling.txt)
#if defined(HVS_EXPERIMENT_EXACT) && HVS_EXPERIMENT_EXACT == 2
#define HVS_LDS_SQ(t) t
#else
#define HVS_LDS_SQ(t) t * t
#endif
#define HVS_LDS_STEP(LO, HI, B)                                                                     \
    {                                                                                               \
        hvs_f2 t0 = hvs_f2{LO.x, LO.y} - q2[4 * (B) + 0], t1 = hvs_f2{LO.z, LO.w} - q2[4 * (B) + 1]; \
        hvs_f2 t2 = hvs_f2{HI.x, HI.y} - q2[4 * (B) + 2], t3 = hvs_f2{HI.z, HI.w} - q2[4 * (B) + 3]; \
        t0 = HVS_LDS_SQ(t0);                                                                        \
        t1 = HVS_LDS_SQ(t1);                                                                        \
        t2 = HVS_LDS_SQ(t2);                                                                        \
        t3 = HVS_LDS_SQ(t3);                                                                        \
        a0 = a0 + t0;                                                                               \
        a1 = a1 + t1;                                                                               \
        a2 = a2 + t2;                                                                               \
        a3 = a3 + t3;                                                                               \
    }
// one group: two b-steps from registers (X0..X3), then the reads that refill them (float4 index NEXT.., or none)
#define HVS_LDS_GROUP(X0, X1, X2, X3, B, NEXT)          \
    HVS_LDS_STEP(X0, X1, (B))                           \
    HVS_LDS_STEP(X2, X3, (B) + 1)                       \
    __builtin_amdgcn_sched_barrier(0);                  \
    if ((NEXT) >= 0 && (NEXT) + 3 < 24) {               \
        X0 = rowp[(NEXT) >= 0 ? (NEXT) : 0];            \
        X1 = rowp[(NEXT) >= 0 ? (NEXT) + 1 : 0];        \
        X2 = rowp[(NEXT) >= 0 ? (NEXT) + 2 : 0];        \
        X3 = rowp[(NEXT) >= 0 ? (NEXT) + 3 : 0];        \
    }                                                   \
    __builtin_amdgcn_sched_barrier(0);
#ifndef HVS_LDS_RING3
#define HVS_LDS_RING3 1   // three single-step register sets (6 reads in flight, 24 registers); 0: two double-step sets (8 reads, 32
                          // registers: needs two workgroups per CU -- measured 0.216 against 0.237 of the FP32 peak on type-0)
#endif
__device__ __forceinline__ float hvs_exact_dist_pk_lds(const float4* rowp, const hvs_f2* q2)
{
    hvs_f2 a0 = hvs_f2{0.0f, 0.0f}, a1 = a0, a2 = a0, a3 = a0;
#if HVS_LDS_RING3
    float4 A0 = rowp[0], A1 = rowp[1], B0 = rowp[2], B1 = rowp[3], C0 = rowp[4], C1 = rowp[5];
    __builtin_amdgcn_sched_barrier(0);
#if defined(HVS_EXPERIMENT_EXACT) && HVS_EXPERIMENT_EXACT == 1
#define HVS_LDS_REFILL(NEXT) ((NEXT) >= 0 && (((NEXT) / 2) & 1) == 0)
#else
#define HVS_LDS_REFILL(NEXT) ((NEXT) >= 0)
#endif
#define HVS_LDS_ONE(X0, X1, B, NEXT)                    \
    HVS_LDS_STEP(X0, X1, (B))                           \
    __builtin_amdgcn_sched_barrier(0);                  \
    if (HVS_LDS_REFILL(NEXT)) {                         \
        X0 = rowp[(NEXT) >= 0 ? (NEXT) : 0];            \
        if ((NEXT) + 1 < 25) X1 = rowp[(NEXT) >= 0 ? (NEXT) + 1 : 0]; \
    }                                                   \
    __builtin_amdgcn_sched_barrier(0);
    HVS_LDS_ONE(A0, A1, 0, 6)
    HVS_LDS_ONE(B0, B1, 1, 8)
    HVS_LDS_ONE(C0, C1, 2, 10)
    HVS_LDS_ONE(A0, A1, 3, 12)
    HVS_LDS_ONE(B0, B1, 4, 14)
    HVS_LDS_ONE(C0, C1, 5, 16)
    HVS_LDS_ONE(A0, A1, 6, 18)
    HVS_LDS_ONE(B0, B1, 7, 20)
    HVS_LDS_ONE(C0, C1, 8, 22)
    HVS_LDS_ONE(A0, A1, 9, 24)     // A0 <- dims 96..99 (the masked tail); A1 unused
    HVS_LDS_ONE(B0, B1, 10, -1)
    HVS_LDS_ONE(C0, C1, 11, -1)
    const float4 TL = A0;
#undef HVS_LDS_ONE
#else
    float4 A0 = rowp[0], A1 = rowp[1], A2 = rowp[2], A3 = rowp[3];
    float4 B0 = rowp[4], B1 = rowp[5], B2 = rowp[6], B3 = rowp[7];
    __builtin_amdgcn_sched_barrier(0);
    HVS_LDS_GROUP(A0, A1, A2, A3, 0, 8)
    HVS_LDS_GROUP(B0, B1, B2, B3, 2, 12)
    HVS_LDS_GROUP(A0, A1, A2, A3, 4, 16)
    HVS_LDS_GROUP(B0, B1, B2, B3, 6, 20)
    float4 TL;
    HVS_LDS_STEP(A0, A1, 8)
    HVS_LDS_STEP(A2, A3, 9)
    __builtin_amdgcn_sched_barrier(0);
    TL = rowp[24];  // dims 96..99: the masked tail
    __builtin_amdgcn_sched_barrier(0);
    HVS_LDS_STEP(B0, B1, 10)
    HVS_LDS_STEP(B2, B3, 11)
#endif
    {
        hvs_f2 t2 = hvs_f2{TL.x, TL.y} - q2[48];
        hvs_f2 t3 = hvs_f2{TL.z, TL.w} - q2[49];
        t2 = t2 * t2;
        t3 = t3 * t3;
        a2 = a2 + t2;
        a3 = a3 + t3;
    }
    const hvs_f2 s01 = a0 + a2;  // (a0+a4, a1+a5)
    const hvs_f2 s23 = a1 + a3;  // (a2+a6, a3+a7)
    const float a = s01.x + s01.y;
    const float b2 = s23.x + s23.y;
    return a + b2;
}
#undef HVS_LDS_GROUP
#undef HVS_LDS_STEP

struct HvsLdsRow1 {
    const float* p;
    __device__ __forceinline__ float operator[](int i) const { return p[i]; }
};

// (three workgroups per CU = three waves per SIMD, 168 registers: the 100 query components, three register sets of row data
// in flight (hvs_exact_dist_pk_lds) and the staging registers of a 16-row block fit without spilling)
#ifndef HVS_LDS_SCAN_WGS
#define HVS_LDS_SCAN_WGS 3
#endif
template <bool SCALAR_ORDER, int CAP>
__global__ __launch_bounds__(256, HVS_LDS_SCAN_WGS) void hvs_k_scan_exact_lds(
    const float* __restrict__ D, const float* __restrict__ Q, const uint32_t* __restrict__ qorder, uint32_t nq,
    uint32_t nq_pad, uint32_t sn, uint32_t rows_per_chunk, uint64_t* __restrict__ cand, uint32_t* __restrict__ cand_cnt,
    unsigned long long* __restrict__ counters, uint32_t knn)
{
    __shared__ float4 srow[2][HVS_LDS_ROWS * HVS_LDS_ROW_F / 4];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t qwave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t slot = qwave * 64u + lane;
    const uint32_t chunk = blockIdx.y;
    const bool wave_active = qwave * 64u < nq;  // inactive waves still help staging and meet the barriers

    const bool have_q = slot < nq;
    const uint32_t qi = qorder[have_q ? slot : (nq - 1u)];
    const float* __restrict__ qrow = Q + (size_t)qi * HVS_QCOLS;
    HvsQParams p = hvs_parse_query(qrow);
    if (!have_q) p.type = 4u;
    hvs_f2 q2[HVS_NDIM / 2];
#pragma unroll
    for (int i = 0; i < HVS_NDIM / 4; ++i) {
        const float4 v4 = *reinterpret_cast<const float4*>(qrow + 4 + 4 * i);
        q2[2 * i] = hvs_f2{v4.x, v4.y};
        q2[2 * i + 1] = hvs_f2{v4.z, v4.w};
    }

    const uint32_t r0 = chunk * rows_per_chunk;
    uint32_t r1 = r0 + rows_per_chunk;
    if (r1 > sn || r1 < r0) r1 = sn;
    if (r0 >= r1) return;  // uniform over the workgroup

    // staging: item e = (row r, piece c): c < 25 -> floats 2+4c..5+4c of the row, c == 25 -> (C, T, 0, 0)
    constexpr uint32_t kItems = HVS_LDS_ROWS * 26u;
    float4 stg[(kItems + 255u) / 256u];
    auto load_block = [&](uint32_t j0) {
#pragma unroll
        for (uint32_t k = 0; k < (kItems + 255u) / 256u; ++k) {
            const uint32_t e = threadIdx.x + 256u * k;
            const uint32_t r = e / 26u, c = e % 26u;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < kItems && j0 + r < r1) {
                const float* __restrict__ src = D + (size_t)(j0 + r) * HVS_DCOLS;
                if (c < 25u) {
                    const float2 a = *reinterpret_cast<const float2*>(src + 2 + 4 * c);
                    const float2 b = *reinterpret_cast<const float2*>(src + 4 + 4 * c);
                    v = make_float4(a.x, a.y, b.x, b.y);
                } else {
                    const float2 a = *reinterpret_cast<const float2*>(src);
                    v = make_float4(a.x, a.y, 0.f, 0.f);
                }
            }
            stg[k] = v;
        }
    };
    auto store_block = [&](uint32_t buf) {
#pragma unroll
        for (uint32_t k = 0; k < (kItems + 255u) / 256u; ++k) {
            const uint32_t e = threadIdx.x + 256u * k;
            if (e < kItems) srow[buf][(e / 26u) * (HVS_LDS_ROW_F / 4) + (e % 26u)] = stg[k];
        }
    };

    uint64_t* __restrict__ mylist = cand + ((size_t)chunk * nq_pad + slot) * CAP;
    float tau = __builtin_nanf("");  // (see hvs_k_scan_exact)
    uint32_t cnt = 0;
    uint32_t npass = 0, nscan = 0;
    // queries are sorted by type: a wave of type-0 queries only (the exact engine's slowest class) skips the per-row predicate --
    // one LDS read with its wait at the head of every row, three compares and eight scalar mask operations
    const bool wave_all0 = __ballot(have_q && p.type != 0u) == 0ull;
    const uint32_t nhave = (uint32_t)__popcll(__ballot(have_q));

    load_block(r0);
    store_block(0u);
    __syncthreads();
    uint32_t buf = 0;
    for (uint32_t j0 = r0; j0 < r1; j0 += HVS_LDS_ROWS) {
        const bool more = j0 + HVS_LDS_ROWS < r1;
        if (more) load_block(j0 + HVS_LDS_ROWS);
        if (wave_active) {
            const uint32_t nrow = (r1 - j0) < HVS_LDS_ROWS ? (r1 - j0) : HVS_LDS_ROWS;
            for (uint32_t r = 0; r < nrow; ++r) {
                const float4* rowp = &srow[buf][r * (HVS_LDS_ROW_F / 4)];
                bool pass;
                if (wave_all0) {  // (wave-uniform) a wave of pure k-NN queries takes every row: no attribute read, no predicate
                    pass = have_q;
                    npass += nhave;
                } else {
                    const float4 attr = rowp[25];
                    pass = hvs_row_passes(p, attr.x, attr.y);
                    const uint64_t pmask = __ballot(pass);
                    if (pmask == 0ull) continue;
                    npass += (uint32_t)__popcll(pmask);
                }
                nscan += 64u;
                float dist;
                if (SCALAR_ORDER) {
                    HvsLdsRow1 d1{reinterpret_cast<const float*>(rowp)};
                    HvsPairAsScalar q1{q2};
                    dist = hvs_scalar_order_dist(d1, q1);
                } else {
                    // packed f32 math: measured 14.0 k type-0 queries/s at D=1e7 against 12.5 k with the 300 unpacked ops
                    dist = hvs_exact_dist_pk_lds(rowp, q2);
                }
                if (pass && !(dist >= tau)) {
                    mylist[cnt] = hvs_make_key(dist, j0 + r);
                    ++cnt;
                }
                uint64_t full = __ballot(cnt == (uint32_t)CAP);
                if (full != 0ull) {
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    while (full != 0ull) {
                        const uint32_t l = (uint32_t)__builtin_ctzll(full);
                        full &= full - 1ull;
                        uint64_t* lst = cand + ((size_t)chunk * nq_pad + (qwave * 64u + l)) * CAP;
                        const uint64_t kth = hvs_wave_select_prune<CAP / 64>(lst, (uint32_t)CAP, knn, lane);
                        if (lane == l) {
                            cnt = knn;
                            tau = hvs_key_dist(kth);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                }
            }
        }
        if (more) store_block(buf ^ 1u);
        __syncthreads();
        buf ^= 1u;
    }
    if (have_q) cand_cnt[(size_t)chunk * nq_pad + slot] = cnt;
    if (wave_active && lane == 0u) {
        atomicAdd(&counters[0], (unsigned long long)npass);
        atomicAdd(&counters[1], (unsigned long long)nscan);
    }
}

// ---------------------------------------------------------------------------------------------
// hvs_k_select -- per query: merge the per-chunk candidate lists (the counterpart of
// Knn::merge, optimized_impl.h:337-385 + optimized_parallel.hpp:142-146), pad with the last
// rows of D when fewer than 100 rows matched (optimized_parallel.hpp:149-157: rows n-1, n-2,
// ... regardless of predicate or duplicates, distances by the same exact-order kernel) and
// emit ids in ascending (dist, id) order (get_knn_sorted, optimized_impl.h:392-415).
// One wave per query, 4 queries per 256-thread block, a 256-key LDS buffer per wave.
// ---------------------------------------------------------------------------------------------
template <bool SCALAR_ORDER, int CAP>
__global__ __launch_bounds__(256) void hvs_k_select(
    const float* __restrict__ D, uint32_t n, const float* __restrict__ Q, const uint32_t* __restrict__ qorder,
    uint32_t nq, uint32_t nq_pad, uint32_t nchunks, const uint64_t* __restrict__ cand,
    const uint32_t* __restrict__ cand_cnt, int pad, uint32_t* __restrict__ out_ids, float* __restrict__ out_dists, uint32_t knn)
{
    __shared__ uint64_t sbuf[4][CAP];
    __shared__ uint32_t shist[4][256];  // digit histograms of the radix select
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t slot = blockIdx.x * 4u + w;
    if (slot >= nq) return;  // wave-uniform
    uint64_t* buf = sbuf[w];
    const uint32_t qi = qorder[slot];
    if (qi == 0xFFFFFFFFu) return;  // padding slot of the range-scan layout

    uint32_t cnt = 0;
    for (uint32_t c = 0; c < nchunks; ++c) {
        const uint32_t m = cand_cnt[(size_t)c * nq_pad + slot];
        const uint64_t* __restrict__ lst = cand + ((size_t)c * nq_pad + slot) * CAP;
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
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (cnt > knn) {
        hvs_wave_select_prune<CAP / 64>(buf, cnt, knn, lane, shist[w]);
        cnt = knn;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    // padding: fewer than 100 matching rows in [0,sn)
    const float* __restrict__ qv = Q + (size_t)qi * HVS_QCOLS + 4;
    for (uint32_t base = cnt; base < knn; base += 64u) {
        const uint32_t e = base + lane;
        if (e < knn) {
            const uint32_t id = n - 1u - (e - cnt);
            const float* __restrict__ dv = D + (size_t)id * HVS_DCOLS + 2;
            buf[e] = pad ? hvs_make_key(SCALAR_ORDER ? hvs_scalar_order_dist(dv, qv) : hvs_exact_dist(dv, qv), id) : ~0ull;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // rank sort of exactly 100 keys (duplicates possible after padding: ties broken by slot)
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

// ---------------------------------------------------------------------------------------------
// hvs_k_merge_shards -- D-sharded mode: the multi-GPU counterpart of Knn::merge
// (optimized_impl.h:337-385).  Every shard (GPU) answered ALL queries on its own rows with padding off:
// ids are shard-local (0xFFFFFFFF = empty slot), lists sorted by (dist, id).  Per query (one wave): the
// nshards x 100 keys (dist bits << 32 | global id) are reduced to the 100 smallest; when fewer than 100 rows
// matched anywhere, rows n_total-1, n_total-2, ... are appended with their exact-order distances
// (optimized_parallel.hpp:149-157; `pad_dists[q][s]` = distance of query q to row n_total-1-s), and the 100
// keys leave in ascending (dist, id) order.  ids_all / dists_all: [nshards][nq][100] as all_gather lays them out.
// ---------------------------------------------------------------------------------------------
struct HvsShardRows {
    uint64_t row0[16];  // first global row of each shard
};

template <int CAP>
__global__ __launch_bounds__(256) void hvs_k_merge_shards(const uint32_t* __restrict__ ids_all,
                                                          const float* __restrict__ dists_all, uint32_t nshards,
                                                          uint32_t nq, HvsShardRows rows, uint32_t n_total,
                                                          const float* __restrict__ pad_dists,
                                                          uint32_t* __restrict__ out_ids, float* __restrict__ out_dists, uint32_t knn)
{
    __shared__ uint64_t sbuf[4][CAP];
    __shared__ uint32_t shist[4][256];  // digit histograms of the radix select
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x * 4u + w;
    if (q >= nq) return;  // wave-uniform
    uint64_t* buf = sbuf[w];
    uint32_t cnt = 0;
    for (uint32_t s = 0; s < nshards; ++s) {
        const size_t base = ((size_t)s * nq + q) * knn;
        for (uint32_t off = 0; off < knn; off += 64u) {
            if (cnt + 64u > (uint32_t)CAP) {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                hvs_wave_select_prune<CAP / 64>(buf, cnt, knn, lane, shist[w]);
                cnt = knn;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
            const uint32_t e = off + lane;
            const uint32_t id = e < knn ? ids_all[base + e] : 0xFFFFFFFFu;
            const bool have = id != 0xFFFFFFFFu;
            const uint64_t m = __ballot(have);
            if (have) buf[cnt + hvs_prefix_count(m)] = hvs_make_key(dists_all[base + e], (uint32_t)(rows.row0[s] + id));
            cnt += (uint32_t)__popcll(m);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (cnt > knn) {
        hvs_wave_select_prune<CAP / 64>(buf, cnt, knn, lane, shist[w]);
        cnt = knn;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    for (uint32_t b = cnt; b < knn; b += 64u) {
        const uint32_t e = b + lane;
        if (e < knn) buf[e] = hvs_make_key(pad_dists[(size_t)q * knn + (e - cnt)], n_total - 1u - (e - cnt));
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // rank sort of exactly 100 keys (duplicates possible after padding: ties broken by slot)
    for (uint32_t e = lane; e < knn; e += 64u) {
        const uint64_t ke = buf[e];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < knn; ++j) {
            const uint64_t kj = buf[j];
            rank += (kj < ke || (kj == ke && j < e)) ? 1u : 0u;
        }
        out_ids[(size_t)q * knn + rank] = hvs_key_id(ke);
        if (out_dists) out_dists[(size_t)q * knn + rank] = hvs_key_dist(ke);
    }
}

