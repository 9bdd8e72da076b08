// Lab (round 2): which INT8 MFMA shape should the filter loop be built on?
//   part 1 -- bare chains, operands in registers, constant operands (clock stays up): time per instruction of
//             v_mfma_i32_32x32x32_i8, v_mfma_i32_16x16x64_i8 and the legacy half-K forms 32x32x16 / 16x16x32
//             (is a K=16 tail step cheaper than a zero-padded K=32 one?)
//   part 2 -- the filter loop's shape on RANDOM operands (DVFS regime): A fragments + accumulator inits re-read
//             from LDS for every tile, max/threshold epilogue, one barrier per 8 tiles; 32x32x32 with 4 query
//             blocks of 32 per wave against 16x16x64 with 8 query blocks of 16 (same 32 rows x 128 queries per
//             wave and tile, same 64 B-operand and 64 accumulator registers).  MI355X_MICROARCH.md "DVFS
//             give-back (7)" reports 1.12-1.15x for the 16x16 BF16 shape in this regime.
// build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_shape_lab.hip -o shape_lab.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// ---- part 1 ---------------------------------------------------------------------------------------------
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void bare(const uint4* __restrict__ in, int* __restrict__ out, int iters)
{
    const unsigned lane = threadIdx.x & 63u;
    union { uint4 u; i32x4 v; long l[2]; } a, b;
    a.u = in[lane];
    b.u = in[64 + lane];
    int keep = 0;
    if constexpr (SHAPE == 0) {  // 32x32x32
        i32x16 c[4] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, c[j], 0, 0, 0);
        for (int j = 0; j < 4; ++j) keep += c[j][0];
    } else if constexpr (SHAPE == 1) {  // 16x16x64
        i32x4 c[4] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.v, b.v, c[j], 0, 0, 0);
        for (int j = 0; j < 4; ++j) keep += c[j][0];
    } else if constexpr (SHAPE == 2) {  // legacy 32x32x16
        i32x16 c[4] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a.l[0], b.l[0], c[j], 0, 0, 0);
        for (int j = 0; j < 4; ++j) keep += c[j][0];
    } else {  // legacy 16x16x32
        i32x4 c[4] = {};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a.l[0], b.l[0], c[j], 0, 0, 0);
        for (int j = 0; j < 4; ++j) keep += c[j][0];
    }
    out[blockIdx.x * 256 + threadIdx.x] = keep;
}

// ---- part 2 ---------------------------------------------------------------------------------------------
#define TILE_U4 (4 * 64 + 8)  // 4 KiB of A fragments + 32 int32 accumulator inits
#define STG 8
template <int SHAPE16, int OCC, int PRIO = 0>
__global__ __launch_bounds__(256, OCC) void loop(const uint4* __restrict__ in, int* __restrict__ out, int ntiles, int theta_in)
{
    __shared__ uint4 stile[2][STG * TILE_U4];
    __shared__ uint4 spad[OCC == 1 ? 2048 : 1];  // 32 KiB more: one workgroup per CU = one wave per SIMD
    if (ntiles < 0) spad[threadIdx.x % (OCC == 1 ? 2048 : 1)] = in[threadIdx.x];
    const unsigned lane = threadIdx.x & 63u;
    union { uint4 u; i32x4 v; } c;
    i32x4 bq[16];  // 64 B-operand registers either way
    for (int q = 0; q < 16; ++q) {
        c.u = in[(q % 20) * 64 + lane];
        bq[q] = c.v;
        asm volatile("" : "+v"(bq[q]));
    }
    for (int e = threadIdx.x; e < 2 * STG * TILE_U4; e += 256) (&stile[0][0])[e] = in[(e * 7) % (20 * 64)];
    __syncthreads();
    int theta[8];
    for (int q = 0; q < 8; ++q) theta[q] = theta_in + q;
    int keep = 0;
    unsigned hits = 0;
    const int nstage = ntiles / STG;
    for (int st = 0; st < nstage; ++st) {
        const unsigned cur = st & 1;
#pragma unroll 1
        for (int tt = 0; tt < STG; ++tt) {
            i32x4 af[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                c.u = stile[cur][tt * TILE_U4 + s * 64 + lane];
                af[s] = c.v;
            }
            if constexpr (SHAPE16) {
                // rows of a lane's 4 accumulators: 16 rb + 4 (lane >> 4) + 0..3 -> one b128 read per row block
                i32x4 nrm[2];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    c.u = stile[cur][tt * TILE_U4 + 256 + 4 * rb + (lane >> 4)];
                    nrm[rb] = c.v;
                }
                // af[0], af[1]: row block 0, k-steps 0, 1;  af[2], af[3]: row block 1
                i32x4 acc[2][8];
                if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(3);  // the multiplying wave keeps the matrix pipe
                if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) acc[rb][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[2 * rb], bq[2 * q], nrm[rb], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
                        acc[rb][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[2 * rb + 1], bq[2 * q + 1], acc[rb][q], 0, 0, 0);
                if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
                if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(3);  // the wave in its epilogue goes first
                bool anyhit = false;
#ifdef LAB_FOLD
                // thresholds folded into the accumulators (spare K slots): ONE max over the lane's 64 accumulators, one compare
                int mall = acc[0][0][0];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    int m = max(max(acc[0][q][0], acc[0][q][1]), acc[0][q][2]);
                    m = max(max(m, acc[0][q][3]), acc[1][q][0]);
                    m = max(max(m, acc[1][q][1]), acc[1][q][2]);
                    mall = max(max(mall, m), acc[1][q][3]);
                }
                anyhit = mall >= theta[0];
#else
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    int m = max(max(acc[0][q][0], acc[0][q][1]), acc[0][q][2]);
                    m = max(max(m, acc[0][q][3]), acc[1][q][0]);
                    m = max(max(m, acc[1][q][1]), acc[1][q][2]);
                    m = max(m, acc[1][q][3]);
                    anyhit = anyhit | (m >= theta[q]);
                }
#endif
                if (__ballot(anyhit) != 0ull) {
                    hits++;
                    keep += acc[0][0][3];
                }
            } else {
                i32x16 nrm;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    c.u = stile[cur][tt * TILE_U4 + 256 + 2 * g4 + (lane >> 5)];
                    nrm[4 * g4 + 0] = c.v[0];
                    nrm[4 * g4 + 1] = c.v[1];
                    nrm[4 * g4 + 2] = c.v[2];
                    nrm[4 * g4 + 3] = c.v[3];
                }
                i32x16 acc[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bq[4 * q], nrm, 0, 0, 0);
#pragma unroll
                    for (int s = 1; s < 4; ++s) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[s], bq[4 * q + s], acc[q], 0, 0, 0);
                }
                bool anyhit = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int m = max(max(acc[q][0], acc[q][1]), acc[q][2]);
#pragma unroll
                    for (int r = 3; r < 15; r += 2) m = max(max(m, acc[q][r]), acc[q][r + 1]);
                    m = max(m, acc[q][15]);
                    anyhit = anyhit | (m >= theta[q]);
                }
                if (__ballot(anyhit) != 0ull) {
                    hits++;
                    keep += acc[0][3];
                }
            }
        }
        __syncthreads();
    }
    if (ntiles < 0) keep += (int)spad[(threadIdx.x ^ 1) % (OCC == 1 ? 2048 : 1)].x;
    out[blockIdx.x * 256 + threadIdx.x] = keep + (int)hits;
}

// ---- part 5: narrower waves.  NSUB sub-blocks of 16 queries per wave (8 = the kernel's 128 queries; 4 = 64 queries), WAVES waves
// per workgroup sharing the staged tiles, two workgroups per CU: NSUB = 4 with 8 waves per workgroup is FOUR waves per SIMD at
// <= 128 registers -- the same LDS footprint and query coverage per workgroup, half the matrix work per tile and wave (the
// fragment reads and the loop overhead are amortised over 16 instead of 32 matrix instructions), but four waves to fill the pipe.
template <int NSUB, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void loop_narrow(const uint4* __restrict__ in, int* __restrict__ out, int ntiles, int theta_in)
{
    __shared__ uint4 stile[2][STG * TILE_U4];
    const unsigned lane = threadIdx.x & 63u;
    union { uint4 u; i32x4 v; } c;
    i32x4 bq[NSUB][2];
    for (int q = 0; q < NSUB; ++q)
        for (int ks = 0; ks < 2; ++ks) {
            c.u = in[((2 * q + ks + (threadIdx.x >> 6)) % 20) * 64 + lane];
            bq[q][ks] = c.v;
            asm volatile("" : "+v"(bq[q][ks]));
        }
    for (int e = threadIdx.x; e < 2 * STG * TILE_U4; e += 64 * WAVES) (&stile[0][0])[e] = in[(e * 7) % (20 * 64)];
    __syncthreads();
    int theta[NSUB];
    for (int q = 0; q < NSUB; ++q) theta[q] = theta_in + q;
    int keep = 0;
    unsigned hits = 0;
    const int nstage = ntiles / STG;
    i32x4 af[4], nrm[2];
    auto load_tile = [&](unsigned cur, int tt) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            c.u = stile[cur][tt * TILE_U4 + s * 64 + lane];
            af[s] = c.v;
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            c.u = stile[cur][tt * TILE_U4 + 256 + 4 * rb + (lane >> 4)];
            nrm[rb] = c.v;
        }
    };
    load_tile(0u, 0);
    for (int st = 0; st < nstage; ++st) {
        const unsigned cur = st & 1;
#pragma unroll 1
        for (int tt = 0; tt < STG; ++tt) {
            i32x4 acc[2][NSUB];
#pragma unroll
            for (int q = 0; q < NSUB; ++q)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=&v"(acc[rb][q]) : "v"(af[2 * rb]), "v"(bq[q][0]), "v"(nrm[rb]));
#pragma unroll
            for (int q = 0; q < NSUB; ++q)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[rb][q]) : "v"(af[2 * rb + 1]), "v"(bq[q][1]));
            __builtin_amdgcn_sched_barrier(0);
            load_tile(cur, tt + 1 < STG ? tt + 1 : tt);  // (the next tile's reads travel under the epilogue)
            __builtin_amdgcn_sched_barrier(0);
            bool anyhit = false;
#pragma unroll
            for (int q = 0; q < NSUB; ++q) {
                int m = max(max(acc[0][q][0], acc[0][q][1]), acc[0][q][2]);
                m = max(max(m, acc[0][q][3]), acc[1][q][0]);
                m = max(max(m, acc[1][q][1]), acc[1][q][2]);
                m = max(m, acc[1][q][3]);
                anyhit = anyhit | (m >= theta[q]);
            }
            if (__ballot(anyhit) != 0ull) {
                hits++;
                keep += acc[0][0][3];
            }
        }
        __syncthreads();
    }
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = keep + (int)hits;
}

// ---- part 4: the 16x16x64 loop software-pipelined by HALF tiles inside each wave: the 16 matrix instructions of half-tile
// h (16 rows x 128 queries x K=128) are interleaved with the max/threshold epilogue of half-tile h-1 (the other accumulator
// set) -- 1.5 vector instructions per matrix-instruction gap -- and the LDS reads of half-tile h+1 follow the last read of
// the current fragments.  Same registers as the phase-separated loop of part 2 (two sets of 8 accumulator blocks).
template <int OCC>
__global__ __launch_bounds__(256, OCC) void loop_pipe(const uint4* __restrict__ in, int* __restrict__ out, int ntiles, int theta_in)
{
    __shared__ uint4 stile[2][STG * TILE_U4];
    __shared__ uint4 spad[OCC == 1 ? 2048 : 1];
    if (ntiles < 0) spad[threadIdx.x % (OCC == 1 ? 2048 : 1)] = in[threadIdx.x];
    const unsigned lane = threadIdx.x & 63u;
    union { uint4 u; i32x4 v; } c;
    i32x4 bq[16];
    for (int q = 0; q < 16; ++q) {
        c.u = in[(q % 20) * 64 + lane];
        bq[q] = c.v;
        asm volatile("" : "+v"(bq[q]));
    }
    for (int e = threadIdx.x; e < 2 * STG * TILE_U4; e += 256) (&stile[0][0])[e] = in[(e * 7) % (20 * 64)];
    __syncthreads();
    int theta[8];
    for (int q = 0; q < 8; ++q) theta[q] = theta_in + q;
    int keep = 0;
    unsigned hits = 0;
    i32x4 accA[8], accB[8];
    for (int j = 0; j < 8; ++j) accA[j] = accB[j] = i32x4{0, 0, 0, 0};
    i32x4 af0, af1, nh;  // fragments (k-steps 0, 1) and accumulator inits of the half-tile being multiplied
    auto load_half = [&](unsigned cur, int tt, int rb) {
        c.u = stile[cur][tt * TILE_U4 + (2 * rb) * 64 + lane]; af0 = c.v;
        c.u = stile[cur][tt * TILE_U4 + (2 * rb + 1) * 64 + lane]; af1 = c.v;
        c.u = stile[cur][tt * TILE_U4 + 256 + 4 * rb + (lane >> 4)]; nh = c.v;
    };
    // one half-tile: cur <- af * bq + nh (16 matrix instructions), epilogue of prv in the gaps; hm = lanes of prv at/above threshold
    auto half = [&](i32x4 (&cur)[8], const i32x4 (&prv)[8], unsigned long long (&hm)[8], unsigned long long& any) {
        int t[8];
        any = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=&v"(cur[j]) : "v"(af0), "v"(bq[2 * j]), "v"(nh));
            if (j & 1) {
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(t[j >> 1]) : "v"(prv[j >> 1][3]));
                asm volatile("v_cmp_ge_i32_e64 %0, %1, %2" : "=s"(hm[j >> 1]) : "v"(t[j >> 1]), "v"(theta[j >> 1]));
            } else {
                asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(t[j >> 1]) : "v"(prv[j >> 1][0]), "v"(prv[j >> 1][1]), "v"(prv[j >> 1][2]));
                if (j) asm volatile("s_or_b64 %0, %0, %1" : "+s"(any) : "s"(hm[(j >> 1) - 1]));
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(cur[j]) : "v"(af1), "v"(bq[2 * j + 1]));
            if (j & 1) {
                asm volatile("v_max_i32 %0, %0, %1" : "+v"(t[4 + (j >> 1)]) : "v"(prv[4 + (j >> 1)][3]));
                asm volatile("v_cmp_ge_i32_e64 %0, %1, %2" : "=s"(hm[4 + (j >> 1)]) : "v"(t[4 + (j >> 1)]), "v"(theta[4 + (j >> 1)]));
            } else {
                asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(t[4 + (j >> 1)]) : "v"(prv[4 + (j >> 1)][0]), "v"(prv[4 + (j >> 1)][1]), "v"(prv[4 + (j >> 1)][2]));
                asm volatile("s_or_b64 %0, %0, %1" : "+s"(any) : "s"(hm[3 + (j >> 1)]));
            }
        }
        asm volatile("s_or_b64 %0, %0, %1" : "+s"(any) : "s"(hm[7]));
    };
    const int nstage = ntiles / STG;
    load_half(0u, 0, 0);
    for (int st = 0; st < nstage; ++st) {
        const unsigned cur = st & 1;
#pragma unroll 1
        for (int tt = 0; tt < STG; ++tt) {
            unsigned long long hm[8], any;
            half(accA, accB, hm, any);            // rows 0..15 of tile tt; epilogue of the previous tile's rows 16..31
            load_half(cur, tt, 1);
            if (any) { hits++; keep += accB[0][3]; }
            half(accB, accA, hm, any);            // rows 16..31; epilogue of rows 0..15
            load_half(cur, tt + 1 < STG ? tt + 1 : tt, 0);
            if (any) { hits++; keep += accA[0][3]; }
        }
        __syncthreads();
    }
    if (ntiles < 0) keep += (int)spad[(threadIdx.x ^ 1) % (OCC == 1 ? 2048 : 1)].x;
    out[blockIdx.x * 256 + threadIdx.x] = keep + (int)hits + accA[1][0] + accB[1][0];
}

// ---- part 3: bare chains on RANDOM operands, inline asm (no compiler shuffles), 16 independent accumulator blocks per wave:
// the chip's power / clock ceiling for each shape with nothing but matrix instructions in the loop
template <int SHAPE16, int OCC>
__global__ __launch_bounds__(256, OCC) void bare_random(const uint4* __restrict__ in, int* __restrict__ out, int iters)
{
    __shared__ uint4 spad[OCC == 1 ? 6144 : 1];  // 96 KiB: one workgroup per CU = one wave per SIMD
    if (iters < 0) spad[threadIdx.x % (OCC == 1 ? 6144 : 1)] = in[threadIdx.x];
    const unsigned lane = threadIdx.x & 63u;
    union { uint4 u; i32x4 v; } c;
    i32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        c.u = in[(i * 3 + 1) * 64 + lane]; a[i] = c.v;
        c.u = in[(i * 5 + 2) * 64 + lane]; b[i] = c.v;
    }
    int keep = 0;
    if constexpr (SHAPE16) {
        i32x4 acc[16];
        for (int j = 0; j < 16; ++j) acc[j] = i32x4{j, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[j & 3]), "v"(b[(j >> 2) & 3]));
        }
        for (int j = 0; j < 16; ++j) keep += acc[j][0];
    } else {
        i32x16 acc[4];
        for (int j = 0; j < 4; ++j) acc[j] = i32x16{j, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[r]), "v"(b[j]));
        }
        for (int j = 0; j < 4; ++j) keep += acc[j][0];
    }
    if (iters < 0) keep += (int)spad[(threadIdx.x ^ 1) % (OCC == 1 ? 6144 : 1)].x;
    out[blockIdx.x * 256 + threadIdx.x] = keep;
}

static float run(void (*launch)(int), int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        launch(r);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    return best;
}

static uint4* g_in;
static int* g_out;
static int g_iters = 20000, g_tiles = 8192;
template <int S> static void l_bare(int) { hipLaunchKernelGGL(bare<S>, dim3(256), dim3(256), 0, 0, g_in, g_out, g_iters); }
template <int S, int O> static void l_brand(int) { hipLaunchKernelGGL((bare_random<S, O>), dim3(O == 1 ? 1024 : 2048), dim3(256), 0, 0, g_in, g_out, g_iters); }
template <int O> static void l_pipe(int) { hipLaunchKernelGGL((loop_pipe<O>), dim3(O == 1 ? 1024 : 2048), dim3(256), 0, 0, g_in, g_out, g_tiles, 0x7fffff00); }
template <int P> static void l_prio(int) { hipLaunchKernelGGL((loop<1, 2, P>), dim3(2048), dim3(256), 0, 0, g_in, g_out, g_tiles, 0x7fffff00); }
template <int NS, int WV> static void l_narrow(int) { hipLaunchKernelGGL((loop_narrow<NS, WV>), dim3(2048), dim3(64 * WV), 0, 0, g_in, g_out, g_tiles, 0x7fffff00); }
template <int S, int O> static void l_loop(int) { hipLaunchKernelGGL((loop<S, O>), dim3(O == 1 ? 1024 : 2048), dim3(256), 0, 0, g_in, g_out, g_tiles, 0x7fffff00); }

int main(int argc, char** argv)
{
    std::vector<unsigned> h(20 * 64 * 4, 0x01010101u);
    hipMalloc(&g_in, h.size() * 4);
    hipMalloc(&g_out, 4096 * 256 * 4);
    hipMemcpy(g_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    // part 1: one 256-thread block per CU = one wave per SIMD; cycles = time * 2.4 GHz / instructions per wave
    const char* names[4] = {"32x32x32", "16x16x64", "32x32x16 (legacy)", "16x16x32 (legacy)"};
    float ms[4] = {run(l_bare<0>, 3), run(l_bare<1>, 3), run(l_bare<2>, 3), run(l_bare<3>, 3)};
    for (int s = 0; s < 4; ++s)
        std::printf("bare %-18s %8.3f ms  %.1f cycles/instruction at 2.4 GHz (constant operands, 1 wave/SIMD)\n", names[s], ms[s],
                    ms[s] * 1e-3 * 2.4e9 / (4.0 * g_iters));
    // part 2: random operands
    unsigned long long st = 12345;
    for (size_t i = 0; i < h.size(); ++i) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        h[i] = (unsigned)(st >> 32);
    }
    hipMemcpy(g_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    if (argc > 2 && !std::strcmp(argv[1], "hold")) {  // keep ONE kernel running for ~4 s so rocm-smi can sample clocks/power under it
        void (*f)(int) = !std::strcmp(argv[2], "bare16") ? l_brand<1, 2> : !std::strcmp(argv[2], "bare32") ? l_brand<0, 2> :
                         !std::strcmp(argv[2], "loop16") ? l_loop<1, 2> : !std::strcmp(argv[2], "pipe16") ? l_pipe<2> : l_pipe<1>;
        double total = 0;
        int launches = 0;
        while (total < 4000.0) { total += run(f, 5) * 5; launches += 5; }
        std::printf("hold %s: %d launches, %.2f ms each\n", argv[2], launches, total / launches);
        return 0;
    }
    if (argc > 1 && !std::strcmp(argv[1], "bits")) {
        // part 4 (round 3): does the operands' bit width move the rate (DVFS: fewer toggling multiplier bits, higher clock)?
        // every int8 operand byte is a random value shifted right arithmetically by `sh` bits (range +-127 >> sh); `zeros`:
        // 28 of every 128 K positions are zero, as in the filter's K = 100 tiles
        for (int zeros = 0; zeros < 2; ++zeros)
            for (int sh = 0; sh <= 4; ++sh) {
                unsigned long long s2 = 999;
                std::vector<unsigned> hb(h.size());
                for (size_t i = 0; i < hb.size(); ++i) {
                    unsigned w = 0;
                    for (int b = 0; b < 4; ++b) {
                        s2 = s2 * 6364136223846793005ull + 1442695040888963407ull;
                        int v = (int)(signed char)(s2 >> 40) >> sh;
                        // K position of this byte inside the 16-byte fragment row: lanes 48..63 of the second k-step hold k = 112..127
                        if (zeros && ((i * 4 + b) % 64) >= 50) v = 0;  // ~22 % of the bytes
                        w |= ((unsigned)v & 0xFFu) << (8 * b);
                    }
                    hb[i] = w;
                }
                hipMemcpy(g_in, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
                const float lo = run(l_loop<1, 2>, 3), br = run(l_brand<1, 2>, 3);
                std::printf("operands +-%3d%s: filter-shaped loop 16x16x64 %.2f ms %.2f G pair-blocks/s | bare chains %.2f ms %.2f G\n", 127 >> sh,
                            zeros ? ", 22 %% zero bytes" : "", lo, 2048.0 * 4 * g_tiles * 4 / (lo * 1e-3) / 1e9, br,
                            2048.0 * 4 * g_iters * 2 / (br * 1e-3) / 1e9);
            }
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        const float a2 = run(l_loop<0, 2>, 3), b2 = run(l_loop<1, 2>, 3), a1 = run(l_loop<0, 1>, 3), b1 = run(l_loop<1, 1>, 3);
        const double pb2 = 2048.0 * 4 * g_tiles * 4, pb1 = 1024.0 * 4 * g_tiles * 4;  // 32x32 pair blocks
        std::printf("loop random  2 waves/SIMD: 32x32x32 %.2f ms %.2f G pair-blocks/s | 16x16x64 %.2f ms %.2f G  (ratio %.3f)\n", a2,
                    pb2 / (a2 * 1e-3) / 1e9, b2, pb2 / (b2 * 1e-3) / 1e9, a2 / b2);
        std::printf("loop random  1 wave /SIMD: 32x32x32 %.2f ms %.2f G pair-blocks/s | 16x16x64 %.2f ms %.2f G  (ratio %.3f)\n", a1,
                    pb1 / (a1 * 1e-3) / 1e9, b1, pb1 / (b1 * 1e-3) / 1e9, a1 / b1);
    }
    for (int rep = 0; rep < 2; ++rep) {
        const float p2 = run(l_pipe<2>, 3), p1 = run(l_pipe<1>, 3);
        std::printf("loop random, half-tile software pipeline, 16x16x64: 2 waves/SIMD %.2f ms %.2f G pair-blocks/s | 1 wave/SIMD %.2f ms %.2f G\n", p2,
                    2048.0 * 4 * g_tiles * 4 / (p2 * 1e-3) / 1e9, p1, 1024.0 * 4 * g_tiles * 4 / (p1 * 1e-3) / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        const float q1 = run(l_prio<1>, 3), q2 = run(l_prio<2>, 3);
        std::printf("loop random 16x16x64 2 waves/SIMD with s_setprio: high while multiplying %.2f ms %.2f G | high in the epilogue %.2f ms %.2f G\n", q1,
                    2048.0 * 4 * g_tiles * 4 / (q1 * 1e-3) / 1e9, q2, 2048.0 * 4 * g_tiles * 4 / (q2 * 1e-3) / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        const float n84 = run(l_narrow<8, 4>, 3), n48 = run(l_narrow<4, 8>, 3), n44 = run(l_narrow<4, 4>, 3);
        // pair blocks per launch: 2048 workgroups x waves x tiles x (NSUB / 2)
        std::printf("loop random 16x16x64 asm chains: 128 queries/wave, 2 waves/SIMD %.2f ms %.2f G | 64 queries/wave, 4 waves/SIMD %.2f ms %.2f G | "
                    "64 queries/wave, 2 waves/SIMD %.2f ms %.2f G\n", n84, 2048.0 * 4 * g_tiles * 4 / (n84 * 1e-3) / 1e9, n48,
                    2048.0 * 8 * g_tiles * 2 / (n48 * 1e-3) / 1e9, n44, 2048.0 * 4 * g_tiles * 2 / (n44 * 1e-3) / 1e9);
    }
    // part 3: per iteration a wave issues 16 instructions = 4 (32x32x32: 16 x 65536 MAC... ) resp. 1 (16x16x64) pair blocks of 32x32x128
    for (int rep = 0; rep < 2; ++rep) {
        const float x2 = run(l_brand<0, 2>, 3), y2 = run(l_brand<1, 2>, 3), x1 = run(l_brand<0, 1>, 3), y1 = run(l_brand<1, 1>, 3);
        // pair blocks (32 rows x 32 queries x K=128) per launch: 32x32x32: 16 instr = 4 blocks; 16x16x64: 16 instr = 2 blocks
        const double w2 = 2048.0 * 4 * g_iters, w1 = 1024.0 * 4 * g_iters;
        std::printf("bare random 2 waves/SIMD: 32x32x32 %.2f ms %.2f G pair-blocks/s | 16x16x64 %.2f ms %.2f G\n", x2, w2 * 4 / (x2 * 1e-3) / 1e9, y2,
                    w2 * 2 / (y2 * 1e-3) / 1e9);
        std::printf("bare random 1 wave /SIMD: 32x32x32 %.2f ms %.2f G pair-blocks/s | 16x16x64 %.2f ms %.2f G\n", x1, w1 * 4 / (x1 * 1e-3) / 1e9, y1,
                    w1 * 2 / (y1 * 1e-3) / 1e9);
    }
    return 0;
}
