// Lab: what does a filter loop built on v_mfma_i32_32x32x32_i8 sustain on this chip?  Same structure as
// mfma_loop_lab.hip (bare loop, then the kernel's ingredients by compile-time switches), INT8 operands:
// a 32-row tile is 4 k-steps x 64 lanes x 16 B = 4 KiB (K = 128), the per-row norm term enters as the
// accumulator init (C operand of the first MFMA, 16 x int32 per lane read from LDS).
//   -DLAB_EPI      v_max3_i32 trees + threshold compares (+ never-taken survivor branch)
//   -DLAB_BARRIER  one __syncthreads() per 4 tiles
//   -DLAB_LDS      A fragments (4 ds_read_b128) + row norms (4 ds_read_b128) re-read from LDS every tile
//   -DLAB_DMA      LDS-DMA of the next stage (4 tiles x 4 KiB + norms), inline asm as the kernel
//   -DLAB_RANDOM   random int8 operands (all bits toggle)
//   -DLAB_PAD_LDS=N  N bytes of unused LDS per workgroup (limits the workgroups per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define KS 4
#define TILE_U4 (KS * 64 + 8)  // 4 KiB of fragments + 32 int32 norms
#ifndef LAB_QB
#define LAB_QB 4
#endif
#ifndef LAB_OCC
#define LAB_OCC 2
#endif
#ifndef LAB_NAME
#define LAB_NAME "i8lab"
#endif
#define STAGE_U4 (4 * TILE_U4)  // 1056 uint4 per stage = 16.5 KiB; 4 waves x 64 lanes move 256 per instruction

__global__ __launch_bounds__(256, LAB_OCC) void lab(const uint4* __restrict__ in, const uint4* __restrict__ tiles, int* __restrict__ out,
                                                    int ntiles, int theta_in)
{
    __shared__ uint4 stile[2][STAGE_U4 + 256];
#ifdef LAB_PAD_LDS
    __shared__ uint4 spad[LAB_PAD_LDS / 16];  // occupancy limiter: -DLAB_PAD_LDS=bytes
    if (ntiles < 0) spad[threadIdx.x] = in[threadIdx.x];
#endif
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    union { uint4 u; i32x4 b; } c;
    i32x4 bq[LAB_QB][KS], af[KS];
    for (int q = 0; q < LAB_QB; ++q)
        for (int s = 0; s < KS; ++s) { c.u = in[((q % 4) * KS + s) * 64 + lane]; bq[q][s] = c.b; asm volatile("" : "+v"(bq[q][s])); }
    for (int s = 0; s < KS; ++s) { c.u = in[(16 + s) * 64 + lane]; af[s] = c.b; }
    for (int e = threadIdx.x; e < 2 * (STAGE_U4 + 256); e += 256) (&stile[0][0])[e] = in[e % (20 * 64)];
    __syncthreads();
    int theta[LAB_QB]; unsigned ra[LAB_QB], rb[LAB_QB];
    for (int q = 0; q < LAB_QB; ++q) { theta[q] = theta_in + q; ra[q] = 0; rb[q] = ~0u; }
    int keep = 0;
    unsigned hits = 0;
    i32x16 nrm = i32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int nstage = ntiles / 4;
    for (int st = 0; st < nstage; ++st) {
        const unsigned cur = st & 1;
#ifdef LAB_DMA
#pragma unroll
        for (int k = 0; k < 5; ++k) {  // 5 x 256 uint4 >= 1056 per stage (the last one partly redundant)
            const unsigned ch = __builtin_amdgcn_readfirstlane(wv) + 4u * k;
            const uint4* src = tiles + ((size_t)(blockIdx.x % 64) * 4096 + (size_t)((st + 1) % 1024) * 4) * TILE_U4 + ch * 64 + lane;
            const uint4* dst = &stile[cur ^ 1][ch * 64];
            const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)dst);
            unsigned keepm0;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keepm0) : "v"(src), "s"(lds_addr) : "memory");
        }
#endif
#pragma unroll 1
        for (int tt = 0; tt < 4; ++tt) {
            const unsigned bp = (unsigned)(st * 4 + tt) * 3u + 1u;
#ifdef LAB_LDS
#pragma unroll
            for (int s = 0; s < KS; ++s) { c.u = stile[cur][tt * TILE_U4 + s * 64 + lane]; af[s] = c.b; }
            // row norms: accumulator r of a lane belongs to row (r&3) + 8*(r>>2) + 4*(lane>>5): 4 broadcast b128 reads
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                c.u = stile[cur][tt * TILE_U4 + KS * 64 + 2 * g4 + (lane >> 5)];
                nrm[4 * g4 + 0] = c.b[0]; nrm[4 * g4 + 1] = c.b[1]; nrm[4 * g4 + 2] = c.b[2]; nrm[4 * g4 + 3] = c.b[3];
            }
#else
            asm volatile("" : "+v"(af[0]));
            asm volatile("" : "+v"(nrm));
#endif
            i32x16 acc[LAB_QB];
#pragma unroll
            for (int q = 0; q < LAB_QB; ++q) {
                acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bq[q][0], nrm, 0, 0, 0);
#pragma unroll
                for (int s = 1; s < KS; ++s) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[s], bq[q][s], acc[q], 0, 0, 0);
            }
#ifdef LAB_EPI
            bool anyhit = false;
#pragma unroll
            for (int q = 0; q < LAB_QB; ++q) {
                int m = max(max(acc[q][0], acc[q][1]), acc[q][2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) m = max(max(m, acc[q][r]), acc[q][r + 1]);
                m = max(m, acc[q][15]);
                anyhit = anyhit | ((m >= theta[q]) & (bp * 32u + 32u > ra[q]) & (bp * 32u < rb[q]));
            }
            if (__ballot(anyhit) != 0ull) { hits++; keep += acc[0][3]; }
#else
#pragma unroll
            for (int q = 0; q < LAB_QB; ++q) keep += acc[q][0];
#endif
        }
#ifdef LAB_BARRIER
#ifdef LAB_DMA
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
#endif
    }
#ifdef LAB_PAD_LDS
    if (ntiles < 0) keep += (int)spad[threadIdx.x ^ 1].x;
#endif
    out[blockIdx.x * 256 + threadIdx.x] = keep + (int)hits;
}

int main(int argc, char** argv)
{
    const int ntiles = argc > 1 ? atoi(argv[1]) : 8192;
    uint4 *in, *tiles; int* out;
    std::vector<unsigned> h(20 * 64 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x01010101u;
#ifdef LAB_RANDOM
    auto rb = [](unsigned long long& st) { st = st * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(st >> 32); };
    unsigned long long stt = 12345;
    for (size_t i = 0; i < h.size(); ++i) h[i] = rb(stt);
#endif
    hipMalloc(&in, h.size() * 4); hipMalloc(&out, 4096 * 256 * 4);
    const size_t tile_bytes = ((size_t)64 * 4096 + 8) * TILE_U4 * 16;
    hipMalloc(&tiles, tile_bytes); hipMemset(tiles, 0x01, tile_bytes);
#ifdef LAB_RANDOM
    {
        std::vector<unsigned> big(64u << 20);
        unsigned long long s2 = 777;
        for (size_t i = 0; i < big.size(); ++i) big[i] = rb(s2);
        for (size_t off = 0; off + big.size() * 4 <= tile_bytes; off += big.size() * 4) hipMemcpy((char*)tiles + off, big.data(), big.size() * 4, hipMemcpyHostToDevice);
    }
#endif
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = argc > 2 ? atoi(argv[2]) : (LAB_OCC == 1 ? 1024 : 2048);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(lab, dim3(blocks), dim3(256), 0, 0, in, tiles, out, ntiles, 0x7fffff00);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double mfma = (double)blocks * 4 * ntiles * KS * LAB_QB;
        // 32x32x32 MACs = 65536 int ops per instruction; a tile of 32 rows x (LAB_QB*32) queries per wave
        std::printf("%s: %.2f ms  %.0f TOP/s  %.1f G tile-blocks/s (32x32 pair blocks)\n", LAB_NAME, ms, mfma * 65536.0 / (ms * 1e-3) / 1e12,
                    (double)blocks * 4 * ntiles * LAB_QB / (ms * 1e-3) / 1e9);
    }
    return 0;
}
