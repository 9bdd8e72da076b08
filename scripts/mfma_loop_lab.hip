// Lab for the filter kernel's steady-state loop: starts from the bare MFMA loop (mfma_peak.hip) and adds
// the real kernel's ingredients one by one (compile-time switches) to see which one costs MFMA rate.
//   -DLAB_EPI      v_max3 trees + threshold compares (+ never-taken survivor branch)
//   -DLAB_BARRIER  one __syncthreads() per 4 tiles
//   -DLAB_LDS      A fragments re-read from LDS every tile (7 ds_read_b128)
//   -DLAB_DMA      7 LDS-DMA chunks per wave per 4 tiles (inline asm, as the kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define TILE_U4 448
#ifndef LAB_QB
#define LAB_QB 4
#endif
#ifndef LAB_OCC
#define LAB_OCC 2
#endif
#ifndef LAB_DMA_POLICY
#define LAB_DMA_POLICY ""
#endif

__global__ __launch_bounds__(256, LAB_OCC) void lab(const uint4* __restrict__ in, const uint4* __restrict__ tiles, float* __restrict__ out,
                                              int ntiles, float theta_in)
{
    __shared__ uint4 stile[2][4 * TILE_U4];
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    union { uint4 u; bf16x8 b; } c;
    bf16x8 bq[LAB_QB][7], af[7];
    for (int q = 0; q < LAB_QB; ++q)
        for (int s = 0; s < 7; ++s) { c.u = in[((q % 4) * 7 + s) * 64 + lane]; bq[q][s] = c.b; asm volatile("" : "+v"(bq[q][s])); }
    for (int s = 0; s < 7; ++s) { c.u = in[(28 + s) * 64 + lane]; af[s] = c.b; }
    for (int e = threadIdx.x; e < 2 * 4 * TILE_U4; e += 256) (&stile[0][0])[e] = in[e % (35 * 64)];
    __syncthreads();
    float theta[LAB_QB]; unsigned ra[LAB_QB], rb[LAB_QB];
    for (int q = 0; q < LAB_QB; ++q) { theta[q] = theta_in + (float)q; ra[q] = 0; rb[q] = ~0u; }
    float keep = 0.f;
    unsigned hits = 0;
    const int nstage = ntiles / 4;
    for (int st = 0; st < nstage; ++st) {
        const unsigned cur = st & 1;
#ifdef LAB_REGSTAGE
        uint4 stg[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const unsigned ch = wv + 4u * k;
            stg[k] = tiles[((size_t)(blockIdx.x % 64) * 4096 + (size_t)((st + 1) % 1024) * 4 + ch / 7) * TILE_U4 + (ch % 7) * 64 + lane];
        }
#endif
#ifdef LAB_DMA
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const unsigned ch = __builtin_amdgcn_readfirstlane(wv) + 4u * k;
            const uint4* src = tiles + ((size_t)(blockIdx.x % 64) * 4096 + (size_t)((st + 1) % 1024) * 4 + ch / 7) * TILE_U4 + (ch % 7) * 64 + lane;
            const uint4* dst = &stile[cur ^ 1][(ch / 7) * TILE_U4 + (ch % 7) * 64];
            const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)dst);
            unsigned keepm0;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" LAB_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keepm0) : "v"(src), "s"(lds_addr) : "memory");
        }
#endif
#pragma unroll 1
        for (int tt = 0; tt < 4; ++tt) {
            const unsigned bp = (unsigned)(st * 4 + tt) * 3u + 1u;
#ifdef LAB_LDS
#pragma unroll
            for (int s = 0; s < 7; ++s) { c.u = stile[cur][tt * TILE_U4 + s * 64 + lane]; af[s] = c.b; }
#else
            asm volatile("" : "+v"(af[0]));
#endif
            f32x16 acc[LAB_QB];
#pragma unroll
            for (int q = 0; q < LAB_QB; ++q) {
                acc[q] = f32x16{0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
#pragma unroll
                for (int s = 0; s < 7; ++s) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], bq[q][s], acc[q], 0, 0, 0);
            }
#ifdef LAB_EPI
            bool anyhit = false;
#pragma unroll
            for (int q = 0; q < LAB_QB; ++q) {
                float m = fmaxf(fmaxf(acc[q][0], acc[q][1]), acc[q][2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) m = fmaxf(fmaxf(m, acc[q][r]), acc[q][r + 1]);
                m = fmaxf(m, acc[q][15]);
                anyhit = anyhit | ((m >= theta[q]) & (bp * 32u + 32u > ra[q]) & (bp * 32u < rb[q]));
            }
            if (__ballot(anyhit) != 0ull) { hits++; keep += acc[0][3]; }
#else
            keep += acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
#endif
        }
#ifdef LAB_REGSTAGE
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const unsigned ch = wv + 4u * k;
            stile[cur ^ 1][(ch / 7) * TILE_U4 + (ch % 7) * 64 + lane] = stg[k];
        }
#endif
#ifdef LAB_BARRIER
#ifdef LAB_DMA
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
#endif
    }
    out[blockIdx.x * 256 + threadIdx.x] = keep + (float)hits;
}

int main(int argc, char** argv)
{
    const int ntiles = argc > 1 ? atoi(argv[1]) : 8192;
    uint4 *in, *tiles; float* out;
    std::vector<unsigned> h(35 * 64 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3F803F80u ^ (unsigned)((i * 2654435761u) & 0x007F007Fu);
#ifdef LAB_RANDOM
    // random BF16 operands in [-6,6) like the real tiles (sign, exponent and mantissa bits all toggle)
    auto rb = [](unsigned long long& st) { st = st * 6364136223846793005ull + 1442695040888963407ull; float f = ((st >> 40) * (1.0f / 16777216.0f)) * 12.0f - 6.0f; unsigned u; memcpy(&u, &f, 4); return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16; };
    unsigned long long stt = 12345;
    for (size_t i = 0; i < h.size(); ++i) { unsigned lo = rb(stt), hi = rb(stt); h[i] = lo | (hi << 16); }
#endif
    hipMalloc(&in, h.size() * 4); hipMalloc(&out, 4096 * 256 * 4);
    const size_t tile_bytes = (size_t)64 * 4096 * TILE_U4 * 16;  // 1.9 GB of tiles to stream
    hipMalloc(&tiles, tile_bytes); hipMemset(tiles, 0x3c, tile_bytes);
#ifdef LAB_RANDOM
    {
        std::vector<unsigned> big(64u << 20);
        unsigned long long s2 = 777;
        for (size_t i = 0; i < big.size(); ++i) { unsigned lo = rb(s2), hi = rb(s2); big[i] = lo | (hi << 16); }
        for (size_t off = 0; off + big.size() * 4 <= tile_bytes; off += big.size() * 4) hipMemcpy((char*)tiles + off, big.data(), big.size() * 4, hipMemcpyHostToDevice);
    }
#endif
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = LAB_OCC == 1 ? 1024 : 2048;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(lab, dim3(blocks), dim3(256), 0, 0, in, tiles, out, ntiles, 1.0e30f);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double mfma = (double)blocks * 4 * ntiles * 7 * LAB_QB;
        std::printf("%s: %.2f ms  %.0f TFLOP/s\n", LAB_NAME, ms, mfma * 32768.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
