// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate with the filter kernel's register shape
// (4 interleaved chains of 7, 2 waves per SIMD), with and without the v_max3 epilogue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int EPI>
__global__ __launch_bounds__(256, 2) void k(const uint4* __restrict__ in, float* __restrict__ out, int iters)
{
    const unsigned lane = threadIdx.x & 63u;
    union { uint4 u; bf16x8 b; } c;
    bf16x8 bq[4][7], af[7];
    for (int q = 0; q < 4; ++q)
        for (int s = 0; s < 7; ++s) { c.u = in[(q * 7 + s) * 64 + lane]; bq[q][s] = c.b; }
    for (int s = 0; s < 7; ++s) { c.u = in[(28 + s) * 64 + lane]; af[s] = c.b; }
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        f32x16 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[q] = f32x16{0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], bq[q][s], acc[q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (EPI) {
                float m = fmaxf(fmaxf(acc[q][0], acc[q][1]), acc[q][2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) m = fmaxf(fmaxf(m, acc[q][r]), acc[q][r + 1]);
                m = fmaxf(m, acc[q][15]);
                keep = fmaxf(keep, m);
            } else {
                keep += acc[q][0];
            }
        }
        asm volatile("" : "+v"(af[0]));
    }
    out[blockIdx.x * 256 + threadIdx.x] = keep;
}

int main(int argc, char** argv)
{
    uint4* in; float* out;
    std::vector<unsigned> h(35 * 64 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3F803F80u ^ (unsigned)((i * 2654435761u) & 0x007F007Fu);  // bf16 values near 1
    hipMalloc(&in, h.size() * 4); hipMalloc(&out, 2048 * 256 * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = (argc > 1 ? atoi(argv[1]) : 4000), blocks = 512;  // 2 workgroups per CU
    for (int epi = 0; epi < 2; ++epi) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            if (epi) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
            else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double mfma = (double)blocks * 4 * iters * 28;
            const double tf = mfma * 32768.0 / (ms * 1e-3) / 1e12;
            const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 28 * 2);  // per MFMA per SIMD at 2.4 GHz nominal
            std::printf("epilogue=%d rep=%d: %.3f ms  %.0f TFLOP/s  (%.1f nominal-2.4GHz cycles per MFMA per SIMD)\n", epi, rep, ms, tf, cyc);
        }
    }
    return 0;
}
