// Measures the accumulation error of one v_mfma_f32_32x32x16_bf16 chain of 7 k-steps (K = 112, the
// filter kernel's contraction) against an exact f64 evaluation of the same BF16 products, in units
// of u * sum|a_k b_k| (u = 2^-24).  DESIGN.md 3.1 budgets mu = 256 u sum|terms| for it.
// Prints the largest ratio seen; exit code 1 if it exceeds 64 (a 4x margin below the budget).
// -DCHECK_F16: the same for v_mfma_f32_32x32x16_f16 (the FP16 tile format).  Operands stay inside the half range; the
// "wide-range" tiles hold half DENORMALS, which the matrix pipe may flush to zero: their error is measured against the
// allowance the bound makes for it (2^-14 per such element times the other operand, HVS_F16_FLUSH in csrc/hvs_filter.h).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef CHECK_F16
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void chain(const uint4* __restrict__ afrag, const uint4* __restrict__ bfrag, float* __restrict__ out)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned t = blockIdx.x;  // one 32x32 tile per block (64 threads)
    union { uint4 u; bf16x8 b; } a, b;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int ks = 0; ks < 7; ++ks) {
        a.u = afrag[(t * 7 + ks) * 64 + lane];
        b.u = bfrag[(t * 7 + ks) * 64 + lane];
        acc = MFMA16(a.b, b.b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const unsigned row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31u;
        out[(t * 32 + row) * 32 + col] = acc[r];
    }
}

#ifdef CHECK_F16
static unsigned short bf16_bits(float f) { _Float16 h = (_Float16)f; unsigned short b; memcpy(&b, &h, 2); return b; }
static float bf16_val(unsigned short b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }
#else
static unsigned short bf16_bits(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }
static float bf16_val(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
#endif

int main()
{
    const int T = 256;  // tiles
    std::vector<float> A((size_t)T * 32 * 112), Bm((size_t)T * 112 * 32);
    srand(7);
    auto rnd = [](double lo, double hi) { return lo + (hi - lo) * (rand() / (double)RAND_MAX); };
    for (int t = 0; t < T; ++t) {
        const int mode = t % 4;  // 0: [-6,6) like the data; 1: wide dynamic range; 2: heavy cancellation; 3: like the real tile (norm pieces)
        for (int i = 0; i < 32; ++i)
            for (int k = 0; k < 112; ++k) {
                double a = rnd(-6, 6), b = rnd(-6, 6);
#ifdef CHECK_F16
                if (mode == 1) { a *= pow(2.0, -(rand() % 22)); }   // down to 6 x 2^-21: half denormals
#else
                if (mode == 1) { a *= pow(2.0, (rand() % 40) - 20); b *= pow(2.0, (rand() % 40) - 20); }
#endif
                if (mode == 2) { a = (k & 1) ? 1000.0 + rnd(-1, 1) : -1000.0 + rnd(-1, 1); b = 1.0 + rnd(-1e-2, 1e-2); }
                if (mode == 3 && k >= 100) { a = k == 100 ? -rnd(500, 1800) : (k == 101 ? rnd(-4, 4) : (k == 102 ? rnd(-0.02, 0.02) : 0.0)); }
                A[((size_t)t * 32 + i) * 112 + k] = bf16_val(bf16_bits((float)a));
            }
        for (int k = 0; k < 112; ++k)
            for (int j = 0; j < 32; ++j) {
                double b = rnd(-6, 6);
#ifndef CHECK_F16
                if (mode == 1) b *= pow(2.0, (rand() % 40) - 20);
#endif
                if (mode == 2) b = 1.0 + rnd(-1e-2, 1e-2);
                if (mode == 3 && k >= 100) b = k < 103 ? 1.0 : 0.0;
                Bm[((size_t)t * 112 + k) * 32 + j] = bf16_val(bf16_bits((float)b));
            }
    }
    // fragment layouts of cdna_hip_programming.md section 3: lane l holds A[l&31][8(l>>5)+j], B[8(l>>5)+j][l&31]
    std::vector<unsigned> af((size_t)T * 7 * 64 * 4), bf((size_t)T * 7 * 64 * 4);
    for (int t = 0; t < T; ++t)
        for (int ks = 0; ks < 7; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int p = 0; p < 4; ++p) {
                    unsigned wa = 0, wb = 0;
                    for (int e = 0; e < 2; ++e) {
                        const int k = 16 * ks + 8 * (l >> 5) + 2 * p + e;
                        wa |= (unsigned)bf16_bits(A[((size_t)t * 32 + (l & 31)) * 112 + k]) << (16 * e);
                        wb |= (unsigned)bf16_bits(Bm[((size_t)t * 112 + k) * 32 + (l & 31)]) << (16 * e);
                    }
                    af[(((size_t)t * 7 + ks) * 64 + l) * 4 + p] = wa;
                    bf[(((size_t)t * 7 + ks) * 64 + l) * 4 + p] = wb;
                }
    uint4 *da, *db; float* dout;
    if (hipMalloc(&da, af.size() * 4) != hipSuccess) { std::printf("no GPU\n"); return 2; }
    hipMalloc(&db, bf.size() * 4); hipMalloc(&dout, (size_t)T * 32 * 32 * 4);
    hipMemcpy(da, af.data(), af.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bf.data(), bf.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain, dim3(T), dim3(64), 0, 0, da, db, dout);
    std::vector<float> out((size_t)T * 32 * 32);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    double worst[4] = {0, 0, 0, 0};
    size_t wrong_layout = 0;
    bool flushed_seen = false;
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double exact = 0.0, sabs = 0.0, flush = 0.0;
                for (int k = 0; k < 112; ++k) {
                    const double av = (double)A[((size_t)t * 32 + i) * 112 + k], bv = (double)Bm[((size_t)t * 112 + k) * 32 + j];
                    const double p = av * bv;
                    exact += p;
                    sabs += std::fabs(p);
#ifdef CHECK_F16
                    if (std::fabs(av) < 6.103515625e-05) flush += std::fabs(p);   // a denormal half the pipe may have flushed
                    if (std::fabs(bv) < 6.103515625e-05) flush += std::fabs(p);
#endif
                }
                double err = std::fabs((double)out[((size_t)t * 32 + i) * 32 + j] - exact);
                if (flush > 0.0) {
                    flushed_seen = flushed_seen || err > 64.0 * sabs * 5.9604644775390625e-08;
                    err = err > flush ? err - flush : 0.0;   // what the flush allowance does not cover must fit the budget
                }
                const double ratio = sabs > 0 ? err / (sabs * 5.9604644775390625e-08) : 0.0;
                if (ratio > worst[t % 4]) worst[t % 4] = ratio;
                if (err > 1e-3 * sabs + 1e-30) ++wrong_layout;
            }
    std::printf("max |mfma - exact| / (u * sum|a b|): data-like %.3f  wide-range %.3f  cancellation %.3f  tile-like %.3f  (budget 256)\n",
                worst[0], worst[1], worst[2], worst[3]);
    if (flushed_seen) std::printf("note: operands below 2^-14 were flushed by the matrix pipe (covered by the bound's flush allowance)\n");
    if (wrong_layout) { std::printf("FAIL: %zu outputs far from the exact product (fragment layout?)\n", wrong_layout); return 1; }
    for (double w : worst) if (w > 64.0) { std::printf("FAIL: accumulation error above a quarter of the budget\n"); return 1; }
    std::printf("OK\n");
    return 0;
}
