// Checks the operand layout the INT8 filter's 16x16 tile format assumes for v_mfma_i32_16x16x64_i8 (gfx950) and that a
// chain of 2 k-steps (K = 128) started from a per-row accumulator init is the EXACT integer
// n[row] + sum_k a[row][k] b[k][col]:
//   lane l holds A[l & 15][16 (l >> 4) + 0..15] and B[16 (l >> 4) + 0..15][l & 15] as 16 packed int8 (per 64-wide k-step),
//   accumulator i of lane l is C[4 (l >> 4) + i][l & 15].
// Prints OK / exit code 0 when every one of the 256 random 16x16 blocks matches the host integers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void chain(const uint4* __restrict__ afrag, const uint4* __restrict__ bfrag, const int* __restrict__ nrm,
                      int* __restrict__ out)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned t = blockIdx.x;
    union { uint4 u; i32x4 v; } a, b;
    i32x4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = nrm[t * 16 + 4 * (lane >> 4) + i];
    for (int ks = 0; ks < 2; ++ks) {
        a.u = afrag[(t * 2 + ks) * 64 + lane];
        b.u = bfrag[(t * 2 + ks) * 64 + lane];
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.v, b.v, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) out[(t * 16 + 4 * (lane >> 4) + i) * 16 + (lane & 15u)] = acc[i];
}

int main()
{
    const int T = 256;
    std::vector<signed char> A((size_t)T * 16 * 128), B((size_t)T * 128 * 16);
    std::vector<int> N((size_t)T * 16);
    srand(13);
    for (auto& v : A) v = (signed char)(rand() % 255 - 127);
    for (auto& v : B) v = (signed char)(rand() % 255 - 127);
    for (auto& v : N) v = -(rand() % 900000);
    std::vector<unsigned char> af((size_t)T * 2 * 64 * 16), bf((size_t)T * 2 * 64 * 16);
    for (int t = 0; t < T; ++t)
        for (int ks = 0; ks < 2; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 16; ++j) {
                    const int k = 64 * ks + 16 * (l >> 4) + j;
                    af[(((size_t)t * 2 + ks) * 64 + l) * 16 + j] = (unsigned char)A[((size_t)t * 16 + (l & 15)) * 128 + k];
                    bf[(((size_t)t * 2 + ks) * 64 + l) * 16 + j] = (unsigned char)B[((size_t)t * 128 + k) * 16 + (l & 15)];
                }
    uint4 *da, *db; int *dn, *dout;
    hipMalloc(&da, af.size()); hipMalloc(&db, bf.size()); hipMalloc(&dn, N.size() * 4); hipMalloc(&dout, (size_t)T * 256 * 4);
    hipMemcpy(da, af.data(), af.size(), hipMemcpyHostToDevice);
    hipMemcpy(db, bf.data(), bf.size(), hipMemcpyHostToDevice);
    hipMemcpy(dn, N.data(), N.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain, dim3(T), dim3(64), 0, 0, da, db, dn, dout);
    std::vector<int> out((size_t)T * 256);
    if (hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("HIP error\n"); return 2; }
    size_t bad = 0;
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                long long s = N[(size_t)t * 16 + i];
                for (int k = 0; k < 128; ++k) s += (int)A[((size_t)t * 16 + i) * 128 + k] * (int)B[((size_t)t * 128 + k) * 16 + j];
                if (s != out[((size_t)t * 16 + i) * 16 + j]) ++bad;
            }
    std::printf("mismatches: %zu of %zu\n%s\n", bad, out.size(), bad ? "FAIL" : "OK");
    return bad ? 1 : 0;
}
