// Checks the operand layout the INT8 filter assumes for v_mfma_i32_32x32x32_i8 (gfx950) and that a chain of
// 4 k-steps (K = 128) started from a per-row accumulator init is the EXACT integer  n[row] + sum_k a[row][k] b[k][col]:
//   lane l holds A[l & 31][16 (l >> 5) + 0..15] and B[16 (l >> 5) + 0..15][l & 31] as 16 packed int8,
//   accumulator r of lane l is C[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
// Prints OK / exit code 0 when every one of the 256 random tiles matches the host integers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__global__ void chain(const uint4* __restrict__ afrag, const uint4* __restrict__ bfrag, const int* __restrict__ nrm,
                      int* __restrict__ out)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned t = blockIdx.x;
    union { uint4 u; i32x4 v; } a, b;
    i32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = nrm[t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
    for (int ks = 0; ks < 4; ++ks) {
        a.u = afrag[(t * 4 + ks) * 64 + lane];
        b.u = bfrag[(t * 4 + ks) * 64 + lane];
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const unsigned row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31u;
        out[(t * 32 + row) * 32 + col] = acc[r];
    }
}

int main()
{
    const int T = 256;
    std::vector<signed char> A((size_t)T * 32 * 128), B((size_t)T * 128 * 32);
    std::vector<int> N((size_t)T * 32);
    srand(11);
    for (auto& v : A) v = (signed char)(rand() % 255 - 127);
    for (auto& v : B) v = (signed char)(rand() % 255 - 127);
    for (auto& v : N) v = -(rand() % 900000);
    std::vector<unsigned char> af((size_t)T * 4 * 64 * 16), bf((size_t)T * 4 * 64 * 16);
    for (int t = 0; t < T; ++t)
        for (int ks = 0; ks < 4; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 16; ++j) {
                    const int k = 32 * ks + 16 * (l >> 5) + j;
                    af[(((size_t)t * 4 + ks) * 64 + l) * 16 + j] = (unsigned char)A[((size_t)t * 32 + (l & 31)) * 128 + k];
                    bf[(((size_t)t * 4 + ks) * 64 + l) * 16 + j] = (unsigned char)B[((size_t)t * 128 + k) * 32 + (l & 31)];
                }
    uint4 *da, *db; int *dn, *dout;
    hipMalloc(&da, af.size()); hipMalloc(&db, bf.size()); hipMalloc(&dn, N.size() * 4); hipMalloc(&dout, (size_t)T * 1024 * 4);
    hipMemcpy(da, af.data(), af.size(), hipMemcpyHostToDevice);
    hipMemcpy(db, bf.data(), bf.size(), hipMemcpyHostToDevice);
    hipMemcpy(dn, N.data(), N.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain, dim3(T), dim3(64), 0, 0, da, db, dn, dout);
    std::vector<int> out((size_t)T * 1024);
    if (hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("HIP error\n"); return 2; }
    size_t bad = 0;
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                long long s = N[(size_t)t * 32 + i];
                for (int k = 0; k < 128; ++k) s += (int)A[((size_t)t * 32 + i) * 128 + k] * (int)B[((size_t)t * 128 + k) * 32 + j];
                if (s != out[((size_t)t * 32 + i) * 32 + j]) ++bad;
            }
    std::printf("mismatches: %zu of %zu\n%s\n", bad, out.size(), bad ? "FAIL" : "OK");
    return bad ? 1 : 0;
}
