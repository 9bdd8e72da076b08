// Can a memory-gather kernel run UNDER a persistent matrix kernel on the same CUs?  The matrix kernel is the filter-shaped loop of
// mfma_shape_lab.hip (two workgroups per CU, 66 KiB of LDS and ~160 registers each: 192 registers per lane and ~28 KiB of LDS stay
// free on every CU); the gather kernel reads random 512-byte rows (4 lanes x ... like the re-score kernel) with <= 64 registers and no LDS.
// Streams with priorities: matrix kernel on the high-priority stream.  Prints each kernel alone and both together.
//   hipcc --offload-arch=gfx950 -O3 scripts/coexist_lab.hip -o scripts/coexist_lab.out
#define main shape_lab_main
#include "mfma_shape_lab.hip"
#undef main

__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ buf, size_t nrows, int* __restrict__ out, int per_thread, unsigned seed)
{
    // 4 lanes per row of 512 B: lane t reads 8 x 16 B at stride 64 B (whole 128-byte lines per 4 lanes over two instructions)
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    unsigned long long st = (gid >> 2) * 0x9E3779B97F4A7C15ull + seed;
    const unsigned t4 = threadIdx.x & 3u;
    unsigned acc = 0;
    for (int i = 0; i < per_thread; ++i) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const size_t row = (size_t)((st >> 20) % nrows);
        const uint4* r = buf + row * 32u;  // 512 B = 32 uint4
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint4 v = r[4 * b + t4];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345u) out[gid & 1023u] = (int)acc;
}

int main()
{
    std::vector<unsigned> h(20 * 64 * 4);
    unsigned long long st = 12345;
    for (auto& x : h) { st = st * 6364136223846793005ull + 1442695040888963407ull; x = (unsigned)(st >> 32); }
    hipMalloc(&g_in, h.size() * 4);
    hipMalloc(&g_out, 4096 * 256 * 4);
    hipMemcpy(g_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const size_t nrows = 8u << 20;  // 4 GiB of rows
    uint4* rows;
    hipMalloc(&rows, nrows * 512);
    hipMemset(rows, 1, nrows * 512);
    int lo, hi;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t s_hi, s_lo;
    hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, hi);
    hipStreamCreateWithPriority(&s_lo, hipStreamNonBlocking, lo);
    std::printf("stream priorities: lowest %d highest %d\n", lo, hi);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto big = [&](hipStream_t s) { hipLaunchKernelGGL((loop<1, 2>), dim3(2048), dim3(256), 0, s, g_in, g_out, g_tiles, 0x7fffff00); };
    const int gblocks = 1 << 16, per_thread = 8;  // 2^16 x 64 rows x 8 = 3.4e7 rows = 17 GB per launch
    auto small = [&](hipStream_t s, int i) { hipLaunchKernelGGL(gather, dim3(gblocks), dim3(256), 0, s, rows, nrows, g_out, per_thread, 77u + i); };
    auto wall = [&](auto&& body) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        body();
        hipDeviceSynchronize();
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    const int reps = 10;
    for (int round = 0; round < 2; ++round) {
        const double tb = wall([&] { for (int i = 0; i < reps; ++i) big(s_hi); });
        const double ts = wall([&] { for (int i = 0; i < reps; ++i) small(s_lo, i); });
        const double both = wall([&] { for (int i = 0; i < reps; ++i) { big(s_hi); small(s_lo, i); } });
        const double both_rev = wall([&] { for (int i = 0; i < reps; ++i) { small(s_lo, i); big(s_hi); } });
        const double same = wall([&] { for (int i = 0; i < reps; ++i) { big(s_hi); small(s_hi, i); } });
        std::printf("%d x matrix loop alone %.1f ms | %d x gather alone %.1f ms (%.2f TB/s) | together, two streams %.1f ms (gather launched first: %.1f ms) | one stream %.1f ms\n",
                    reps, tb, reps, ts, reps * (double)gblocks * 64 * per_thread * 512 / (ts * 1e-3) / 1e12, both, both_rev, same);
    }
    return 0;
}
