// hvs_main.cpp -- command-line driver with the reference's process-level contract
// (reference src/test.cpp:20-111):
//     hvs_search.out [source_path] [query_path] [output_path]
// defaults ../data/default-data.bin, ../data/query.bin, ../result_data/hvs.bin; any other argc
// prints the usage line and exits 1 (test.cpp:51-64).  Reads D / Q in the io.h format
// (io.h:111-136) with one bulk read each, times only the vec_query-equivalent region
// (test.cpp:82-88: data already in host memory -> ids back in host memory), prints
// "Vector Search took <ms> ms" on stderr (test.cpp:91-92), writes output.bin (io.h:23-36) and
// <output>.dist with the scalar-order distances of the chosen rows (test.cpp:97-110, io.h:38-78).
// k: the reference's KNN_LIMIT is a compile-time 100 (optimized_impl.h:26); HVS_K=<8..256> runs the driver with another k
// (rows of output.bin / .dist then hold k entries).
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hvs.h"

static bool read_bin(const std::string& path, uint32_t cols, std::vector<float>& rows, uint32_t& n)
{
    std::cout << "Reading Data: " << path << std::endl;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    uint32_t hdr = 0;
    if (std::fread(&hdr, sizeof(hdr), 1, f) != 1) {
        std::fclose(f);
        return false;
    }
    std::cout << "# of points: " << hdr << std::endl;
    rows.resize((size_t)hdr * cols);
    const size_t got = std::fread(rows.data(), sizeof(float) * cols, hdr, f);
    std::fclose(f);
    n = (uint32_t)got;  // like io.h:125 only whole rows actually present are used
    rows.resize((size_t)n * cols);
    std::cout << "Finish Reading Data" << std::endl;
    return true;
}

int main(int argc, char** argv)
{
    std::cout << "Running MI355X Vector Search\n";
    std::string source_path = "../data/default-data.bin";
    std::string query_path = "../data/query.bin";
    std::string knn_save_path = "../result_data/hvs.bin";
    switch (argc) {
    case 4: knn_save_path = argv[3]; [[fallthrough]];
    case 3: query_path = argv[2]; [[fallthrough]];
    case 2: source_path = argv[1]; [[fallthrough]];
    case 1: break;
    default: std::cout << argv[0] << " [source_path] [query_path] [output_path]\n"; return 1;
    }
    const float sample_proportion = 1.0f;  // test.cpp:68

    std::vector<float> nodes, queries;
    uint32_t n = 0, nq = 0;
    if (!read_bin(source_path, 102, nodes, n)) {
        std::cerr << "cannot read " << source_path << "\n";
        return 2;
    }
    std::cout << n << "\n";
    if (!read_bin(query_path, 104, queries, nq)) {
        std::cerr << "cannot read " << query_path << "\n";
        return 2;
    }

    // like the reference, which sizes its own worker pool from the machine (optimized_parallel.hpp:73-78), the driver
    // uses every GPU of the node: one per 32768 queries, at most all; HVS_GPUS=n overrides
    int gpus = hvs_device_count();
    if (gpus < 1) gpus = 1;
    int use = (int)std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)gpus, nq / 32768u));
    if (const char* e = std::getenv("HVS_GPUS"))
        if (std::atoi(e) >= 1) use = std::min(std::atoi(e), gpus);
    hvs_ctx* ctx = nullptr;
    if (hvs_create_multi(&ctx, use) != HVS_OK) {
        std::cerr << hvs_last_global_error() << "\n";
        return 3;
    }
    uint32_t K = 100;
    if (const char* e = std::getenv("HVS_K")) {
        if (hvs_set_k(ctx, (uint32_t)std::atoi(e)) != HVS_OK) {
            std::cerr << "HVS_K: " << hvs_last_error(ctx) << "\n";
            hvs_destroy(ctx);
            return 3;
        }
        K = hvs_get_k(ctx);
    }
    (void)hvs_reserve(ctx, nq);
    std::vector<uint32_t> ids((size_t)nq * K);
    std::cout << "# data points:  " << n << "\n# data point dim:  102\n# queries:      " << nq << "\n";
    const auto t0 = std::chrono::steady_clock::now();
    int rc = hvs_load_data(ctx, nodes.data(), n);
    if (rc == HVS_OK && nq) rc = hvs_query(ctx, queries.data(), nq, sample_proportion, ids.data(), nullptr);
    const auto t1 = std::chrono::steady_clock::now();
    if (rc != HVS_OK) {
        std::cerr << "hvs error " << rc << ": " << hvs_last_error(ctx) << "\n";
        hvs_destroy(ctx);
        return 3;
    }
    hvs_timing tm{};
    if (nq) hvs_last_timing(ctx, &tm);
    std::cerr << "Vector Search took " << std::chrono::duration<double, std::milli>(t1 - t0).count() << " ms"
              << " (" << hvs_num_gpus(ctx) << " GPU(s): query host->host " << tm.host_ms << " ms, of it on the device " << tm.query_ms
              << " ms; data upload+index " << tm.load_ms << " ms)" << std::endl;
    hvs_destroy(ctx);

    FILE* f = std::fopen(knn_save_path.c_str(), "wb");
    if (!f) {
        std::cerr << "cannot write " << knn_save_path << "\n";
        return 2;
    }
    std::fwrite(ids.data(), sizeof(uint32_t), ids.size(), f);
    std::fclose(f);

    // .dist: one scalar-order distance per (query, neighbour) -- nq x k rows of D fetched at random (io.h:50-78 walks them
    // one by one; at BASELINE configs[3] size that is 4 x 10^8 rows of 408 bytes, tens of seconds on one core): the
    // queries are cut over the machine's cores, the file is written in order from the filled array
    std::vector<float> dist((size_t)nq * K);
    {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const unsigned nth = (unsigned)std::max<uint64_t>(1u, std::min<uint64_t>(hw, (uint64_t)nq / 256u));
        auto work = [&](unsigned t) {
            const uint32_t q0 = (uint32_t)((uint64_t)nq * t / nth), q1 = (uint32_t)((uint64_t)nq * (t + 1u) / nth);
            for (uint32_t i = q0; i < q1; ++i) {
                const float* b = &queries[(size_t)i * 104 + 4];
                for (uint32_t k = 0; k < K; ++k) {
                    const float* a = &nodes[(size_t)ids[(size_t)i * K + k] * 102];
                    float sum = 0.0f;  // io.h:38-48 calc_dist: sequential order
                    for (int x = 0; x < 100; ++x) {
                        float diff = a[2 + x] - b[x];
                        diff = diff * diff;
                        sum = sum + diff;
                    }
                    dist[(size_t)i * K + k] = sum;
                }
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nth; ++t) th.emplace_back(work, t);
        work(0u);
        for (auto& t : th) t.join();
    }
    f = std::fopen((knn_save_path + ".dist").c_str(), "wb");
    if (!f) return 2;
    std::fwrite(&nq, sizeof(uint32_t), 1, f);
    std::fwrite(dist.data(), sizeof(float), dist.size(), f);
    std::fclose(f);
    return 0;
}
