// hvs_vec_query.hpp -- header-only C++ shim that restores the reference's source-level seam
// on top of the C ABI (include/hvs.h).
//
// The reference selects its engine by including one header that defines
//     void vec_query(vector<vector<float>>& nodes, vector<vector<float>>& queries,
//                    float sample_proportion, vector<vector<uint32_t>>& knn_results);
// (include/baseline.hpp:68-69, include/optimized.hpp:54-55, include/optimized_parallel.hpp:61-62;
// chosen by -DIMPL in src/test.cpp:6-13).  A maintainer adds `#elif IMPL == 4` +
// `#include "hvs_vec_query.hpp"` there and links libhvs.so; nothing else in test.cpp changes
// (tests/seam_main.cpp is that translation unit, built and run by the test-suite).
//
// Semantics kept: results are APPENDED to knn_results (optimized_parallel.hpp:159 push_back's),
// the three size lines go to stdout (optimized_parallel.hpp:69-71), nothing is returned, and -- like the
// reference, which sizes its own thread pool from the machine (optimized_parallel.hpp:73-78) -- the call uses
// every GPU of the node: one per 32768 queries, at most all (HVS_GPUS=n overrides).  Where the reference has
// undefined behaviour (n < 100, short rows) this shim throws std::runtime_error.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "hvs.h"

namespace hvs_detail {
inline void check(int rc, hvs_ctx* ctx, const char* what)
{
    if (rc == HVS_OK) return;
    std::string msg = std::string(what) + ": " + (ctx ? hvs_last_error(ctx) : hvs_last_global_error());
    if (ctx) hvs_destroy(ctx);
    throw std::runtime_error(msg);
}
inline std::vector<float> flatten(const std::vector<std::vector<float>>& rows, size_t cols, const char* what)
{
    std::vector<float> flat(rows.size() * cols);
    for (size_t i = 0; i < rows.size(); ++i) {
        if (rows[i].size() < cols) throw std::runtime_error(std::string(what) + ": row shorter than expected");
        std::copy(rows[i].begin(), rows[i].begin() + (std::ptrdiff_t)cols, flat.begin() + (std::ptrdiff_t)(i * cols));
    }
    return flat;
}
// GPUs a call of nq queries is spread over
inline int gpus_for(uint32_t nq)
{
    int have = hvs_device_count();
    if (have < 1) have = 1;  // hvs_create_multi reports the missing GPU
    if (const char* e = std::getenv("HVS_GPUS")) {
        const int want = std::atoi(e);
        if (want >= 1) return std::min(want, have);
    }
    return (int)std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)have, nq / 32768u));
}
}  // namespace hvs_detail

inline void vec_query(std::vector<std::vector<float>>& nodes, std::vector<std::vector<float>>& queries,
                      float sample_proportion, std::vector<std::vector<uint32_t>>& knn_results)
{
    const uint32_t n = (uint32_t)nodes.size();
    const uint32_t d = n ? (uint32_t)nodes[0].size() : 0u;
    const uint32_t nq = (uint32_t)queries.size();
    std::cout << "# data points:  " << n << "\n";
    std::cout << "# data point dim:  " << d << "\n";
    std::cout << "# queries:      " << nq << "\n";
    const std::vector<float> D = hvs_detail::flatten(nodes, 102, "vec_query(nodes)");
    const std::vector<float> Q = hvs_detail::flatten(queries, 104, "vec_query(queries)");
    hvs_ctx* ctx = nullptr;
    hvs_detail::check(hvs_create_multi(&ctx, hvs_detail::gpus_for(nq)), nullptr, "hvs_create_multi");
    hvs_detail::check(hvs_reserve(ctx, nq), ctx, "hvs_reserve");
    hvs_detail::check(hvs_load_data(ctx, D.data(), n), ctx, "hvs_load_data");
    std::vector<uint32_t> ids((size_t)nq * 100);
    if (nq) hvs_detail::check(hvs_query(ctx, Q.data(), nq, sample_proportion, ids.data(), nullptr), ctx, "hvs_query");
    hvs_destroy(ctx);
    for (uint32_t i = 0; i < nq; ++i)
        knn_results.emplace_back(ids.begin() + (std::ptrdiff_t)((size_t)i * 100), ids.begin() + (std::ptrdiff_t)((size_t)(i + 1) * 100));
}
