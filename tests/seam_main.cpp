// seam_main.cpp -- the translation unit a maintainer of the reference gets when src/test.cpp:6-13 grows an
// `#elif IMPL == 4` branch that includes this repo's include/hvs_vec_query.hpp instead of one of the reference's
// engine headers: read D and Q in the io.h formats (reference include/io.h:111-136), call
//     vec_query(nodes, queries, sample_proportion, knn_results)            (src/test.cpp:85)
// with the reference's exact signature, write output.bin (io.h:23-36).  Own text; the only thing shared with the
// reference is the seam itself.  Built by __graft_entry__.build(), run by tests/test_gpu_parity.py.
//
//   seam_main.out <source_path> <query_path> <output_path> [preseed]
// `preseed` rows are put into knn_results before the call: the reference appends (push_back,
// optimized_parallel.hpp:159), it does not clear -- the program checks that they are still there, untouched, and
// writes only the appended rows.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "hvs_vec_query.hpp"

static bool read_rows(const std::string& path, uint32_t cols, std::vector<std::vector<float>>& rows)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    uint32_t n = 0;
    f.read(reinterpret_cast<char*>(&n), sizeof(n));
    rows.assign(n, std::vector<float>(cols));
    for (uint32_t i = 0; i < n; ++i)
        if (!f.read(reinterpret_cast<char*>(rows[i].data()), (std::streamsize)(cols * sizeof(float)))) {
            rows.resize(i);
            break;
        }
    return true;
}

int main(int argc, char** argv)
{
    if (argc < 4 || argc > 5) {
        std::cout << argv[0] << " source_path query_path output_path [preseed]\n";
        return 1;
    }
    const uint32_t preseed = argc == 5 ? (uint32_t)std::atoi(argv[4]) : 0u;
    std::vector<std::vector<float>> nodes, queries;
    if (!read_rows(argv[1], 102, nodes) || !read_rows(argv[2], 104, queries)) {
        std::cerr << "cannot read the inputs\n";
        return 2;
    }
    std::vector<std::vector<uint32_t>> knn_results;
    for (uint32_t i = 0; i < preseed; ++i) knn_results.push_back(std::vector<uint32_t>{i, 7u, 9u});
    const float sample_proportion = 1.0f;  // src/test.cpp:68
    try {
        vec_query(nodes, queries, sample_proportion, knn_results);
    } catch (const std::exception& e) {
        std::cerr << "vec_query failed: " << e.what() << "\n";
        return 3;
    }
    if (knn_results.size() != (size_t)preseed + queries.size()) {
        std::cerr << "vec_query did not append one row per query\n";
        return 4;
    }
    for (uint32_t i = 0; i < preseed; ++i)
        if (knn_results[i] != std::vector<uint32_t>{i, 7u, 9u}) {
            std::cerr << "vec_query touched rows that were already in knn_results\n";
            return 4;
        }
    std::ofstream out(argv[3], std::ios::binary);
    for (size_t i = preseed; i < knn_results.size(); ++i) {
        if (knn_results[i].size() != 100) {
            std::cerr << "row " << i << " does not hold 100 ids\n";
            return 4;
        }
        out.write(reinterpret_cast<const char*>(knn_results[i].data()), 100 * sizeof(uint32_t));
    }
    std::cerr << "seam ok: " << queries.size() << " queries appended after " << preseed << " existing rows\n";
    return out.good() ? 0 : 2;
}
