// hvs_compare.cpp -- result checker with the reference's command line (reference src/compare_data.cpp):
//     hvs_compare.out a.bin b.bin [c.bin ...]
// For every unordered pair of arguments it opens <arg>.dist (uint32 nq, then nq x 100 f32;
// include/io.h:50-105) and compares the distances position by position with the reference's absolute
// tolerance 0.002 (compare_data.cpp:5,40-61), printing the reference's three verdict lines
// (compare_data.cpp:64-77).  Like the reference it never fails the process for a difference.
//
// Extension: with  --data D.bin --queries Q.bin  the id files <arg> themselves
// (output.bin: nq x 100 uint32) are compared tie-aware: the exact-order distance sequences must be
// identical bit for bit and ids may differ only inside groups of equal distance.  With --strict the
// exit code is 1 when any pair fails that check.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

static const double error_delta = 0.002;

static bool read_dist(const std::string& path, std::vector<float>& v, uint32_t& nq)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    if (std::fread(&nq, 4, 1, f) != 1) { std::fclose(f); return false; }
    v.resize((size_t)nq * 100);
    const size_t got = std::fread(v.data(), 4, v.size(), f);
    std::fclose(f);
    return got == v.size();
}

static bool read_rows(const std::string& path, uint32_t cols, std::vector<float>& v, uint32_t& n)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    if (std::fread(&n, 4, 1, f) != 1) { std::fclose(f); return false; }
    v.resize((size_t)n * cols);
    const size_t got = std::fread(v.data(), 4 * cols, n, f);
    std::fclose(f);
    n = (uint32_t)got;
    return true;
}

static bool read_ids(const std::string& path, std::vector<uint32_t>& v)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    v.resize((size_t)sz / 4);
    const size_t got = std::fread(v.data(), 4, v.size(), f);
    std::fclose(f);
    return got == v.size() && v.size() % 100 == 0;
}

// exact-order distance of the hot path (optimized_impl.h:96-125 + hsum :37-47); built with -ffp-contract=off
static float exact_dist(const float* d, const float* q)
{
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < 12; ++b)
        for (int j = 0; j < 8; ++j) {
            float t = d[8 * b + j] - q[8 * b + j];
            t = t * t;
            acc[j] = acc[j] + t;
        }
    for (int j = 4; j < 8; ++j) {
        float t = d[92 + j] - q[92 + j];
        t = t * t;
        acc[j] = acc[j] + t;
    }
    const float s0 = acc[0] + acc[4], s1 = acc[1] + acc[5], s2 = acc[2] + acc[6], s3 = acc[3] + acc[7];
    const float a = s0 + s1, b2 = s2 + s3;
    return a + b2;
}

// What the reference's checker reports about two .dist files (src/compare_data.cpp:5-78): whether all distances
// agree exactly, agree within the absolute tolerance, or how many do not.  The verdict lines are the CLI contract
// (run.sh greps nothing, people read them); the bookkeeping behind them is this file's own.
struct DistVerdict {
    double worst = 0.0;               // largest |a - b|
    uint64_t beyond = 0;              // entries with |a - b| >= error_delta
    std::vector<size_t> examples;     // the first few of them (flat index)
};

static DistVerdict judge_dists(const std::vector<float>& a, const std::vector<float>& b)
{
    DistVerdict v;
    for (size_t i = 0; i < a.size(); ++i) {
        const double gap = std::fabs((double)a[i] - (double)b[i]);
        v.worst = std::max(v.worst, gap);
        if (gap >= error_delta) {
            if (v.examples.size() < 49) v.examples.push_back(i);
            ++v.beyond;
        }
    }
    return v;
}

static void compare_dist(const std::string& a_path, const std::string& b_path)
{
    std::cout << "\nComparing: " << a_path << " " << b_path << std::endl;
    std::vector<float> a, b;
    uint32_t na = 0, nb = 0;
    if (!read_dist(a_path, a, na) || !read_dist(b_path, b, nb)) {
        std::cerr << "cannot read " << a_path << " / " << b_path << std::endl;
        return;
    }
    if (na != nb || a.size() != b.size()) {
        std::cerr << "Datasets have different number of queries! " << na << ", " << nb << std::endl;
        return;
    }
    const size_t per_query = na ? a.size() / na : 1;  // k of the files (100 unless written with another k)
    const DistVerdict v = judge_dists(a, b);
    for (size_t i : v.examples)
        std::cerr << i / per_query << " - " << i % per_query << ": distance difference of " << std::fabs((double)a[i] - (double)b[i])
                  << " between " << std::setprecision(15) << a[i] << " and " << b[i] << std::endl;
    if (v.worst == 0.0) {
        std::cout << "Datasets are the same!" << std::endl;
        return;
    }
    if (v.beyond == 0)
        std::cout << "Datasets are similar under error delta!" << std::endl;
    else
        std::cout << "ERROR: Found a total of " << v.beyond << " differences!" << std::endl;
    std::cout << "Max Floating Point Error Difference: " << std::setprecision(15) << v.worst << std::endl;
}

// returns number of queries violating the tie-aware rule
static uint32_t compare_ids(const std::string& a_path, const std::string& b_path, const std::vector<float>& D, uint32_t n,
                            const std::vector<float>& Q, uint32_t nq)
{
    std::vector<uint32_t> a, b;
    if (!read_ids(a_path, a) || !read_ids(b_path, b) || a.size() != b.size() || a.size() != (size_t)nq * 100) {
        std::cerr << "id files unreadable or of different shape: " << a_path << " " << b_path << std::endl;
        return nq;
    }
    uint32_t identical = 0, tie_only = 0, bad = 0;
    std::vector<std::pair<uint32_t, uint32_t>> ka(100), kb(100);  // (dist bits, id)
    for (uint32_t i = 0; i < nq; ++i) {
        const uint32_t *ra = &a[(size_t)i * 100], *rb = &b[(size_t)i * 100];
        if (std::equal(ra, ra + 100, rb)) { ++identical; continue; }
        bool ok = true;
        for (int k = 0; k < 100 && ok; ++k) {
            if (ra[k] >= n || rb[k] >= n) { ok = false; break; }
            float da = exact_dist(&D[(size_t)ra[k] * 102 + 2], &Q[(size_t)i * 104 + 4]);
            float db = exact_dist(&D[(size_t)rb[k] * 102 + 2], &Q[(size_t)i * 104 + 4]);
            uint32_t ua, ub;
            std::memcpy(&ua, &da, 4);
            std::memcpy(&ub, &db, 4);
            ka[k] = {ua, ra[k]};
            kb[k] = {ub, rb[k]};
        }
        if (ok) {
            std::sort(ka.begin(), ka.end());
            std::sort(kb.begin(), kb.end());
            const uint32_t kth = ka[99].first;
            for (int k = 0; k < 100 && ok; ++k) {
                if (ka[k].first != kb[k].first) ok = false;                              // distance multisets differ
                else if (ka[k].second != kb[k].second && ka[k].first != kth) {
                    // below the k-th distance the id multiset per distance must match; sorted by (dist,id) it must be equal
                    ok = false;
                }
            }
        }
        if (ok) ++tie_only; else ++bad;
    }
    std::cout << "ids " << a_path << " vs " << b_path << ": " << identical << " identical, " << tie_only
              << " differ only inside the k-th-distance tie group, " << bad << " VIOLATIONS" << std::endl;
    return bad;
}

int main(int argc, char* argv[])
{
    std::string data_path, query_path;
    bool strict = false;
    std::vector<std::string> outs;
    for (int i = 1; i < argc; ++i) {
        const std::string s = argv[i];
        if (s == "--data" && i + 1 < argc) data_path = argv[++i];
        else if (s == "--queries" && i + 1 < argc) query_path = argv[++i];
        else if (s == "--strict") strict = true;
        else outs.push_back(s);
    }
    for (size_t i = 0; i < outs.size(); ++i)
        for (size_t j = i + 1; j < outs.size(); ++j) compare_dist(outs[i] + ".dist", outs[j] + ".dist");
    uint32_t bad = 0;
    if (!data_path.empty() && !query_path.empty()) {
        std::vector<float> D, Q;
        uint32_t n = 0, nq = 0;
        if (!read_rows(data_path, 102, D, n) || !read_rows(query_path, 104, Q, nq)) {
            std::cerr << "cannot read " << data_path << " / " << query_path << std::endl;
            return 2;
        }
        for (size_t i = 0; i < outs.size(); ++i)
            for (size_t j = i + 1; j < outs.size(); ++j) bad += compare_ids(outs[i], outs[j], D, n, Q, nq);
    }
    return (strict && bad) ? 1 : 0;
}
