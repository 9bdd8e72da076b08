// Host-only check of the level-interleaved block order (csrc/hvs_filter.h): every block is stored
// exactly once, and for any block range the per-level storage runs cover exactly that range; and of the work-item
// segmentation (hvs_make_segs): per level the segments tile the level's storage run exactly once, their size is a
// power of two in 8..HVS_SEG that shrinks for small batches, and the global segment numbering is gap-free.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../project---hybrid-vector-search-queries_amd/csrc/hvs_filter.h"

int main()
{
    const uint32_t sizes[] = {100, 2047, 2048 * 32, 10000, 70000, 300001, 1000000, 10000000, 33554432};
    const uint32_t radices[][2] = {{HVS_RADIX_LAST, HVS_RADIX_MID}, {2, 2}, {8, 16}, {16, 16}, {2, 4}, {64, 32}};  // {last, mid}
    for (const auto& rr : radices)
    for (uint32_t n : sizes) {
        const HvsLevels L = hvs_make_levels(n, rr[0], rr[1]);
        if (!L.pow2) { std::printf("FAIL n=%u: power-of-two radices expected\n", n); return 1; }
        if (L.K && L.radix[L.K] > rr[0]) { std::printf("FAIL n=%u: last radix %u > %u\n", n, L.radix[L.K], rr[0]); return 1; }
        if (L.off[1] - L.off[0] < 16u && L.nblk >= 16u) { std::printf("FAIL n=%u: level 0 has %u blocks\n", n, L.off[1] - L.off[0]); return 1; }
        if (L.off[L.K + 1] != L.nblk) { std::printf("FAIL n=%u: levels cover %u of %u blocks\n", n, L.off[L.K + 1], L.nblk); return 1; }
        std::vector<uint32_t> where(L.nblk, 0xFFFFFFFFu);
        for (uint32_t idx = 0; idx < L.nblk; ++idx) {
            const uint32_t b = hvs_storage_to_block(L, idx);
            if (b >= L.nblk || where[b] != 0xFFFFFFFFu) { std::printf("FAIL n=%u: storage %u -> block %u\n", n, idx, b); return 1; }
            where[b] = idx;
        }
        srand(n);
        for (int trial = 0; trial < 200; ++trial) {
            uint32_t blo = (uint32_t)rand() % L.nblk, bhi = (uint32_t)rand() % (L.nblk + 1);
            if (trial == 0) { blo = 0; bhi = L.nblk; }
            if (blo > bhi) { uint32_t t = blo; blo = bhi; bhi = t; }
            uint64_t covered = 0;
            for (uint32_t j = 0; j <= L.K; ++j) {
                uint32_t lo, hi;
                hvs_level_run(L, j, blo, bhi, lo, hi);
                if (lo > hi || (lo < hi && (lo < L.off[j] || hi > L.off[j + 1]))) { std::printf("FAIL n=%u level %u run [%u,%u)\n", n, j, lo, hi); return 1; }
                for (uint32_t i = lo; i < hi; ++i) {
                    const uint32_t b = hvs_storage_to_block(L, i);
                    if (b < blo || b >= bhi) { std::printf("FAIL n=%u: level %u run leaves [%u,%u): block %u\n", n, j, blo, bhi, b); return 1; }
                }
                covered += hi - lo;
            }
            if (covered != (uint64_t)(bhi - blo)) { std::printf("FAIL n=%u: [%u,%u) covered %llu\n", n, blo, bhi, (unsigned long long)covered); return 1; }
        }
        // rows of a position range that lie in the levels before `level` (the seen fraction behind a guessed threshold)
        for (int trial = 0; trial < 300; ++trial) {
            uint32_t a = (uint32_t)rand() % n, len = trial % 3 == 0 ? (uint32_t)rand() % 200u : (uint32_t)rand() % n;
            uint32_t b = a + len > n ? n : a + len;
            if (trial == 0) { a = 0; b = n; }
            for (uint32_t level = 0; level <= L.K + 1u; ++level) {
                uint64_t want = 0;
                if (b > a && (b - a) < 200000u) {
                    for (uint32_t pos = a; pos < b; ++pos) want += hvs_block_level(L, pos / 32u) < level ? 1u : 0u;
                } else if (b > a) {
                    for (uint32_t blk = a / 32u; blk <= (b - 1u) / 32u; ++blk) {
                        if (hvs_block_level(L, blk) >= level) continue;
                        const uint32_t lo = blk * 32u > a ? blk * 32u : a, hi = (blk + 1u) * 32u < b ? (blk + 1u) * 32u : b;
                        want += hi - lo;
                    }
                }
                const uint32_t got = hvs_rows_seen_before(L, level, a, b);
                if (got != want) { std::printf("FAIL n=%u [%u,%u) level %u: seen %u want %llu\n", n, a, b, level, got, (unsigned long long)want); return 1; }
                if (level == L.K + 1u && got != b - a) { std::printf("FAIL n=%u: all levels do not cover the range\n", n); return 1; }
            }
        }
        for (uint32_t idx = 0; idx < L.nblk; idx += 1u + L.nblk / 5000u) {
            uint32_t j = 0;
            while (j < L.K && idx >= L.off[j + 1]) ++j;
            if (hvs_block_level(L, hvs_storage_to_block(L, idx)) != j) { std::printf("FAIL n=%u: storage %u is level %u\n", n, idx, j); return 1; }
        }
        for (uint32_t nquads : {1u, 20u, 512u, 2052u}) {
            const HvsSegs S = hvs_make_segs(L, nquads, 512u);
            uint32_t expect_first = 0;
            for (uint32_t j = 0; j <= L.K; ++j) {
                const uint32_t T = L.off[j + 1] - L.off[j], seg = S.seg[j];
                if (seg < 8u || seg > HVS_SEG || (seg & (seg - 1u))) { std::printf("FAIL n=%u quads=%u level %u: segment size %u\n", n, nquads, j, seg); return 1; }
                if (S.first[j] != expect_first) { std::printf("FAIL n=%u quads=%u level %u: first segment %u != %u\n", n, nquads, j, S.first[j], expect_first); return 1; }
                const uint32_t nseg = hvs_ceil_div(T, seg);
                if ((uint64_t)nseg * seg < T || (nseg && (uint64_t)(nseg - 1u) * seg >= T)) { std::printf("FAIL n=%u level %u: %u segments of %u for %u blocks\n", n, j, nseg, seg, T); return 1; }
                // a small batch gets enough items to fill the workgroup slots unless the level itself is tiny
                if (seg > 8u && (uint64_t)nseg * nquads < 1024u && T >= 8u * 2u) { std::printf("FAIL n=%u quads=%u level %u: only %u items\n", n, nquads, j, nseg * nquads); return 1; }
                if (nseg >= (1u << 20)) { std::printf("FAIL n=%u level %u: %u segments do not fit the item code\n", n, j, nseg); return 1; }
                expect_first += nseg;
            }
            if (S.first[L.K + 1] != expect_first) { std::printf("FAIL n=%u: total segments\n", n); return 1; }
        }
        std::printf("n=%u nblk=%u K=%u level0=%u blocks radices:", n, L.nblk, L.K, L.off[1] - L.off[0]);
        for (uint32_t j = 1; j <= L.K; ++j) std::printf(" %u", L.radix[j]);
        std::printf("\n");
    }
    std::printf("OK\n");
    return 0;
}
