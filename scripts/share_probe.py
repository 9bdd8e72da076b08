"""Host -> host time of one hvs_query call of a rank's share of the 4 x 10^6-query set (D = 10^7 resident).
Usage: python scripts/share_probe.py [share ...]"""
import importlib, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import hvs_testlib as T
PKG = importlib.import_module("project---hybrid-vector-search-queries_amd")
shares = [int(x) for x in sys.argv[1:]] or [500000, 1000000]
with PKG.Engine(0) as e:
    e.reserve(max(shares))
    e.gen_data(10_000_000, T.SEED_DATA, T.GEN_V1, 100)
    for share in shares:
        e.gen_queries(share, T.SEED_QUERY, T.GEN_V1, 100, -1, 0)
        q = e.download_queries(0, share)
        ids = np.empty((share, 100), np.uint32)
        e.query(q[:65536], 1.0, want_dists=False)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); e.query(q, 1.0, want_dists=False, out_ids=ids); best = min(best, time.perf_counter() - t0)
        t = e.last_timing()
        e.query_resident(0, share, 1.0); e.sync()
        t0 = time.perf_counter(); e.query_resident(0, share, 1.0); e.sync(); res = time.perf_counter() - t0
        print("share %d: host->host %.1f ms (%.0f q/s), device part %.1f ms, resident %.1f ms (%.0f q/s), launches %d" % (share, best * 1e3, share / best, t.query_ms, res * 1e3, share / res, t.main_kernel_launches))
