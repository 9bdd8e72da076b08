"""overlap_probe.py for the "two halves" experiment: env HVS_SMALL_RESCORE / HVS_FILTER_STREAM select the small re-score kernel and the
high-priority filter stream; prints wall time per 2^20 queries for 1 and 2 contexts on GPU 0 and checks the ids of both against each other."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hvs_testlib as T
PKG = importlib.import_module("project---hybrid-vector-search-queries_amd")
n, nq = 10_000_000, 1 << 20
ref = None
for devs in ([0], [0, 0]):
    with PKG.Engine(devices=devs) as e:
        e.reserve(nq)
        e.gen_data(n, T.SEED_DATA, T.GEN_V1, 100)
        e.gen_queries(nq, T.SEED_QUERY, T.GEN_V1, 100, -1, 0)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            e.query_resident(0, nq, 1.0)
            e.sync()
            best = min(best, time.perf_counter() - t0)
        t = e.last_timing()
        ids, _ = e.download_results(0, 4096)
        if ref is None: ref = ids
        print("%d context(s): %.1f ms wall per 2^20 queries (device %.1f ms, filter launches summed %.1f ms over %d) ids equal: %s"
              % (len(devs), best * 1e3, t.query_ms, t.main_kernel_ms, t.main_kernel_launches, np.array_equal(ref, ids)), flush=True)
