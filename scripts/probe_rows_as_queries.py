"""Rows of D used as type-0 queries (what the planner's probe does) through the INT8 filter on a non-uniform law: how many are
retried / sent to the exact engine, and are the answers right?  python scripts/probe_rows_as_queries.py [profile] [n]"""
import importlib, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import hvs_testlib as T
PKG = importlib.import_module("project---hybrid-vector-search-queries_amd")
profile = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
P = 1024
with PKG.Engine(0) as x:
    x.set_engine(PKG.ENGINE_EXACT_SCAN)
    x.gen_data(n, T.SEED_DATA, profile, 100)
    nodes = x.download_data(0, n)
    rows = nodes[(np.arange(P) * (n // P) + (n // P) // 2) % n]
    q = np.full((P, 104), -1.0, np.float32); q[:, 0] = 0; q[:, 4:] = rows[:, 2:]
    qn = q.copy(); qn[:, 4:] += np.float32(1e-3) * np.random.default_rng(1).standard_normal((P, 100)).astype(np.float32)
    x.gen_queries(P, T.SEED_QUERY, profile, 100, 0, 0)
    qr = x.download_queries(0, P)
    want = {k: x.query(v, 1.0) for k, v in (("rows", q), ("rows+noise", qn), ("law", qr))}
for engine in (PKG.ENGINE_MFMA_I8, PKG.ENGINE_MFMA_F16):
    with PKG.Engine(0) as e:
        e.set_engine(engine)
        e.load_data(nodes)
        for k, v in (("rows", q), ("rows+noise", qn), ("law", qr)):
            ids, d = e.query(v, 1.0)
            t = e.last_timing()
            ok = np.array_equal(ids, want[k][0]) and np.array_equal(d.view(np.uint32), want[k][1].view(np.uint32))
            print(f"profile {profile} engine {engine} ran {t.engine} queries={k:10s} identical={ok} retried {t.retry_queries} exact {t.fallback_queries} "
                  f"rescored/query {t.rescored_pairs / P:.0f}")
