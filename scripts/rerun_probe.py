"""Which queries of a batch are answered a second time (retry with a proven threshold / exact engine), by type and
range size.  Usage: python scripts/rerun_probe.py [n] [nq]"""
import importlib, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import hvs_testlib as T
PKG = importlib.import_module("project---hybrid-vector-search-queries_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
with PKG.Engine(0) as e:
    e.gen_data(n, T.SEED_DATA, T.GEN_V1, 100)
    e.gen_queries(nq, T.SEED_QUERY, T.GEN_V1, 100, -1, 0)
    e.query_resident(0, nq, 1.0); e.sync()
    t = e.last_timing()
    q = e.download_queries(0, nq)
    for which, name in ((1, "retry"), (0, "exact")):
        idx = e.last_reruns(which)
        typ = q[idx, 0].astype(int)
        width = np.where(typ >= 2, q[idx, 3] - q[idx, 2], 1.0)
        rows = width * np.where(typ % 2 == 1, 0.01, 1.0) * n
        print(name, len(idx), "of", nq, "by type", np.bincount(typ, minlength=4).tolist())
        for lo, hi in ((0, 100), (100, 300), (300, 1000), (1000, 3000), (3000, 1e4), (1e4, 1e5), (1e5, 1e6), (1e6, 1e9)):
            sel = (rows >= lo) & (rows < hi)
            allq = ((np.where(q[:, 0] >= 2, q[:, 3] - q[:, 2], 1.0) * np.where(q[:, 0].astype(int) % 2 == 1, 0.01, 1.0) * n >= lo)
                    & (np.where(q[:, 0] >= 2, q[:, 3] - q[:, 2], 1.0) * np.where(q[:, 0].astype(int) % 2 == 1, 0.01, 1.0) * n < hi)).sum()
            print("   expected rows in range [%g, %g): %d reruns of %d such queries" % (lo, hi, sel.sum(), allq))
    print("timing", t.as_dict())
