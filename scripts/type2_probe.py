"""Where does a type-2 (timestamp window) batch lose against type 0?  Same 262144 queries, windows rewritten:
  gen    : gen-v1 windows (l uniform, r = l + u (1 - l))           -- the bench's mix
  same   : every window [0.30, 0.55]                                 -- no heterogeneity at all
  slide  : l uniform in [0, 0.75], r = l + 0.25                      -- equal lengths, different places
  nested : l = 0.375 - w/2, r = 0.625 + ... windows of different lengths around one centre
Prints filter time per step, pairs, evaluated pairs and the pair rate."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hvs_testlib as T
PKG = importlib.import_module("project---hybrid-vector-search-queries_amd")
n, nq = 10_000_000, 262144
rng = np.random.default_rng(1)
with PKG.Engine(0) as e:
    e.reserve(nq)
    e.gen_data(n, T.SEED_DATA, T.GEN_V1, 100)
    base = T.gen_queries(nq, T.SEED_QUERY, T.GEN_V1, 100, 2)
    variants = {}
    variants["gen"] = base
    q = base.copy(); q[:, 2] = 0.30; q[:, 3] = 0.55; variants["same"] = q
    q = base.copy(); l = rng.random(nq, dtype=np.float32) * 0.75; q[:, 2] = l; q[:, 3] = l + 0.25; variants["slide"] = q
    q = base.copy(); w = rng.random(nq, dtype=np.float32) * 0.5; q[:, 2] = 0.5 - w / 2; q[:, 3] = 0.5 + w / 2; variants["nested"] = q
    q = base.copy(); q[:, 0] = 0; q[:, 1:4] = -1; variants["type0"] = q
    for name, qq in variants.items():
        e.upload_queries(qq)
        for rep in range(2):
            e.query_resident(0, nq, 1.0); e.sync()
        t = e.last_timing()
        print("%-7s filter %7.2f ms of %7.2f ms  pairs %.3e evaluated %.3e (x%.3f)  %.2f G pairs/s evaluated, frac of 5 P-op/s %.3f"
              % (name, t.main_kernel_ms, t.query_ms, t.pairs, t.scanned_pairs, t.scanned_pairs / max(t.pairs, 1),
                 t.scanned_pairs / t.main_kernel_ms / 1e6, 200.0 * t.pairs / (t.main_kernel_ms * 1e-3) / 5e15))
