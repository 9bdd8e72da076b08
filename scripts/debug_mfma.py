import importlib, os, sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import hvs_testlib as T
PKG = importlib.import_module("project---hybrid-vector-search-queries_amd")

def run(n, nq, ncat=100, seed=1):
    nodes = T.gen_data(n, 1000 + seed, T.GEN_V1, ncat)
    queries = T.gen_queries(nq, 2000 + seed, T.GEN_V1, ncat)
    with PKG.Engine(0) as e:
        e.set_engine(2)
        e.load_data(nodes)
        ids, dists = e.query(queries, 1.0)
        t = e.last_timing()
    ref, refd = T.oracle_query(nodes, queries)
    bad = [i for i in range(nq) if not np.array_equal(ids[i], ref[i])]
    print(f"n={n} nq={nq}: engine={t.engine} fallback={t.fallback_queries} bad={len(bad)}")
    for i in bad[:8]:
        q = queries[i]
        m = int(T._passes(nodes, q).sum())
        inter = len(set(ids[i].tolist()) & set(ref[i].tolist()))
        print(f"  q{i} type={q[0]} v={q[1]} l={q[2]:.4f} r={q[3]:.4f} passing={m} common={inter} got0={dists[i,0]:.3f} ref0={refd[i,0]:.3f} got99={dists[i,99]:.3f} ref99={refd[i,99]:.3f}")
        extra = [int(x) for x in ids[i] if x not in set(ref[i].tolist())][:5]
        print("     extra ids:", extra, "pass?", [bool(T._passes(nodes[x:x+1], q)[0]) for x in extra])
    types = [int(queries[i,0]) for i in bad]
    print("  bad types histogram:", np.bincount(types, minlength=4))

for n, nq in [(1500, 64), (10000, 100), (10000, 400), (70000, 300)]:
    run(n, nq)
