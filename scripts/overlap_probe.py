"""Does the non-matrix work of one batch (re-scoring, merges, prep) hide under the filter of ANOTHER batch?
Two engines on GPU 0 (each with its own stream, workspace and copy of D) run half-size batches at the same time from two
threads; compared with one engine running the full batch.  Usage: python scripts/overlap_probe.py [n] [batch] [steps]"""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
pkg = importlib.import_module("project---hybrid-vector-search-queries_amd")
import hvs_testlib as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2097152
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ftype = int(sys.argv[4]) if len(sys.argv) > 4 else -1

def make(nq_per_step, first_row):
    e = pkg.Engine(0)
    e.reserve(nq_per_step * (steps + 1))
    e.gen_data(n, T.SEED_DATA, 1, 100)
    e.gen_queries(nq_per_step * (steps + 1), T.SEED_QUERY, 1, 100, ftype, first_row=first_row)
    return e

def run(engs, nq_per_step):
    def worker(e, out):
        for b in range(steps + 1):
            if b == 1:
                bar.wait()
                out[0] = time.perf_counter()
            e.query_resident(b * nq_per_step, nq_per_step, 1.0)
            e.sync()
        out[1] = time.perf_counter()
    bar = threading.Barrier(len(engs))
    outs = [[0, 0] for _ in engs]
    th = [threading.Thread(target=worker, args=(e, o)) for e, o in zip(engs, outs)]
    for t in th: t.start()
    for t in th: t.join()
    el = max(o[1] for o in outs) - min(o[0] for o in outs)
    return len(engs) * nq_per_step * steps / el, el / steps * 1e3

one = make(batch, 0)
v, ms = run([one], batch)
print("one engine, %d queries per step: %.3f M queries/s (%.1f ms per step)" % (batch, v / 1e6, ms), flush=True)
del one
for parts in (2, 3):
    engs = [make(batch // parts, i * (batch // parts) * (steps + 1)) for i in range(parts)]
    v, ms = run(engs, batch // parts)
    print("%d engines side by side, %d queries per step each: %.3f M queries/s (%.1f ms per round)" % (parts, batch // parts, v / 1e6, ms), flush=True)
    del engs
