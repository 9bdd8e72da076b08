"""The reference's entry points under their own names, on top of the C ABI.

    vec_query(nodes, queries, sample_proportion, knn_results)
        reference include/optimized_parallel.hpp:61-62 (same in optimized.hpp:54-55,
        baseline.hpp:68-69): appends one list of 100 ids per query to `knn_results`
        (it push_back's, it does not clear -- optimized_parallel.hpp:159).
    ReadBin / SaveKNN / SaveKNNFull / calc_dist
        reference include/io.h:111-136, :23-36, :50-78, :38-48.
"""
from __future__ import annotations

import numpy as np

from .engine import Engine, HvsError

_engine_cache = {}


def vec_query(nodes, queries, sample_proportion, knn_results, n_gpus=None, k=100):
    """Drop-in for the reference's vec_query.  `nodes`: n x 102, `queries`: nq x 104 (sequences of
    rows or arrays).  Like the reference it reports nothing: preconditions are the caller's
    (n >= 100, full rows); unlike it, a violated precondition raises instead of reading out of
    bounds.  Like the reference, which sizes its own thread pool (optimized_parallel.hpp:73-78), the call
    spreads over the node's GPUs: one per 32768 queries, at most all (`n_gpus` overrides).  `k`: the reference's
    compile-time KNN_LIMIT (optimized_impl.h:26), 8..256."""
    nodes = np.ascontiguousarray(nodes, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    print(f"# data points:  {nodes.shape[0]}")
    print(f"# data point dim:  {nodes.shape[1] if nodes.ndim == 2 else 0}")
    print(f"# queries:      {queries.shape[0]}")
    if n_gpus is None:
        from .engine import library
        have = max(1, int(library().hvs_device_count()))
        n_gpus = max(1, min(have, queries.shape[0] // 32768))
    with Engine(n_gpus=n_gpus) as eng:
        if k != 100:
            eng.set_k(k)
        eng.reserve(queries.shape[0])
        eng.load_data(nodes)
        ids = eng.query(queries, sample_proportion, want_dists=False) if queries.shape[0] else np.empty((0, k), np.uint32)
    for row in ids:
        knn_results.append(row.tolist())


def ReadBin(file_path, num_dimensions):
    """io.h:111-136: uint32 N, then rows of `num_dimensions` f32 (reads whole rows until EOF)."""
    print(f"Reading Data: {file_path}")
    with open(file_path, "rb") as f:
        n = int(np.frombuffer(f.read(4), np.uint32)[0])
        print(f"# of points: {n}")
        raw = np.frombuffer(f.read(), np.float32)
    rows = raw.size // num_dimensions
    data = raw[: rows * num_dimensions].reshape(rows, num_dimensions).copy()
    print("Finish Reading Data")
    return data


def SaveKNN(knns, path="output.bin", k=100):
    """io.h:23-36: nq x k uint32 (k = the reference's KNN_LIMIT, 100 unless the engine ran with another k), no header."""
    a = np.ascontiguousarray(knns, np.uint32)
    if a.ndim != 2 or a.shape[1] != k:
        raise HvsError(-1, f"SaveKNN expects nq x {k} ids")
    a.tofile(path)


def calc_dist(a, b):
    """io.h:38-48: sequential f32 sum over dims 2.. (the .dist file's order, not the hot path's)."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    s = np.float32(0.0)
    for i in range(2, a.shape[0]):
        d = np.float32(a[i] - b[i])
        s = np.float32(s + np.float32(d * d))
    return s


def SaveKNNFull(nodes, queries, knn_ids, path="output.bin.dist"):
    """io.h:50-78 + src/test.cpp:97-110: uint32 nq, then for every (query, neighbour) the
    scalar-order distance between that row and the query."""
    nodes = np.asarray(nodes, np.float32)
    queries = np.asarray(queries, np.float32)
    ids = np.asarray(knn_ids, np.uint32)
    out = np.empty(ids.shape, np.float32)
    for i in range(ids.shape[0]):
        diff = nodes[ids[i], 2:] - queries[i, 4:][None, :]
        sq = (diff * diff).astype(np.float32)
        acc = np.zeros(ids.shape[1], np.float32)
        for k in range(sq.shape[1]):
            acc = (acc + sq[:, k]).astype(np.float32)
        out[i] = acc
    with open(path, "wb") as f:
        f.write(np.uint32(ids.shape[0]).tobytes())
        f.write(out.tobytes())
