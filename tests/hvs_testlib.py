"""Shared helpers for the test-suite, bench.py's CPU leg and the golden tooling.

* ctypes binding of the oracle (oracle/liboracle.so -- TEST INFRASTRUCTURE, never
  used by the product path),
* a numpy twin of the gen-v1 generator (include/hvs_gen.h),
* reader/writer of the reference's binary formats (reference include/io.h:23-36,
  111-136),
* the tie-aware parity checker defined in SURVEY.md section 8c.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
GOLDEN_DIR = os.path.join(REPO, "tests", "golden")

DCOLS, QCOLS, K = 102, 104, 100
GEN_V0, GEN_V1 = 0, 1
GEN_CLUSTER, GEN_PCA, GEN_HEAVY = 2, 3, 4           # non-uniform vector laws (include/hvs_gen.h)
GEN_V1_OUT = 5                                        # gen-v1 with 1 % of the queries outside the data's box
SEED_DATA, SEED_QUERY = 0xD47A5EED, 0x9E3779B9

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


def _fp(a):
    return a.ctypes.data_as(_f32p) if a is not None else None


def _up(a):
    return a.ctypes.data_as(_u32p) if a is not None else None


_oracle = None


def build_oracle():
    """(Re)build oracle/liboracle.so (and oracle/_ref when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def oracle():
    global _oracle
    if _oracle is not None:
        return _oracle
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "hvs_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, os.path.join(ORACLE_DIR, "liboracle.so")], check=True)
    lib = C.CDLL(path)
    lib.hvs_oracle_dist_simd_order.restype = C.c_float
    lib.hvs_oracle_dist_simd_order.argtypes = [_f32p, _f32p]
    lib.hvs_oracle_dist_scalar_order.restype = C.c_float
    lib.hvs_oracle_dist_scalar_order.argtypes = [_f32p, _f32p]
    lib.hvs_oracle_sn.restype = C.c_uint32
    lib.hvs_oracle_sn.argtypes = [C.c_float, C.c_uint32]
    lib.hvs_oracle_predicate.restype = C.c_int
    lib.hvs_oracle_predicate.argtypes = [_f32p, _f32p]
    lib.hvs_oracle_vec_query.restype = C.c_int
    lib.hvs_oracle_vec_query.argtypes = [_f32p, C.c_uint32, _f32p, C.c_uint32, C.c_float, _u32p, _f32p, C.c_int]
    lib.hvs_oracle_vec_query_knn.restype = C.c_int
    lib.hvs_oracle_vec_query_knn.argtypes = [_f32p, C.c_uint32, _f32p, C.c_uint32, C.c_float, _u32p, _f32p,
                                             C.c_int, C.c_int, C.c_int]
    lib.hvs_oracle_pin_threads.restype = None
    lib.hvs_oracle_pin_threads.argtypes = [C.c_int]
    lib.hvs_oracle_place_rows.restype = C.c_int
    lib.hvs_oracle_place_rows.argtypes = [_f32p, _f32p, C.c_uint32, C.c_float, C.c_int, C.c_int]
    lib.hvs_oracle_vec_query_baseline.restype = C.c_int
    lib.hvs_oracle_vec_query_baseline.argtypes = [_f32p, C.c_uint32, _f32p, C.c_uint32, C.c_float, _u32p, _f32p]
    lib.hvs_oracle_dist_file_values.restype = None
    lib.hvs_oracle_dist_file_values.argtypes = [_f32p, _f32p, C.c_uint32, _u32p, _f32p]
    lib.hvs_oracle_dists_for_ids.restype = None
    lib.hvs_oracle_dists_for_ids.argtypes = [_f32p, _f32p, C.c_uint32, _u32p, _f32p]
    lib.hvs_oracle_gen_data.restype = None
    lib.hvs_oracle_gen_data.argtypes = [_f32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32]
    lib.hvs_oracle_gen_queries.restype = None
    lib.hvs_oracle_gen_queries.argtypes = [_f32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, C.c_int]
    lib.hvs_oracle_set_k.restype = C.c_int
    lib.hvs_oracle_set_k.argtypes = [C.c_int]
    lib.hvs_oracle_get_k.restype = C.c_int
    _oracle = lib
    return lib


# --------------------------------------------------------------------------- oracle wrappers

def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def oracle_dist(dvec, qvec, order="simd"):
    d, q = _c(dvec, np.float32), _c(qvec, np.float32)
    assert d.size >= 100 and q.size >= 100
    fn = oracle().hvs_oracle_dist_simd_order if order == "simd" else oracle().hvs_oracle_dist_scalar_order
    return np.float32(fn(_fp(d), _fp(q)))


class oracle_k:
    """`with oracle_k(k):` -- the oracle's k (the reference's compile-time KNN_LIMIT) for the calls inside."""

    def __init__(self, k):
        self.k = int(k)

    def __enter__(self):
        self.old = oracle().hvs_oracle_get_k()
        assert oracle().hvs_oracle_set_k(self.k) == 0, "k outside 8..256"
        return self

    def __exit__(self, *a):
        oracle().hvs_oracle_set_k(self.old)


def oracle_query(nodes, queries, sample_proportion=1.0, engine="canonical", threads=8, part_threads=1,
                 hw_threads=8, run_parallel=False):
    """Returns (ids[nq,k] u32, dists[nq,k] f32), k = 100 unless inside `with oracle_k(k)`.
    engine: canonical | knn | baseline."""
    nodes, queries = _c(nodes, np.float32), _c(queries, np.float32)
    n, nq = nodes.shape[0], queries.shape[0]
    assert nodes.shape[1] == DCOLS and queries.shape[1] == QCOLS
    lib = oracle()
    k = lib.hvs_oracle_get_k()
    ids = np.zeros((nq, k), np.uint32)
    dists = np.zeros((nq, k), np.float32)
    if engine == "canonical":
        rc = lib.hvs_oracle_vec_query(_fp(nodes), n, _fp(queries), nq, sample_proportion, _up(ids), _fp(dists),
                                      threads)
    elif engine == "knn":
        rc = lib.hvs_oracle_vec_query_knn(_fp(nodes), n, _fp(queries), nq, sample_proportion, _up(ids), _fp(dists),
                                          part_threads, hw_threads, int(run_parallel))
    elif engine == "baseline":
        rc = lib.hvs_oracle_vec_query_baseline(_fp(nodes), n, _fp(queries), nq, sample_proportion, _up(ids),
                                               _fp(dists))
    else:
        raise ValueError(engine)
    if rc != 0:
        raise ValueError(f"oracle rejected the input (rc={rc})")
    return ids, dists


def oracle_place_rows(nodes, sample_proportion=1.0, part_threads=0, hw_threads=8):
    """A copy of D whose partitions were first touched by the threads that scan them in the `knn` engine (NUMA placement for
    the timed CPU baseline; hvs_oracle_place_rows).  Returns (copy, threads)."""
    nodes = _c(nodes, np.float32)
    out = np.empty_like(nodes)                                   # untouched pages: the copy below places them
    t = oracle().hvs_oracle_place_rows(_fp(out), _fp(nodes), nodes.shape[0], sample_proportion, part_threads, hw_threads)
    return out, int(t)


def passing_rows_per_query(nodes, queries):
    """m(q) of SURVEY 8d -- rows of D passing each query's predicate (optimized_parallel.hpp:105-138) -- from sorted attribute
    columns instead of nq x n predicate evaluations."""
    n = nodes.shape[0]
    cat, ts = nodes[:, 0], nodes[:, 1]
    order = np.lexsort((ts, cat))
    cat_s, ts_s = cat[order], ts[order]
    t_sorted = np.sort(ts)
    out = np.zeros(len(queries), np.int64)
    for i, q in enumerate(queries):
        typ = np.uint32(q[0]) if q[0] >= 0 else np.uint32(0xFFFFFFFF)
        v = np.float32(np.int32(q[1])) if abs(q[1]) < 2**31 else np.float32(np.nan)
        l, r = q[2], q[3]
        if typ == 0:
            out[i] = n
        elif typ == 1:
            out[i] = np.searchsorted(cat_s, v, "right") - np.searchsorted(cat_s, v, "left")
        elif typ == 2:
            out[i] = max(0, np.searchsorted(t_sorted, r, "right") - np.searchsorted(t_sorted, l, "left"))
        elif typ == 3:
            a, b = np.searchsorted(cat_s, v, "left"), np.searchsorted(cat_s, v, "right")
            out[i] = max(0, np.searchsorted(ts_s[a:b], r, "right") - np.searchsorted(ts_s[a:b], l, "left"))
    return out


def oracle_dists_for_ids(nodes, queries, ids, order="simd"):
    nodes, queries, ids = _c(nodes, np.float32), _c(queries, np.float32), _c(ids, np.uint32)
    assert ids.shape[1] == oracle().hvs_oracle_get_k(), "ids rows do not hold the oracle's k entries (use oracle_k)"
    out = np.zeros(ids.shape, np.float32)
    fn = oracle().hvs_oracle_dists_for_ids if order == "simd" else oracle().hvs_oracle_dist_file_values
    fn(_fp(nodes), _fp(queries), queries.shape[0], _up(ids), _fp(out))
    return out


def gen_data(n, seed=SEED_DATA, profile=GEN_V1, ncat=100, row0=0):
    out = np.empty((n, DCOLS), np.float32)
    oracle().hvs_oracle_gen_data(_fp(out), row0, n, seed, profile, ncat)
    return out


def gen_queries(nq, seed=SEED_QUERY, profile=GEN_V1, ncat=100, force_type=-1, row0=0):
    out = np.empty((nq, QCOLS), np.float32)
    oracle().hvs_oracle_gen_queries(_fp(out), row0, nq, seed, profile, ncat, force_type)
    return out


# --------------------------------------------------------------------------- numpy twin of include/hvs_gen.h

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return z ^ (z >> np.uint64(31))


def _u24(seed, rows, cols):
    with np.errstate(over="ignore"):
        base = _mix64(np.array([seed], np.uint64))[0]
        ctr = base + rows.astype(np.uint64)[:, None] * np.uint64(128) + cols.astype(np.uint64)[None, :]
        return (_mix64(ctr) >> np.uint64(40)).astype(np.uint32)


def _u01(u):
    return u.astype(np.float32) * np.float32(5.9604644775390625e-08)


def _affine(u, scale, lo):
    return (np.float32(scale) * u).astype(np.float32) + np.float32(lo)


def _bell(seed, rows, cols):
    """hvs_gen_bell: Irwin-Hall(4) scaled to unit variance, the same f32 operations in the same order."""
    us = [_u01(_u24((seed + o) & 0xFFFFFFFFFFFFFFFF, rows, cols)) for o in (0, 0x1234567, 0x2468ACE, 0x369D035)]
    a = us[0] + us[1]
    b = us[2] + us[3]
    return ((a + b) - np.float32(2.0)) * np.float32(1.7320508)


def _vec_numpy(seed, profile, rows):
    """hvs_gen_vec_elem for all 100 components of `rows` (non-uniform profiles)."""
    cols = np.arange(2, 102)
    k = np.arange(100, dtype=np.uint32)
    if profile == GEN_CLUSTER:
        with np.errstate(over="ignore"):
            base = _mix64(np.array([seed], np.uint64))[0]
            cl = _mix64(base ^ (rows.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))) % np.uint64(64)
        centre_tab = _affine(_u01(_u24(SEED_DATA ^ 0xC1057E5, np.arange(64, dtype=np.uint64), cols)), 9.0, -4.5)
        return centre_tab[cl.astype(np.int64)] + (np.float32(0.6) * _bell(seed, rows, cols)).astype(np.float32)
    if profile == GEN_PCA:
        w = (32 - (k & 15)).astype(np.float32) * np.float32(0.03125)
        wk = w * np.float32(0.5) ** (k >> 4).astype(np.float32)
        return ((np.float32(6.0) * wk).astype(np.float32)[None, :] * _bell(seed, rows, cols)).astype(np.float32)
    v = _u01(_u24((seed + 0x51ED270) & 0xFFFFFFFFFFFFFFFF, rows, np.array([127])))[:, 0]
    v2 = v * v
    r = np.float32(1.0) + np.float32(15.0) * (v2 * v2)
    return (_affine(_u01(_u24(seed, rows, cols)), 3.0, -1.5) * r[:, None]).astype(np.float32)


def gen_data_numpy(n, seed=SEED_DATA, profile=GEN_V1, ncat=100, row0=0):
    rows = np.arange(row0, row0 + n, dtype=np.uint64)
    u = _u24(seed, rows, np.arange(DCOLS))
    out = _affine(_u01(u), 12.0, -6.0)
    if GEN_CLUSTER <= profile <= GEN_HEAVY:
        out[:, 2:] = _vec_numpy(seed, profile, rows)
    if profile != GEN_V0:
        out[:, 0] = (u[:, 0] % np.uint32(ncat)).astype(np.float32)
        out[:, 1] = _u01(u[:, 1])
    else:
        out[:, 0] = _affine(_u01(u[:, 0]), 2.0, -1.0)
        out[:, 1] = _affine(_u01(u[:, 1]), 6.0, -3.0)
    return out


def gen_queries_numpy(nq, seed=SEED_QUERY, profile=GEN_V1, ncat=100, force_type=-1, row0=0):
    rows = np.arange(row0, row0 + nq, dtype=np.uint64)
    u = _u24(seed, rows, np.arange(QCOLS))
    out = _affine(_u01(u), 12.0, -6.0)
    typ = (u[:, 0] & np.uint32(3)) if force_type < 0 else np.full(nq, force_type, np.uint32)
    out[:, 0] = typ.astype(np.float32)
    has_c = (typ & 1) != 0
    has_t = (typ & 2) != 0
    if profile >= GEN_CLUSTER:
        with np.errstate(over="ignore"):
            base = _mix64(np.array([(seed + 0x0B0F) & 0xFFFFFFFFFFFFFFFF], np.uint64))[0]
            outlier = (_mix64(base + rows) % np.uint64(100)) == 0
        vec = _vec_numpy(seed, profile, rows) if profile <= GEN_HEAVY else out[:, 4:].copy()
        pushed = vec.copy()
        w3 = (32 - np.arange(3)).astype(np.float32) * np.float32(0.03125)
        far = {GEN_CLUSTER: np.full(3, 7.25, np.float32), GEN_PCA: (np.float32(23.0) * w3).astype(np.float32),
               GEN_HEAVY: np.full(3, 26.5, np.float32), GEN_V1_OUT: np.full(3, 6.625, np.float32)}[profile]
        pushed[:, :3] = np.where((u[:, 4:7] & np.uint32(1)) != 0, far[None, :], -far[None, :])
        out[:, 4:] = np.where(outlier[:, None], pushed, vec)
    if profile != GEN_V0:
        v = (u[:, 1] % np.uint32(ncat)).astype(np.float32)
        l = _u01(u[:, 2])
        hi = np.float32(1.0)
    else:
        v = _affine(_u01(u[:, 1]), 2.0, -1.0)
        l = _affine(_u01(u[:, 2]), 6.0, -3.0)
        hi = np.float32(4.0)
    span = (hi - l).astype(np.float32)
    r = l + (_u01(u[:, 3]) * span).astype(np.float32)
    out[:, 1] = np.where(has_c, v, np.float32(-1))
    out[:, 2] = np.where(has_t, l, np.float32(-1))
    out[:, 3] = np.where(has_t, r, np.float32(-1))
    return out.astype(np.float32)


# --------------------------------------------------------------------------- reference binary formats

def write_bin(path, rows):
    """D / Q file: uint32 N then N rows of f32 (reference io.h:111-136, README.md:31-44)."""
    rows = _c(rows, np.float32)
    with open(path, "wb") as f:
        f.write(np.uint32(rows.shape[0]).tobytes())
        f.write(rows.tobytes())


def read_bin(path, cols):
    with open(path, "rb") as f:
        n = int(np.frombuffer(f.read(4), np.uint32)[0])
        return np.frombuffer(f.read(), np.float32, n * cols).reshape(n, cols).copy()


def read_knn(path):
    """output.bin: nq x 100 uint32, no header (reference io.h:23-36)."""
    return np.fromfile(path, np.uint32).reshape(-1, K)


def read_dist_file(path):
    """<output>.dist: uint32 nq then nq x 100 f32 (reference io.h:50-78)."""
    with open(path, "rb") as f:
        nq = int(np.frombuffer(f.read(4), np.uint32)[0])
        return np.frombuffer(f.read(), np.float32, nq * K).reshape(nq, K).copy()


def sha256_of(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


# --------------------------------------------------------------------------- tie-aware parity

def _passes(nodes, q):
    """Predicate of one query over all rows (reference optimized_parallel.hpp:93-96,105-138)."""
    t = q[0]
    typ = int(t) if -1.0 < t < 4.0 else 4                        # uint32(t) truncates toward zero: (-1, 0) -> type 0
    vf = np.float32(int(q[1])) if -2147483648.0 <= float(q[1]) < 2147483648.0 else None
    c, tt = nodes[:, 0], nodes[:, 1]
    if typ == 0:
        return np.ones(nodes.shape[0], bool)
    if typ == 1:
        return (c == vf) if vf is not None else np.zeros(nodes.shape[0], bool)
    if typ == 2:
        return (tt >= q[2]) & (tt <= q[3])
    if typ == 3:
        return ((c == vf) if vf is not None else False) & (tt >= q[2]) & (tt <= q[3])
    return np.zeros(nodes.shape[0], bool)


def check_parity(nodes, queries, got_ids, ref_ids, sample_proportion=1.0, got_dists=None, order="simd",
                 max_report=5):
    """Tie-aware comparison of `got_ids` with `ref_ids` (golden / oracle / reference output).

    (1) exact-order distance sequences are bit-identical;
    (2) below the k-th distance the id multisets per distance value are identical;
    (3) at the k-th distance any choice among valid equal-distance rows is accepted
        (a valid row passes the predicate inside [0,sn) -- or is a pad row when the
        query matched fewer than 100 rows -- and has exactly that distance).
    Returns a dict with counts; raises AssertionError on the first violations.
    """
    nodes, queries = _c(nodes, np.float32), _c(queries, np.float32)
    got_ids, ref_ids = _c(got_ids, np.uint32), _c(ref_ids, np.uint32)
    K = oracle().hvs_oracle_get_k()                      # 100 unless inside `with oracle_k(k)`
    assert got_ids.shape == ref_ids.shape == (queries.shape[0], K), (got_ids.shape, ref_ids.shape)
    n = nodes.shape[0]
    assert got_ids.max(initial=0) < n, "id out of range"
    G = oracle_dists_for_ids(nodes, queries, ref_ids, order)
    B = oracle_dists_for_ids(nodes, queries, got_ids, order)
    if got_dists is not None:
        gd = _c(got_dists, np.float32)
        bad = np.nonzero(gd.view(np.uint32) != B.view(np.uint32))
        assert bad[0].size == 0, f"reported distances differ from exact-order distances at {bad[0][:5]},{bad[1][:5]}"
    sn = int(oracle().hvs_oracle_sn(sample_proportion, n))
    stats = dict(queries=int(queries.shape[0]), identical=0, tie_permuted=0, boundary_choice=0)
    errors = []
    for i in range(queries.shape[0]):
        if np.array_equal(got_ids[i], ref_ids[i]):
            stats["identical"] += 1
            continue
        Bs, Gs = np.sort(B[i]), np.sort(G[i])
        if not np.array_equal(Bs.view(np.uint32), Gs.view(np.uint32)):
            errors.append(f"q{i}: distance multiset differs (first at rank "
                          f"{int(np.nonzero(Bs != Gs)[0][0])}: got {Bs[np.nonzero(Bs != Gs)[0][0]]!r} "
                          f"ref {Gs[np.nonzero(Bs != Gs)[0][0]]!r})")
            continue
        if not np.all(np.diff(B[i]) >= 0):
            errors.append(f"q{i}: output not sorted by distance")
            continue
        dk = Gs[-1]
        ok = True
        for x in np.unique(Gs):
            b = np.sort(got_ids[i][B[i] == x])
            g = np.sort(ref_ids[i][G[i] == x])
            if np.array_equal(b, g):
                continue
            if x != dk:
                errors.append(f"q{i}: ids differ at non-boundary distance {x!r}: got {b[:6]} ref {g[:6]}")
                ok = False
                break
            passing = _passes(nodes[:sn], queries[i])
            m = int(passing.sum())
            if m >= K:
                valid = all(int(r) < sn and passing[int(r)] for r in b) and len(set(b.tolist())) == len(b)
            else:
                valid = all((int(r) < sn and passing[int(r)]) or int(r) >= n - (K - m) for r in b)
            if not valid:
                errors.append(f"q{i}: boundary tie group holds an invalid row: got {b[:6]} ref {g[:6]}")
                ok = False
                break
            stats["boundary_choice"] += 1
        if ok:
            stats["tie_permuted"] += 1
    assert not errors, f"{len(errors)} parity violations, e.g.: " + " | ".join(errors[:max_report])
    return stats


def fp_kat_vectors():
    """The crafted pair of reference src/fp_inaccuracy_test.cpp:79-88 (102 floats each)."""
    vb = [np.float32(0.11232)]
    for i in range(1, 102):
        vb.append(np.float32(float(vb[i - 1]) * (1.321431 if i % 2 == 0 else -0.87382)))
    a = np.array(vb, np.float32)
    return a, a[::-1].copy()
