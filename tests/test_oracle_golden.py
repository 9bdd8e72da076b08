"""Pins the oracle (oracle/hvs_oracle.c) against the reference's own known answers and
against output.bin files written by the real reference binaries (tests/golden/*.npz,
made by tests/golden/make_goldens.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

import hvs_testlib as T

ALL = sorted(glob.glob(os.path.join(T.GOLDEN_DIR, "*.npz")))
KGOLDENS = [g for g in ALL if os.path.basename(g).startswith("k")]     # k != 100 (tests/golden/make_goldens_k.py)
GOLDENS = [g for g in ALL if g not in KGOLDENS]
SMALL = [g for g in GOLDENS if "d1m" not in g]


def load(path):
    z = np.load(path)
    nodes = T.gen_data(int(z["n"]), int(z["seed_data"]), int(z["profile"]), int(z["ncat"]))
    queries = T.gen_queries(int(z["nq"]), int(z["seed_query"]), int(z["profile"]), int(z["ncat"]),
                            int(z["force_type"]))
    assert T.sha256_of(nodes) == str(z["sha256_data"]), "generator drifted from the golden's inputs"
    assert T.sha256_of(queries) == str(z["sha256_queries"])
    return z, nodes, queries


def test_fp_known_answer():
    # reference src/fp_inaccuracy_test.cpp:77-97 prints these three values
    a, b = T.fp_kat_vectors()
    assert T.oracle_dist(a[2:], b[2:], "scalar") == np.float32(277762.34375)
    assert T.oracle_dist(a[2:], b[2:], "simd") == np.float32(277762.28125)
    d64 = np.sum((a[2:].astype(np.float64) - b[2:].astype(np.float64)) ** 2)
    assert abs(d64 - 277762.245000211) < 0.05  # the reference's double line uses double inputs


def test_generator_numpy_twin():
    for profile, ncat in ((T.GEN_V1, 100), (T.GEN_V0, 7), (T.GEN_CLUSTER, 10), (T.GEN_PCA, 100), (T.GEN_HEAVY, 3),
                          (T.GEN_V1_OUT, 100)):
        a = T.gen_data(777, 12345, profile, ncat, row0=5)
        b = T.gen_data_numpy(777, 12345, profile, ncat, row0=5)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for ft in (-1, 0, 3):
            qa = T.gen_queries(333, 999, profile, ncat, ft, row0=11)
            qb = T.gen_queries_numpy(333, 999, profile, ncat, ft, row0=11)
            assert np.array_equal(qa.view(np.uint32), qb.view(np.uint32))
    d = T.gen_data(4096)
    assert d[:, 0].min() >= 0 and d[:, 0].max() <= 99 and np.all(d[:, 0] == np.floor(d[:, 0]))
    assert d[:, 1].min() >= 0 and d[:, 1].max() < 1
    assert d[:, 2:].min() >= -6 and d[:, 2:].max() <= 6
    q = T.gen_queries(4096)
    t2 = q[q[:, 0] >= 2]
    assert np.all(t2[:, 3] >= t2[:, 2]) and np.all(t2[:, 3] <= 1.0)
    assert set(np.unique(q[:, 0]).tolist()) == {0.0, 1.0, 2.0, 3.0}


def test_sn_is_a_float_product():
    lib = T.oracle()
    assert lib.hvs_oracle_sn(1.0, 10_000_000) == 10_000_000
    assert lib.hvs_oracle_sn(0.5, 1001) == 500
    assert lib.hvs_oracle_sn(0.001, 10_000) == 10
    assert lib.hvs_oracle_sn(1.0, 16_777_217) == 16_777_216  # n not representable in f32 (optimized_parallel.hpp:67)
    assert lib.hvs_oracle_sn(0.0, 1000) == 0


@pytest.mark.parametrize("path", SMALL, ids=[os.path.basename(p)[:-4] for p in SMALL])
def test_oracle_matches_reference_outputs(path):
    z, nodes, queries = load(path)
    can_ids, can_d = T.oracle_query(nodes, queries, engine="canonical")
    # serial reference engine (optimized.hpp): tie-aware equality with the canonical answer
    st = T.check_parity(nodes, queries, can_ids, z["ids_optimized"], got_dists=can_d)
    assert st["identical"] + st["tie_permuted"] == st["queries"]
    # faithful Knn emulation reproduces the reference's own choice inside tie groups too
    knn_ids, _ = T.oracle_query(nodes, queries, engine="knn", part_threads=1)
    assert np.array_equal(np.sort(knn_ids, axis=1), np.sort(z["ids_optimized"], axis=1))
    # parallel reference engine: T = max(1, min(hw=8, sn/100000)) partitions + serial merge
    par_ids, _ = T.oracle_query(nodes, queries, engine="knn", part_threads=0, hw_threads=8)
    assert np.array_equal(np.sort(par_ids, axis=1), np.sort(z["ids_optimized_parallel"], axis=1))
    T.check_parity(nodes, queries, can_ids, z["ids_optimized_parallel"])
    if "ids_baseline" in z:
        b_ids, b_d = T.oracle_query(nodes, queries, engine="baseline")
        T.check_parity(nodes, queries, b_ids, z["ids_baseline"], got_dists=b_d, order="scalar")
    # the .dist side file (scalar-order distances of the chosen ids; src/test.cpp:97-110)
    for eng in ("optimized", "optimized_parallel", "baseline"):
        if "ids_" + eng in z:
            vals = T.oracle_dists_for_ids(nodes, queries, z["ids_" + eng], order="scalar")
            assert np.array_equal(vals.view(np.uint32), z["distfile_" + eng].view(np.uint32))


@pytest.mark.parametrize("path", KGOLDENS, ids=[os.path.basename(p)[:-4] for p in KGOLDENS])
def test_oracle_matches_reference_built_with_another_k(path):
    """k != 100: the reference compiled with KNN_LIMIT = 8 / 10 / 256 (optimized_impl.h:26) wrote these
    ids; the oracle with the same k must agree -- canonical engine up to ties, the faithful Knn emulation (find_worst
    steps over K - K%8 slots + the overlapping last 8, optimized_impl.h:212-233) as id sets, serial and threaded."""
    z, nodes, queries = load(path)
    k = int(z["k"])
    with T.oracle_k(k):
        can_ids, can_d = T.oracle_query(nodes, queries, engine="canonical")
        assert can_ids.shape == (queries.shape[0], k)
        st = T.check_parity(nodes, queries, can_ids, z["ids_optimized"], got_dists=can_d)
        assert st["identical"] + st["tie_permuted"] == st["queries"]
        T.check_parity(nodes, queries, can_ids, z["ids_optimized_parallel"])
        knn_ids, _ = T.oracle_query(nodes, queries, engine="knn", part_threads=1)
        assert np.array_equal(np.sort(knn_ids, axis=1), np.sort(z["ids_optimized"], axis=1))
        par_ids, _ = T.oracle_query(nodes, queries, engine="knn", part_threads=0, hw_threads=8)
        assert np.array_equal(np.sort(par_ids, axis=1), np.sort(z["ids_optimized_parallel"], axis=1))
        if k > 100:   # types 1/3 match fewer than k rows here: padding with duplicates (optimized_parallel.hpp:149-157)
            assert sum(len(set(r.tolist())) < k for r in z["ids_optimized"]) > 0
    assert T.oracle().hvs_oracle_get_k() == 100


def test_oracle_matches_reference_1m_slice():
    path = os.path.join(T.GOLDEN_DIR, "d1m_x256.npz")
    z, nodes, queries = load(path)
    sel = np.arange(0, 256, 4)  # 64 of the 256 golden queries keeps the CPU suite short
    can_ids, can_d = T.oracle_query(nodes, queries[sel], engine="canonical")
    T.check_parity(nodes, queries[sel], can_ids, z["ids_optimized"][sel], got_dists=can_d)
    T.check_parity(nodes, queries[sel], can_ids, z["ids_optimized_parallel"][sel])
    par_ids, _ = T.oracle_query(nodes, queries[sel], engine="knn", part_threads=0, hw_threads=8, run_parallel=True)
    assert np.array_equal(np.sort(par_ids, axis=1), np.sort(z["ids_optimized_parallel"][sel], axis=1))


def test_padding_and_duplicates_are_exercised():
    z, nodes, queries = load(os.path.join(T.GOLDEN_DIR, "pad_2k_x200.npz"))
    ids = z["ids_optimized"]
    dup = sum(len(set(r.tolist())) < T.K for r in ids)
    assert dup > 0, "the padding golden should contain duplicate ids (optimized_parallel.hpp:149-157)"
    z0, nodes0, queries0 = load(os.path.join(T.GOLDEN_DIR, "v0_5k_x64.npz"))
    t13 = np.isin(queries0[:, 0], (1.0, 3.0))
    tail = np.arange(5000 - 100, 5000)
    for r in z0["ids_optimized"][t13]:
        assert np.array_equal(np.sort(r), tail)  # types 1/3 match nothing with continuous C


def test_rejects_n_below_k():
    with pytest.raises(ValueError):
        T.oracle_query(T.gen_data(99), T.gen_queries(1))
