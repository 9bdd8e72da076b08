"""Parity of the HIP path (through the C ABI) with the oracle and the committed goldens.
Runs on the GPU box: `pytest -m gpu`.  Bit-exact bar: distance sequences identical, ids
identical outside equal-distance tie groups."""
import ctypes as C
import glob
import importlib
import os

import numpy as np
import pytest

import hvs_testlib as T

pytestmark = pytest.mark.gpu
PKG = importlib.import_module("project---hybrid-vector-search-queries_amd")
KGOLDENS = sorted(glob.glob(os.path.join(T.GOLDEN_DIR, "k*.npz")))                  # k != 100 (make_goldens_k.py)
GOLDENS = [g for g in sorted(glob.glob(os.path.join(T.GOLDEN_DIR, "*.npz"))) if g not in KGOLDENS]


FILTER_ENGINES = [PKG.ENGINE_MFMA_FILTER, PKG.ENGINE_MFMA_I8, PKG.ENGINE_MFMA_F16]
FILTER_IDS = ["mfma_bf16", "mfma_i8", "mfma_f16"]


@pytest.fixture(scope="module", params=[PKG.ENGINE_EXACT_SCAN] + FILTER_ENGINES, ids=["exact"] + FILTER_IDS)
def eng(request):
    """All engines must give the same (bit-exact) answers: the FP32 exact-order scan and the BF16 / INT8
    MFMA bound filters + exact re-scoring."""
    e = PKG.Engine(0)
    e.set_engine(request.param)
    e.engine_id = request.param
    yield e
    e.close()


def _inputs(z):
    nodes = T.gen_data(int(z["n"]), int(z["seed_data"]), int(z["profile"]), int(z["ncat"]))
    queries = T.gen_queries(int(z["nq"]), int(z["seed_query"]), int(z["profile"]), int(z["ncat"]),
                            int(z["force_type"]))
    return nodes, queries


@pytest.mark.parametrize("path", GOLDENS, ids=[os.path.basename(p)[:-4] for p in GOLDENS])
def test_matches_reference_goldens(eng, path):
    z = np.load(path)
    nodes, queries = _inputs(z)
    eng.load_data(nodes)
    ids, dists = eng.query(queries, 1.0)
    T.check_parity(nodes, queries, ids, z["ids_optimized"], got_dists=dists)
    T.check_parity(nodes, queries, ids, z["ids_optimized_parallel"])


def test_device_generator_matches_host(eng):
    eng.gen_data(5000, 4242, 1, 37)
    assert np.array_equal(eng.download_data(0, 5000).view(np.uint32), T.gen_data(5000, 4242, 1, 37).view(np.uint32))
    eng.gen_data(3000, 77, 0, 100)
    assert np.array_equal(eng.download_data(100, 2900).view(np.uint32),
                          T.gen_data(2900, 77, 0, 100, row0=100).view(np.uint32))
    for ft in (-1, 2):
        eng.gen_queries(1000, 99, 1, 100, ft, 17)
        assert np.array_equal(eng.download_queries(0, 1000).view(np.uint32),
                              T.gen_queries(1000, 99, 1, 100, ft, row0=17).view(np.uint32))


@pytest.mark.parametrize("n,nq,ncat,sp", [(100, 7, 3, 1.0), (101, 65, 2, 1.0), (4099, 257, 100, 1.0),
                                          (30000, 130, 100, 0.5), (30000, 64, 100, 0.001), (5000, 33, 100, 0.0)])
def test_matches_oracle_ragged_shapes(eng, n, nq, ncat, sp):
    nodes = T.gen_data(n, 1234 + n, T.GEN_V1, ncat)
    queries = T.gen_queries(nq, 99 + nq, T.GEN_V1, ncat)
    eng.load_data(nodes)
    ids, dists = eng.query(queries, sp)
    ref, _ = T.oracle_query(nodes, queries, sp)
    T.check_parity(nodes, queries, ids, ref, sample_proportion=sp, got_dists=dists)


def test_each_query_type_alone_and_invalid_types(eng):
    nodes = T.gen_data(50000, 5)
    eng.load_data(nodes)
    for ft in (0, 1, 2, 3):
        queries = T.gen_queries(200, 1000 + ft, force_type=ft)
        ids, dists = eng.query(queries, 1.0)
        ref, _ = T.oracle_query(nodes, queries)
        T.check_parity(nodes, queries, ids, ref, got_dists=dists)
    q = T.gen_queries(8, 3)
    q[:, 0] = [4.0, 7.5, -1.0, np.nan, 3.999, 0.5, 1e9, 2.0]   # anything outside {0,1,2,3} matches no row
    ids, dists = eng.query(q, 1.0)
    ref, _ = T.oracle_query(nodes, q)
    T.check_parity(nodes, q, ids, ref, got_dists=dists)


def test_exact_distance_ties_and_duplicates(eng):
    # many identical rows: equal distances everywhere, boundary ties at rank 100
    rng = np.random.default_rng(5)
    base = T.gen_data(64, 9)
    nodes = base[rng.integers(0, 64, 6000)].copy()
    nodes[:, 0] = rng.integers(0, 3, 6000)
    nodes[:, 1] = rng.random(6000, dtype=np.float32)
    queries = T.gen_queries(96, 21, ncat=3)
    eng.load_data(nodes)
    ids, dists = eng.query(queries, 1.0)
    ref, _ = T.oracle_query(nodes, queries)
    st = T.check_parity(nodes, queries, ids, ref, got_dists=dists)
    # canonical rule (dist asc, id asc) is deterministic: exact equality with the oracle
    assert st["identical"] == st["queries"]


def test_fp_known_answer_on_device(eng):
    # reference src/fp_inaccuracy_test.cpp: SIMD order gives 277762.28125 (scalar 277762.34375)
    a, b = T.fp_kat_vectors()
    nodes = T.gen_data(100, 1)
    nodes[99] = a
    q = np.zeros((1, 104), np.float32)
    q[0, 0] = 0
    q[0, 1:4] = -1
    q[0, 4:] = b[2:]
    eng.load_data(nodes)
    ids, dists = eng.query(q, 1.0)
    k = int(np.nonzero(ids[0] == 99)[0][0])
    assert dists[0, k] == np.float32(277762.28125)


def test_resident_api_and_argument_errors(eng):
    nodes = T.gen_data(20000, 8)
    queries = T.gen_queries(700, 9)
    eng.load_data(nodes)
    eng.upload_queries(queries)
    eng.query_resident(100, 500, 1.0)
    eng.sync()
    ids, dists = eng.download_results(100, 500)
    ref, _ = T.oracle_query(nodes, queries[100:600])
    T.check_parity(nodes, queries[100:600], ids, ref, got_dists=dists)
    t = eng.last_timing()
    assert t.nq == 500 and t.query_ms > 0 and t.engine == eng.engine_id and t.fallback_queries == 0
    assert t.main_kernel_ms > 0 or eng.engine_id in FILTER_ENGINES
    passing = sum(int(T._passes(nodes, q).sum()) for q in queries[100:600])
    assert t.pairs == passing
    # stream-ordered hand-off (hvs_stream_wait): work on another stream starts after the context's work; the NULL stream here
    eng.query_resident(0, 700, 1.0)
    eng.stream_wait(0)
    eng.sync()
    ids_all, _ = eng.download_results(0, 700)
    assert np.array_equal(ids_all[100:600], ids)
    with pytest.raises(PKG.HvsError):
        eng.query_resident(600, 200, 1.0)      # range outside the resident set
    with pytest.raises(PKG.HvsError):
        eng.load_data(T.gen_data(99))          # n < 100: the reference's padding would underflow
    with pytest.raises(PKG.HvsError):
        eng.load_data(np.zeros((200, 100), np.float32))


def test_vec_query_mirror(eng):
    nodes = T.gen_data(3000, 2)
    queries = T.gen_queries(40, 3)
    res = [[1, 2, 3]]
    PKG.vec_query(nodes, queries, 1.0, res)
    assert res[0] == [1, 2, 3] and len(res) == 41     # appends, does not clear (optimized_parallel.hpp:159)
    ref, _ = T.oracle_query(nodes, queries)
    T.check_parity(nodes, queries, np.array(res[1:], np.uint32), ref)


def test_cli_driver_matches_reference_files(tmp_path):
    """csrc/hvs_search.out keeps the reference's argv contract and file formats (src/test.cpp:51-110)."""
    import subprocess
    z = np.load(os.path.join(T.GOLDEN_DIR, "config1_10k_x100.npz"))
    nodes, queries = _inputs(z)
    d, q, o = str(tmp_path / "d.bin"), str(tmp_path / "q.bin"), str(tmp_path / "out.bin")
    T.write_bin(d, nodes)
    T.write_bin(q, queries)
    r = subprocess.run([PKG.cli_path(), d, q, o], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Vector Search took" in r.stderr
    ids = T.read_knn(o)
    T.check_parity(nodes, queries, ids, z["ids_optimized"])
    want = T.oracle_dists_for_ids(nodes, queries, ids, order="scalar")
    assert np.array_equal(T.read_dist_file(o + ".dist").view(np.uint32), want.view(np.uint32))
    bad = subprocess.run([PKG.cli_path(), "a", "b", "c", "d"], capture_output=True, text=True)
    assert bad.returncode == 1 and "[source_path] [query_path] [output_path]" in bad.stdout
    # another k through the driver (the reference's KNN_LIMIT is a compile-time constant, optimized_impl.h:26)
    r = subprocess.run([PKG.cli_path(), d, q, o + "10"], capture_output=True, text=True, env=dict(os.environ, HVS_K="10"))
    assert r.returncode == 0, r.stderr
    ids10 = np.fromfile(o + "10", np.uint32).reshape(-1, 10)
    with T.oracle_k(10):
        ref10, _ = T.oracle_query(nodes, queries)
        T.check_parity(nodes, queries, ids10, ref10)
    d10 = np.fromfile(o + "10.dist", np.uint8)
    assert d10.size == 4 + 4 * ids10.size and int(d10[:4].view(np.uint32)[0]) == queries.shape[0]


@pytest.mark.parametrize("fengine", FILTER_ENGINES, ids=FILTER_IDS)
@pytest.mark.parametrize("n", [2048 * 4 + 17, 70_000, 300_001])
def test_mfma_engine_levels_and_ranges(n, fengine):
    """Sizes that give 1, 2 and 3 index levels; queries of every type incl. empty and tiny ranges."""
    nodes = T.gen_data(n, 4000 + n, T.GEN_V1, 20)
    nodes[::997, 1] = np.nan                 # NaN timestamps never satisfy l <= T <= r but do satisfy C == v
    nodes[5::1013, 0] = np.nan
    nodes[7::500, 1] = -0.0
    queries = T.gen_queries(300, 77 + n, T.GEN_V1, 20)
    queries[0, :4] = [2, -1, 0.5, 0.4]       # l > r: empty range -> pure padding
    queries[1, :4] = [3, 5, 0.25, 0.2501]    # a handful of rows
    queries[2, :4] = [1, 1000, -1, -1]       # category that does not exist
    queries[3, :4] = [2, -1, -0.0, 0.0]      # only T == +-0
    queries[4, :4] = [2, -1, np.nan, 1.0]    # NaN bound matches nothing
    queries[5, :4] = [2, -1, -np.inf, np.inf]
    with PKG.Engine(0) as e:
        e.set_engine(fengine)
        e.load_data(nodes)
        ids, dists = e.query(queries, 1.0)
        t = e.last_timing()
        assert t.engine == fengine and t.fallback_queries == 0
        ref, _ = T.oracle_query(nodes, queries)
        T.check_parity(nodes, queries, ids, ref, got_dists=dists)
        passing = sum(int(T._passes(nodes, q).sum()) for q in queries)
        assert t.pairs == passing
        # sampled prefixes: the filter engine down to sn = n/4 (its exact stages drop rows >= sn), the exact
        # engine below that
        for sp, want_engine in ((0.6, fengine), (0.3, fengine), (0.1, PKG.ENGINE_EXACT_SCAN)):
            ids2, d2 = e.query(queries[:80], sp)
            t2 = e.last_timing()
            assert t2.engine == want_engine, (sp, t2.engine)
            ref2, _ = T.oracle_query(nodes, queries[:80], sp)
            T.check_parity(nodes, queries[:80], ids2, ref2, sample_proportion=sp, got_dists=d2)
            sn = int(T.oracle().hvs_oracle_sn(sp, n))
            assert t2.pairs == sum(int(T._passes(nodes[:sn], q).sum()) for q in queries[:80])


@pytest.mark.parametrize("fengine", FILTER_ENGINES, ids=FILTER_IDS)
def test_mfma_engine_overflow_falls_back_to_exact(fengine):
    """Thousands of rows at exactly the same distance overflow a candidate list; those queries are
    re-run by the exact engine and still match the canonical answer."""
    base = T.gen_data(1, 3)[0]
    nodes = np.tile(base, (40000, 1))
    nodes[:, 0] = np.arange(40000) % 4
    nodes[:, 1] = (np.arange(40000) % 1000) / 1000.0
    nodes[::7, 2:] += 0.5
    queries = T.gen_queries(64, 5, ncat=4)
    with PKG.Engine(0) as e:
        e.set_engine(fengine)
        e.load_data(nodes)
        ids, dists = e.query(queries, 1.0)
        assert e.last_timing().fallback_queries > 0
    ref, _ = T.oracle_query(nodes, queries)
    st = T.check_parity(nodes, queries, ids, ref, got_dists=dists)
    assert st["identical"] == st["queries"]


@pytest.mark.parametrize("flag", ["-DCHECK_BF16", "-DCHECK_F16"], ids=["bf16", "f16"])
def test_mfma_accumulation_error_is_inside_the_budget(tmp_path, flag):
    """the filter's bound assumes |mfma chain - exact| <= 256 u sum|terms|; measure it for both 16-bit float
    tile formats (FP16: half denormals may be flushed by the matrix pipe, which the bound allows for)."""
    import subprocess
    exe = str(tmp_path / "mfma_bound_check.out")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-w", flag,
                    os.path.join(T.REPO, "tests", "mfma_bound_check.hip"), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


@pytest.mark.parametrize("src", ["mfma_i8_layout_check.hip", "mfma_i8x16_layout_check.hip"], ids=["32x32x32", "16x16x64"])
def test_int8_mfma_operand_layout(tmp_path, src):
    """The INT8 filters' fragment layouts and accumulator-init semantics of v_mfma_i32_32x32x32_i8 (HVS_FMT_I8) and
    v_mfma_i32_16x16x64_i8 (HVS_FMT_I8X16, the default): exact integers."""
    import subprocess
    exe = str(tmp_path / (src[:-4] + ".out"))
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-w",
                    os.path.join(T.REPO, "tests", src), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_many_small_batches_and_nonfinite_inputs():
    """HVS_MFMA_BATCH / HVS_EXACT_BATCH split one call into many batches; queries or data with inf/NaN
    components are answered by the exact engine."""
    import subprocess
    import sys
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import hvs_testlib as T
PKG = importlib.import_module('project---hybrid-vector-search-queries_amd')
nodes = T.gen_data(40000, 31, T.GEN_V1, 10); queries = T.gen_queries(700, 32, T.GEN_V1, 10)
queries[5, 10] = np.inf; queries[6, 50] = 1e30; queries[7, 20] = np.nan
queries[8, 0] = -0.5          # uint32(-0.5f) == 0: a type-0 query (reference optimized_parallel.hpp:93)
queries[9, :2] = [1.0, -2147483648.0]   # int32(-2^31) is INT_MIN: matches no category, pure padding
ref, refd = T.oracle_query(nodes, queries)
for engine in (1, 2, 3):
    with PKG.Engine(0) as e:
        e.set_engine(engine); e.load_data(nodes)
        ids, d = e.query(queries, 1.0)
        t = e.last_timing()
    # non-finite distances have a defined place in the canonical order (finite < +inf < NaN, ties by id): the
    # answers of queries 5..7 are compared like all others, bit for bit
    assert np.array_equal(ids, ref), np.nonzero((ids != ref).any(axis=1))[0][:10]
    assert np.array_equal(d.view(np.uint32), refd.view(np.uint32))
    ok = [i for i in range(700) if i not in (5, 6, 7)]
    T.check_parity(nodes, queries[ok], ids[ok], ref[ok], got_dists=d[ok])
    assert t.engine == engine, (t.engine, engine)
    if engine >= 2: assert t.fallback_queries >= 3
# data with inf / NaN / overflowing components: every engine request ends in the exact engine, which admits a row
# with a non-finite distance while fewer than 100 rows are held (reference optimized_impl.h:301-304)
bad = nodes.copy(); bad[:, 0] = (np.arange(40000) % 400)        # 100 rows per category
r5, r6, r7 = (np.nonzero(bad[:, 0] == k)[0] for k in (5, 6, 7))
bad[r5[:60], 40] = np.inf; bad[r6[::2], 41] = np.nan; bad[r7[:30], 42] = 3e38   # inf / NaN / overflowing distances
qs = queries[:64].copy(); qs[:, 1] = np.arange(64) % 16; qs[:16, 0] = 1; qs[:16, 2:4] = -1   # type 1 on categories 0..15
ref2, ref2d = T.oracle_query(bad, qs)
with PKG.Engine(0) as e:
    e.set_engine(2); e.load_data(bad)
    ids, d = e.query(qs, 1.0)
    assert e.last_timing().engine == 1      # non-finite data: no index, exact engine
    assert np.array_equal(ids, ref2) and np.array_equal(d.view(np.uint32), ref2d.view(np.uint32))
print('SUBPROCESS-OK')
"""
    env = dict(os.environ, HVS_MFMA_BATCH="256", HVS_EXACT_BATCH="128")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=T.REPO)
    assert "SUBPROCESS-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_full_size_d1e7_engines_agree_and_properties_hold():
    """BASELINE.json headline data size (D = 10^7 rows, generated in HBM).  The MFMA engine must give
    the exact engine's bits on a few thousand mixed queries; size-independent properties are
    checked on all of them and the oracle confirms 256 (64 of every type)."""
    n, nq = 10_000_000, 4096
    with PKG.Engine(0) as e:
        e.set_engine(PKG.ENGINE_MFMA_FILTER)
        e.gen_data(n, T.SEED_DATA, T.GEN_V1, 100)
        e.gen_queries(nq, T.SEED_QUERY, T.GEN_V1, 100, -1, 0)
        queries = e.download_queries(0, nq)
        e.query_resident(0, nq, 1.0)
        e.sync()
        t = e.last_timing()
        ids, dists = e.download_results(0, nq)
        assert t.engine == PKG.ENGINE_MFMA_FILTER and t.fallback_queries == 0
        # the INT8 filter (tiles rebuilt in that format) and the planner's own choice: same bits
        for other in (PKG.ENGINE_MFMA_I8, PKG.ENGINE_AUTO):
            e.set_engine(other)
            e.query_resident(0, nq, 1.0)
            e.sync()
            t8 = e.last_timing()
            ids8, dists8 = e.download_results(0, nq)
            assert t8.engine in FILTER_ENGINES and t8.fallback_queries == 0
            assert other != PKG.ENGINE_MFMA_I8 or t8.engine == PKG.ENGINE_MFMA_I8
            assert np.array_equal(ids, ids8) and np.array_equal(dists.view(np.uint32), dists8.view(np.uint32))
            print("engine", t8.engine, "rescored pairs/query", t8.rescored_pairs / nq, "(bf16:", t.rescored_pairs / nq, ")")
        # the same queries through the exact engine (first 1024: ~10^10 exact pairs)
        e.set_engine(PKG.ENGINE_EXACT_SCAN)
        e.query_resident(0, 1024, 1.0)
        e.sync()
        assert e.last_timing().engine == PKG.ENGINE_EXACT_SCAN
        ids_x, dists_x = e.download_results(0, 1024)
        nodes = e.download_data(0, n)
    assert np.array_equal(ids[:1024], ids_x) and np.array_equal(dists[:1024].view(np.uint32), dists_x.view(np.uint32))
    # properties on all 4096 answers
    assert ids.max() < n and np.all(np.diff(dists, axis=1) >= 0)
    assert np.array_equal(T.oracle_dists_for_ids(nodes, queries, ids).view(np.uint32), dists.view(np.uint32))
    typ = queries[:, 0].astype(int)
    c = nodes[:, 0][ids]
    tt = nodes[:, 1][ids]
    has_c, has_t = (typ & 1) == 1, (typ & 2) == 2
    notpad = ids < n - 100                                    # padding ids come from the last 100 rows only
    assert np.all((c == queries[:, 1:2])[has_c][notpad[has_c]])
    assert np.all(((tt >= queries[:, 2:3]) & (tt <= queries[:, 3:4]))[has_t][notpad[has_t]])
    for row in ids[typ == 0][::29]:
        assert len(set(row.tolist())) == 100                  # no duplicates when >= 100 rows match
    # the oracle on 64 queries of every type (256 in all)
    pick = np.concatenate([np.nonzero(typ == k)[0][:64] for k in range(4)])
    assert pick.size == 256
    ref, _ = T.oracle_query(nodes, queries[pick], threads=16)
    st = T.check_parity(nodes, queries[pick], ids[pick], ref, got_dists=dists[pick])
    print("oracle-checked queries at D=1e7:", st)


@pytest.mark.parametrize("name", ["config1_10k_x100", "pad_2k_x200", "v0_5k_x64"])
def test_baseline_engine_order_matches_baseline_out(name):
    """BASELINE.json configs[0]: the reference's baseline engine sums sequentially (baseline.hpp:53-64);
    HVS_ORDER_SCALAR reproduces baseline.out (ids tie-aware, scalar-order distances bit for bit)."""
    z = np.load(os.path.join(T.GOLDEN_DIR, name + ".npz"))
    nodes, queries = _inputs(z)
    with PKG.Engine(0) as e:
        e.set_distance_order(1)
        e.load_data(nodes)
        ids, dists = e.query(queries, 1.0)
        assert e.last_timing().engine == PKG.ENGINE_EXACT_SCAN
    T.check_parity(nodes, queries, ids, z["ids_baseline"], got_dists=dists, order="scalar")
    ref, _ = T.oracle_query(nodes, queries, engine="baseline")
    T.check_parity(nodes, queries, ids, ref, got_dists=dists, order="scalar")


def test_clustered_data_parity_both_engines():
    """Tight Gaussian clusters (distances crowd together, candidate lists fill up): both engines must
    still agree with the oracle; the MFMA engine may hand some queries to the exact engine."""
    rng = np.random.default_rng(17)
    n, nq, ncl = 120_000, 400, 300
    centers = rng.uniform(-6, 6, (ncl, 100)).astype(np.float32)
    lab = rng.integers(0, ncl, n)
    nodes = np.empty((n, 102), np.float32)
    nodes[:, 2:] = centers[lab] + rng.normal(0, 0.05, (n, 100)).astype(np.float32)
    nodes[:, 0] = rng.integers(0, 8, n)
    nodes[:, 1] = rng.random(n, dtype=np.float32)
    queries = T.gen_queries(nq, 5, ncat=8)
    queries[:, 4:] = centers[rng.integers(0, ncl, nq)] + rng.normal(0, 0.05, (nq, 100)).astype(np.float32)
    ref, _ = T.oracle_query(nodes, queries)
    for engine in [PKG.ENGINE_EXACT_SCAN] + FILTER_ENGINES:
        with PKG.Engine(0) as e:
            e.set_engine(engine)
            e.load_data(nodes)
            ids, dists = e.query(queries, 1.0)
            t = e.last_timing()
        T.check_parity(nodes, queries, ids, ref, got_dists=dists)
        print("engine", engine, "fallback queries", t.fallback_queries, "rescored pairs/query", t.rescored_pairs / nq)


@pytest.mark.parametrize("engine", [1, 2, 3, 4], ids=["exact"] + FILTER_IDS)
def test_data_sharded_mode_virtual_ranks(engine):
    """D-sharded mode on one GPU: 3 contexts hold disjoint row ranges (padding off),
    a 4th holds the last 100 rows; sharding.merge_data_shards gives the whole-set answer."""
    sharding = importlib.import_module("project---hybrid-vector-search-queries_amd.sharding")
    n, nq, world = 150_000, 300, 3
    nodes = T.gen_data(n, 41, T.GEN_V1, 40)
    queries = T.gen_queries(nq, 42, T.GEN_V1, 40)
    queries[0, :4] = [3, 7, 0.5, 0.5001]       # a few rows only -> padding from the global tail
    queries[1, :4] = [1, 999, -1, -1]          # no row at all
    parts = []
    for r in range(world):
        r0, r1 = sharding.row_shard_range(n, r, world)
        with PKG.Engine(0) as e:
            e.set_engine(engine)
            e.set_padding(False)
            e.load_data(nodes[r0:r1])
            ids, dists = e.query(queries, 1.0)
            assert e.last_timing().engine == engine
        parts.append((ids, dists, r0))
    with PKG.Engine(0) as e:
        e.load_data(nodes[n - 100:])
        pad = sharding.tail_pad_dists(lambda q: e.query(q, 1.0), queries)
    ids, dists = sharding.merge_data_shards(parts, n, pad)
    ref, _ = T.oracle_query(nodes, queries)
    T.check_parity(nodes, queries, ids, ref, got_dists=dists)
    assert (parts[0][0] == 0xFFFFFFFF).any(), "some shard answers must be partial"


def test_data_sharded_device_merge_matches_host_merge():
    """hvs_merge_shards_device (the GPU analogue of Knn::merge for row shards) on the layout an all_gather
    produces, [shard][query][100], against sharding.merge_data_shards.  Runs in a subprocess that imports torch
    BEFORE the library (one HIP runtime per process, as bench.py does)."""
    import subprocess
    import sys
    code = r"""
import importlib, sys, numpy as np, torch
torch.cuda.init()
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import hvs_testlib as T
PKG = importlib.import_module('project---hybrid-vector-search-queries_amd')
sharding = importlib.import_module('project---hybrid-vector-search-queries_amd.sharding')
n, nq, world = 150_000, 300, 3
nodes = T.gen_data(n, 41, T.GEN_V1, 40); queries = T.gen_queries(nq, 42, T.GEN_V1, 40)
queries[0, :4] = [3, 7, 0.5, 0.5001]; queries[1, :4] = [1, 999, -1, -1]
parts = []
for r in range(world):
    r0, r1 = sharding.row_shard_range(n, r, world)
    with PKG.Engine(0) as e:
        e.set_padding(False); e.load_data(nodes[r0:r1])
        ids, dists = e.query(queries, 1.0)
    parts.append((ids, dists, r0))
with PKG.Engine(0) as e:
    e.load_data(nodes[n - 100:])
    pad = sharding.tail_pad_dists(lambda q: e.query(q, 1.0), queries)
ids, dists = sharding.merge_data_shards(parts, n, pad)
ids_all = torch.from_numpy(np.stack([p[0] for p in parts]).view(np.int32)).cuda()
d_all = torch.from_numpy(np.stack([p[1] for p in parts])).cuda()
pad_t = torch.from_numpy(np.ascontiguousarray(pad, np.float32)).cuda()
out_i = torch.empty((nq, 100), dtype=torch.int32, device='cuda'); out_d = torch.empty((nq, 100), dtype=torch.float32, device='cuda')
torch.cuda.synchronize()
with PKG.Engine(0) as e:
    e.merge_shards_device(ids_all.data_ptr(), d_all.data_ptr(), [p[2] for p in parts], nq, n, pad_t.data_ptr(), out_i.data_ptr(), out_d.data_ptr())
    e.sync()
    assert np.array_equal(out_i.cpu().numpy().view(np.uint32), ids)
    assert np.array_equal(out_d.cpu().numpy().view(np.uint32), dists.view(np.uint32))
    try:
        e.merge_shards_device(ids_all.data_ptr(), d_all.data_ptr(), [0] * 17, nq, n, pad_t.data_ptr(), out_i.data_ptr())
        raise SystemExit('17 shards must be rejected')
    except PKG.HvsError:
        pass
ref, _ = T.oracle_query(nodes, queries)
T.check_parity(nodes, queries, out_i.cpu().numpy().view(np.uint32), ref, got_dists=out_d.cpu().numpy())
print('SUBPROCESS-OK')
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=T.REPO)
    assert "SUBPROCESS-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_planner_picks_a_format_and_answers_stay_identical():
    """HVS_ENGINE_AUTO: evenly filled data gets INT8 tiles, data whose quantisation band is hopeless
    (one huge coordinate per row scale) does not; the answers are the oracle's either way."""
    n, nq = 60_000, 200
    nodes = T.gen_data(n, 71, T.GEN_V1, 10)
    queries = T.gen_queries(nq, 72, T.GEN_V1, 10)
    heavy = nodes.copy()
    rng = np.random.default_rng(5)
    heavy[:, 2:] = (rng.standard_normal((n, 100)) * np.exp(rng.standard_normal((n, 1)) * 2.5)).astype(np.float32)
    hq = queries.copy()
    hq[:, 4:] = (rng.standard_normal((nq, 100)) * np.exp(rng.standard_normal((nq, 1)) * 2.5)).astype(np.float32)
    chosen = []
    for d, q in ((nodes, queries), (heavy, hq)):
        ref, _ = T.oracle_query(d, q)
        with PKG.Engine(0) as e:
            e.load_data(d)                       # engine AUTO
            ids, dists = e.query(q, 1.0)
            t = e.last_timing()
            chosen.append(t.engine)
            T.check_parity(d, q, ids, ref, got_dists=dists)
            e.set_engine(PKG.ENGINE_MFMA_I8)     # forced INT8 on the same data: same bits, maybe via fallbacks
            ids8, dists8 = e.query(q, 1.0)
            assert np.array_equal(ids, ids8) and np.array_equal(dists.view(np.uint32), dists8.view(np.uint32))
            print("auto engine", t.engine, "forced int8 fallback queries", e.last_timing().fallback_queries)
    assert chosen[0] == PKG.ENGINE_MFMA_I8 and chosen[1] in FILTER_ENGINES + [PKG.ENGINE_EXACT_SCAN]


@pytest.mark.parametrize("fengine", FILTER_ENGINES, ids=FILTER_IDS)
def test_filter_engines_on_degenerate_value_ranges(fengine):
    """All rows equal, vectors scaled to 1e-20 / 1e+15, queries far outside the data's bounding box: the bound
    either holds or the engine steps aside (other format / exact engine); answers never change."""
    base = T.gen_data(40_000, 81, T.GEN_V1, 5)
    queries = T.gen_queries(96, 82, T.GEN_V1, 5)
    cases = []
    same = base.copy(); same[:, 2:] = same[0, 2:]
    cases.append((same, queries))
    for sc in (1e-20, 1e15):   # (squared distances still finite in f32)
        d = base.copy(); d[:, 2:] *= np.float32(sc)
        q = queries.copy(); q[:, 4:] *= np.float32(sc)
        cases.append((d, q))
    far = queries.copy(); far[::3, 4:] += 50.0; far[1::3, 4:] *= -40.0
    cases.append((base, far))
    for d, q in cases:
        ref, _ = T.oracle_query(d, q)
        with PKG.Engine(0) as e:
            e.set_engine(fengine)
            e.load_data(d)
            ids, dists = e.query(q, 1.0)
            print("engine ran:", e.last_timing().engine, "fallback", e.last_timing().fallback_queries)
        T.check_parity(d, q, ids, ref, got_dists=dists)


@pytest.mark.parametrize("force_type", [0, -1], ids=["config1_type0", "config2_mixed"])
def test_baseline_configs_1_and_2_d1e6_q1e4(force_type):
    """BASELINE configs[1] / [2] at their exact size: D = 10^6, Q = 10^4 in one call, type-0 only and mixed types.  Every
    engine returns the exact engine's bits for all 10^4 queries; the oracle confirms a sample."""
    n, nq = 1_000_000, 10_000
    with PKG.Engine(0) as e:
        e.gen_data(n, T.SEED_DATA, T.GEN_V1, 100)
        e.gen_queries(nq, T.SEED_QUERY, T.GEN_V1, 100, force_type, 0)
        queries = e.download_queries(0, nq)
        res = {}
        for engine in [PKG.ENGINE_AUTO, PKG.ENGINE_EXACT_SCAN] + FILTER_ENGINES:
            e.set_engine(engine)
            e.query_resident(0, nq, 1.0)
            e.sync()
            t = e.last_timing()
            assert t.nq == nq and (engine == PKG.ENGINE_AUTO or t.engine == engine)
            res[engine] = e.download_results(0, nq)
            print("engine", engine, "ran", t.engine, "device ms %.2f" % t.query_ms, "retried", t.retry_queries, "fallback", t.fallback_queries)
        nodes = e.download_data(0, n)
    want_i, want_d = res[PKG.ENGINE_EXACT_SCAN]
    for engine, (ids, d) in res.items():
        assert np.array_equal(ids, want_i) and np.array_equal(d.view(np.uint32), want_d.view(np.uint32)), engine
    sel = np.arange(0, nq, nq // 64)[:64]
    ref, _ = T.oracle_query(nodes, queries[sel], threads=8)
    T.check_parity(nodes, queries[sel], want_i[sel], ref, got_dists=want_d[sel])


def test_largest_batch_2pow21_queries_filters_agree():
    """The largest batch the library forms (2^21 queries, the bench default) at D = 10^7: the INT8 and BF16
    filters must return identical bits for every query, and the exact scan must confirm a 2048-query sample."""
    n, nq = 10_000_000, 1 << 21
    with PKG.Engine(0) as e:
        e.gen_data(n, T.SEED_DATA, T.GEN_V1, 100)
        e.gen_queries(nq, T.SEED_QUERY + 5, T.GEN_V1, 100, -1, 0)
        res = {}
        for engine in FILTER_ENGINES:
            e.set_engine(engine)
            e.query_resident(0, nq, 1.0)
            e.sync()
            t = e.last_timing()
            assert t.engine == engine and t.fallback_queries == 0 and t.nq == nq
            res[engine] = e.download_results(0, nq)
        a = res[PKG.ENGINE_MFMA_I8]
        for other in (PKG.ENGINE_MFMA_FILTER, PKG.ENGINE_MFMA_F16):
            b = res[other]
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), other
        sel = np.arange(0, nq, nq // 2048)[:2048]
        q = e.download_queries(0, nq)[sel]
        e.set_engine(PKG.ENGINE_EXACT_SCAN)
        ids, d = e.query(q, 1.0)
        assert np.array_equal(ids, a[0][sel]) and np.array_equal(d.view(np.uint32), a[1][sel].view(np.uint32))
        assert np.all(np.diff(a[1], axis=1) >= 0) and a[0].max() < n


@pytest.mark.parametrize("name,preseed", [("config1_10k_x100", 0), ("pad_2k_x200", 3)])
def test_seam_translation_unit_with_reference_signature(tmp_path, name, preseed):
    """tests/seam_main.cpp is src/test.cpp's body with include/hvs_vec_query.hpp as the engine header (the `IMPL == 4`
    branch of INTEGRATION.md): the reference's exact vec_query signature, append-not-clear semantics
    (optimized_parallel.hpp:159), output.bin parity with the reference's own optimized.out."""
    import subprocess
    z = np.load(os.path.join(T.GOLDEN_DIR, name + ".npz"))
    nodes, queries = _inputs(z)
    d, q, o = str(tmp_path / "d.bin"), str(tmp_path / "q.bin"), str(tmp_path / "out.bin")
    T.write_bin(d, nodes)
    T.write_bin(q, queries)
    r = subprocess.run([PKG.seam_path(), d, q, o] + ([str(preseed)] if preseed else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
    assert "# data points:  %d" % nodes.shape[0] in r.stdout and "seam ok" in r.stderr
    ids = T.read_knn(o)
    T.check_parity(nodes, queries, ids, z["ids_optimized"])
    T.check_parity(nodes, queries, ids, z["ids_optimized_parallel"])


@pytest.mark.parametrize("engine", [PKG.ENGINE_AUTO, PKG.ENGINE_EXACT_SCAN], ids=["auto", "exact"])
def test_multi_gpu_context_virtual_ranks(engine):
    """hvs_create_on_devices with three parts on GPU 0 ("virtual ranks"): D replicated by device-to-device copies, the
    queries of a call cut into contiguous ranges, every part writing its slice of the caller's arrays.  Bit-equal to
    the one-GPU context through hvs_query (both gather modes) and through the resident API."""
    n, nq = 120_000, 10_001                       # 10001 = 3334 + 3334 + 3333: uneven parts
    nodes = T.gen_data(n, 91, T.GEN_V1, 30)
    queries = T.gen_queries(nq, 92, T.GEN_V1, 30)
    queries[17, 10] = np.inf                      # overflow/fallback queries in every part (the peer gather re-sends their rows)
    queries[5000, 11] = np.inf
    queries[9000, 12] = np.nan
    with PKG.Engine(0) as one:
        one.set_engine(engine)
        one.load_data(nodes)
        ids1, d1 = one.query(queries, 1.0)
        t1 = one.last_timing()
    with PKG.Engine(devices=[0, 0, 0]) as m:
        assert m.num_gpus == 3
        m.set_engine(engine)
        m.reserve(nq)
        m.load_data(nodes)
        assert m.n == n and np.array_equal(m.download_data(5, 7).view(np.uint32), nodes[5:12].view(np.uint32))
        for mode in (0, 1):
            m.set_gather(mode)
            ids, d = m.query(queries, 1.0)
            assert np.array_equal(ids, ids1) and np.array_equal(d.view(np.uint32), d1.view(np.uint32)), mode
            t = m.last_timing()
            assert t.n_gpus == 3 and t.nq == nq and t.pairs == t1.pairs and t.engine == t1.engine and t.host_ms > 0
        # resident API: upload / generate, ranges that straddle parts
        m.upload_queries(queries)
        assert np.array_equal(m.download_queries(3000, 1000).view(np.uint32), queries[3000:4000].view(np.uint32))
        m.query_resident(100, 9000, 1.0)
        m.sync()
        ids, d = m.download_results(100, 9000)
        assert np.array_equal(ids, ids1[100:9100]) and np.array_equal(d.view(np.uint32), d1[100:9100].view(np.uint32))
        m.gen_queries(5000, 77, T.GEN_V1, 30, -1, 123)
        assert np.array_equal(m.download_queries(0, 5000).view(np.uint32),
                              T.gen_queries(5000, 77, T.GEN_V1, 30, -1, row0=123).view(np.uint32))
        with pytest.raises(PKG.HvsError):
            m.query_resident(4000, 2000, 1.0)     # outside the resident set
        with pytest.raises(PKG.HvsError):
            m.export_results_device(0, 10, 1)     # device pointers belong to one GPU
        # another k, a sampled prefix of the rows, and the peer gather without distances
        m.set_k(37)
        with PKG.Engine(0) as one:
            one.set_engine(engine)
            one.set_k(37)
            one.load_data(nodes)
            want37 = one.query(queries, 0.6, want_dists=False)
        for mode in (1, 0):
            m.set_gather(mode)
            got37 = m.query(queries, 0.6, want_dists=False)
            assert got37.shape == (nq, 37) and np.array_equal(got37, want37), mode
    ref, _ = T.oracle_query(nodes, queries[:300])
    T.check_parity(nodes, queries[:300], ids1[:300], ref, got_dists=d1[:300])
    with T.oracle_k(37):
        ref37, _ = T.oracle_query(nodes, queries[:200], 0.6)
        T.check_parity(nodes, queries[:200], want37[:200], ref37, sample_proportion=0.6)


def test_host_pipeline_many_pieces_and_caller_buffers():
    """hvs_query is a pipeline of 65536-query pieces (pinned staging, H2D one batch ahead, D2H under the next batch) over
    batches that do not end on piece boundaries (small first and last batch, equal shares between): a call of several
    pieces and several batches must return exactly what the resident path returns."""
    import subprocess
    import sys
    code = r"""
import importlib, sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import hvs_testlib as T
PKG = importlib.import_module('project---hybrid-vector-search-queries_amd')
n, nq = 200_000, 300_000                      # 5 staging pieces; batches of 16384 (first, last) and 3 x ~89k queries between
nodes = T.gen_data(n, 55, T.GEN_V1, 50); queries = T.gen_queries(nq, 56, T.GEN_V1, 50)
queries[70000, 8] = np.inf; queries[299999, 9] = np.inf     # fallback queries in the first and the last piece
with PKG.Engine(0) as e:
    e.load_data(nodes)
    e.upload_queries(queries); e.query_resident(0, nq, 1.0); e.sync()
    want_i, want_d = e.download_results(0, nq)
    ids = np.full((nq, 100), 0xDEADBEEF, np.uint32); d = np.full((nq, 100), -1, np.float32)
    e.query(queries, 1.0, out_ids=ids, out_dists=d)
    t = e.last_timing()
    assert np.array_equal(ids, want_i) and np.array_equal(d.view(np.uint32), want_d.view(np.uint32))
    assert t.nq == nq and t.fallback_queries == 2 and t.host_ms >= t.query_ms > 0
    ids2 = e.query(queries[:70001], 1.0, want_dists=False)
    assert np.array_equal(ids2, want_i[:70001])
sel = np.r_[0:64, 69990:70010, nq - 64:nq]
ref, _ = T.oracle_query(nodes, queries[sel])
T.check_parity(nodes, queries[sel], want_i[sel], ref, got_dists=want_d[sel])
print('SUBPROCESS-OK')
"""
    env = dict(os.environ, HVS_MFMA_BATCH="131072")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=T.REPO)
    assert "SUBPROCESS-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("nq,devices", [(300_000, None), (600_000, None), ((1 << 20) + 1, None), (1_000_000, [0, 0])],
                         ids=["300k", "600k", "2^20+1", "2x500k"])
def test_host_path_stages_every_batch_it_runs(nq, devices):
    """hvs_query at the DEFAULT batch size on an index-eligible D: call sizes between one staging send and a ramped
    schedule (a leaf of an 8-GPU run of BASELINE configs[3] gets 5 x 10^5 queries) and one query past 2^20.  The resident
    query buffer is poisoned with other queries first, so a batch that runs before its input was staged cannot compare
    equal by accident; all ids must equal the resident path's."""
    n = 40_000
    nodes = T.gen_data(n, 61, T.GEN_V1, 20)
    queries = T.gen_queries(nq, 62, T.GEN_V1, 20)
    poison = T.gen_queries(nq, 63, T.GEN_V1, 20)
    with (PKG.Engine(devices=devices) if devices else PKG.Engine(0)) as e:
        e.load_data(nodes)
        e.upload_queries(queries)
        e.query_resident(0, nq, 1.0)
        e.sync()
        want = e.download_results(0, nq, want_dists=False)
        assert e.last_timing().engine in FILTER_ENGINES
        e.upload_queries(poison)                  # the device-side query buffer now holds other rows
        got = np.full((nq, 100), 0xDEADBEEF, np.uint32)
        e.query(queries, 1.0, want_dists=False, out_ids=got)
        assert e.last_timing().nq == nq
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, "first wrong query %d of %d wrong" % (bad[0], bad.size)
    sel = np.r_[0:16, 262140:262160, nq - 16:nq]
    ref, _ = T.oracle_query(nodes, queries[sel])
    T.check_parity(nodes, queries[sel], want[sel], ref)


@pytest.mark.parametrize("profile", [T.GEN_CLUSTER, T.GEN_PCA, T.GEN_HEAVY, T.GEN_V1_OUT], ids=["clustered", "pca", "heavy_tails", "v1_out_of_box"])
def test_nonuniform_vector_laws_parity(profile):
    """Non-uniform vector laws (include/hvs_gen.h; the reference's contest data is clustered / PCA-like, README.md:58-60)
    at n = 10^6, 1 % of the queries outside the data's bounding box: both filter engines bit-equal to the exact engine,
    the oracle confirms a sample incl. out-of-box queries, and an out-of-box query stays in the INT8 filter (it pays a
    wider band: hvs_k_prep_slots) instead of falling back to the exact engine."""
    n, nq = 1_000_000, 8192
    with PKG.Engine(0) as x:
        x.set_engine(PKG.ENGINE_EXACT_SCAN)
        x.gen_data(n, T.SEED_DATA, profile, 100)
        x.gen_queries(nq, T.SEED_QUERY, profile, 100, -1, 0)
        queries = x.download_queries(0, nq)
        nodes = x.download_data(0, n)
        want_i, want_d = x.query(queries, 1.0)
    assert np.array_equal(nodes[:2000].view(np.uint32), T.gen_data(2000, T.SEED_DATA, profile, 100).view(np.uint32))
    assert np.array_equal(queries[:2000].view(np.uint32), T.gen_queries(2000, T.SEED_QUERY, profile, 100).view(np.uint32))
    lo, hi = nodes[:, 2:].min(0), nodes[:, 2:].max(0)
    outside = ((queries[:, 4:] < lo) | (queries[:, 4:] > hi)).any(axis=1)
    assert 0.005 < outside.mean() < 0.02
    for engine in [PKG.ENGINE_AUTO] + FILTER_ENGINES:
        with PKG.Engine(0) as e:
            e.set_engine(engine)
            e.load_data(nodes)
            ids, d = e.query(queries, 1.0)
            t = e.last_timing()
            assert np.array_equal(ids, want_i) and np.array_equal(d.view(np.uint32), want_d.view(np.uint32)), engine
            print("profile", profile, "engine", engine, "ran", t.engine, "fallback", t.fallback_queries, "retried", t.retry_queries,
                  "rescored/query %.0f" % (t.rescored_pairs / nq), "outside the box", int(outside.sum()), "device ms %.1f" % t.query_ms)
            if engine == PKG.ENGINE_AUTO:
                # the planner's probe picks a format whose band fits this data: (almost) nothing is left to the exact engine,
                # out-of-box queries included
                assert t.engine in FILTER_ENGINES and t.fallback_queries <= nq // 100, (t.engine, t.fallback_queries)
                if profile == T.GEN_V1_OUT:   # uniform data: INT8 tiles, and a query outside the box pays a wider band only
                    assert t.engine == PKG.ENGINE_MFMA_I8 and t.fallback_queries <= outside.sum() // 10
    sel = np.r_[np.nonzero(outside)[0][:24], 0:40]
    ref, _ = T.oracle_query(nodes, queries[sel], threads=8)
    T.check_parity(nodes, queries[sel], want_i[sel], ref, got_dists=want_d[sel])


def test_auto_engine_changes_tile_format_when_queries_leave_the_box():
    """HVS_ENGINE_AUTO picks INT8 tiles for evenly filled data (the planner's probe uses rows of D as queries).  A call whose
    queries lie far outside the data's bounding box has no usable INT8 bound for them: instead of the exact engine at 1 % of
    a filter's rate, the 16-bit float tiles are built in mid-call and answer them, and later calls use those tiles."""
    n, nq = 200_000, 6000
    nodes = T.gen_data(n, 81, T.GEN_V1, 10)
    inside = T.gen_queries(nq, 82, T.GEN_V1, 10)
    outside = inside.copy()
    outside[:, 4:] *= np.float32(3.0)                       # every query far outside [-6, 6)^100
    with PKG.Engine(0) as x:
        x.set_engine(PKG.ENGINE_EXACT_SCAN)
        x.load_data(nodes)
        want_in = x.query(inside, 1.0)
        want_out = x.query(outside, 1.0)
    with PKG.Engine(0) as e:
        e.load_data(nodes)                                  # AUTO
        ids, d = e.query(inside, 1.0)
        t = e.last_timing()
        assert t.engine == PKG.ENGINE_MFMA_I8 and t.flags == 0 and t.fallback_queries == 0
        assert np.array_equal(ids, want_in[0]) and np.array_equal(d.view(np.uint32), want_in[1].view(np.uint32))
        ids, d = e.query(outside, 1.0)
        t = e.last_timing()
        print("first call outside the box: engine", t.engine, "flags", t.flags, "exact fallback", t.fallback_queries, "device ms %.1f" % t.query_ms)
        assert t.flags & 2 and t.engine in (PKG.ENGINE_MFMA_F16, PKG.ENGINE_MFMA_FILTER) and t.fallback_queries < nq // 20
        assert np.array_equal(ids, want_out[0]) and np.array_equal(d.view(np.uint32), want_out[1].view(np.uint32))
        ids, d = e.query(outside, 1.0)
        t2 = e.last_timing()
        print("second call: engine", t2.engine, "flags", t2.flags, "exact fallback", t2.fallback_queries, "device ms %.1f" % t2.query_ms)
        assert t2.flags == 0 and t2.engine == t.engine and t2.fallback_queries < nq // 20
        assert np.array_equal(ids, want_out[0]) and np.array_equal(d.view(np.uint32), want_out[1].view(np.uint32))
    ref, _ = T.oracle_query(nodes, outside[:64])
    T.check_parity(nodes, outside[:64], want_out[0][:64], ref, got_dists=want_out[1][:64])


_GUESS_CODE = r"""
import importlib, os, sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import hvs_testlib as T
PKG = importlib.import_module('project---hybrid-vector-search-queries_amd')
n, nq = 400_000, 6000
rng = np.random.default_rng(7)
def run(nodes, queries, tag, want_retry):
    with PKG.Engine(0) as x:
        x.set_engine(PKG.ENGINE_EXACT_SCAN); x.load_data(nodes)
        want_i, want_d = x.query(queries, 1.0)
    for engine in (PKG.ENGINE_MFMA_I8, PKG.ENGINE_MFMA_FILTER, PKG.ENGINE_MFMA_F16):
        with PKG.Engine(0) as e:
            e.set_engine(engine); e.load_data(nodes)
            ids, d = e.query(queries, 1.0)
            t = e.last_timing()
            assert t.engine == engine or engine == PKG.ENGINE_MFMA_F16
            bad = np.nonzero((ids != want_i).any(axis=1) | (d.view(np.uint32) != want_d.view(np.uint32)).any(axis=1))[0]
            assert bad.size == 0, (tag, engine, bad[:8], t.as_dict())
            print(tag, 'engine', engine, 'retried', t.retry_queries, 'exact fallback', t.fallback_queries, 'rescored/query %.0f' % (t.rescored_pairs / nq))
            if want_retry: assert t.retry_queries > 0, tag
            assert len(e.last_reruns(1)) == t.retry_queries and len(e.last_reruns(0)) == t.fallback_queries
    return want_i
# 1. uniform data
nodes = T.gen_data(n, 71, T.GEN_V1, 10); queries = T.gen_queries(nq, 72, T.GEN_V1, 10)
w = run(nodes, queries, 'uniform', os.environ.get('HVS_GUESS_PFAIL') == '1')
ref, _ = T.oracle_query(nodes, queries[:96]); T.check_parity(nodes, queries[:96], w[:96], ref)
# 1b. other k through the retry batches (their candidate capacity follows k (radix - 1))
if os.environ.get('HVS_GUESS_PFAIL') == '1':
    for k in (8, 200):
        with PKG.Engine(0) as x:
            x.set_engine(PKG.ENGINE_EXACT_SCAN); x.set_k(k); x.load_data(nodes)
            wk = x.query(queries[:3000], 1.0, want_dists=False)
        with PKG.Engine(0) as e:
            e.set_engine(PKG.ENGINE_MFMA_I8); e.set_k(k); e.load_data(nodes)
            gk = e.query(queries[:3000], 1.0, want_dists=False)
            t = e.last_timing()
            assert np.array_equal(gk, wk), k
            assert t.retry_queries > 0, k
            print('k', k, 'retried', t.retry_queries, 'exact fallback', t.fallback_queries)
# 2. vectors drift with the timestamp: rows that are neighbours in the T ordering are neighbours in space
drift = nodes.copy(); drift[:, 2:] += 40.0 * drift[:, 1:2] * np.sign(rng.standard_normal(100)).astype(np.float32)
qd = queries.copy(); qd[:, 4:] += 40.0 * rng.random((nq, 1), dtype=np.float32) * np.sign(rng.standard_normal(100)).astype(np.float32)
run(drift, qd, 'drift', False)
# 3. bursts: the true neighbours of a query sit in a few ADJACENT blocks of both orderings (same C, consecutive T), which the
#    rows seen before a level either miss altogether or over-represent
burst = nodes.copy(); qb = queries[:2000].copy()
for j in range(0, 2000, 4):
    rows = rng.integers(0, n - 200)
    c, t0 = float(j % 10), 0.2 + 0.6 * rng.random()
    m = 150
    burst[rows:rows + m, 0] = c
    burst[rows:rows + m, 1] = (t0 + 1e-6 * np.arange(m)).astype(np.float32)
    burst[rows:rows + m, 2:] = qb[j, 4:] + 0.05 * rng.standard_normal((m, 100)).astype(np.float32)
    typ = j // 4 % 4
    qb[j, 0] = typ; qb[j, 1] = c if typ in (1, 3) else -1; qb[j, 2:4] = (0.1, 0.9) if typ >= 2 else (-1, -1)
run(burst, qb, 'burst', False)
print('SUBPROCESS-OK')
"""


@pytest.mark.parametrize("pfail", ["1", "3", "6"])
def test_guessed_thresholds_are_verified_whatever_the_guess(pfail):
    """The filter engines guess each level's threshold from the rows seen so far and verify the answer at the end (csrc/
    hvs_filter.h, "Guessed thresholds"): with a reckless guess (failure target 10^-1: many queries are retried with the proven
    threshold), the default and a timid one; on uniform data, on data whose vectors drift along the T ordering and on data
    whose true neighbours sit in a few adjacent index blocks.  Always bit-equal to the exact engine."""
    import subprocess
    import sys
    env = dict(os.environ, HVS_GUESS_PFAIL=pfail)
    r = subprocess.run([sys.executable, "-c", _GUESS_CODE], capture_output=True, text=True, env=env, cwd=T.REPO)
    print(r.stdout[-1500:])
    assert "SUBPROCESS-OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_config3_host_path_4e6_queries_one_call():
    """BASELINE configs[3], one GPU's view of it: D = 10^7, the whole 4 x 10^6-query set handed to hvs_query as ONE
    call from host memory (reference scope src/test.cpp:82-88: host RAM in, host RAM out), ids identical to the
    device-resident path; the oracle confirms a sample."""
    n, nq = 10_000_000, 4_000_000
    queries = T.gen_queries(nq, T.SEED_QUERY, T.GEN_V1, 100)
    with PKG.Engine(0) as e:
        e.reserve(nq)
        e.gen_data(n, T.SEED_DATA, T.GEN_V1, 100)
        ids = e.query(queries, 1.0, want_dists=False)
        t = e.last_timing()
        assert t.nq == nq and t.engine in FILTER_ENGINES and t.fallback_queries == 0
        print("host->host %.0f ms for %d queries = %.0f queries/s (device %.0f ms, %d filter launches)"
              % (t.host_ms, nq, nq / t.host_ms * 1e3, t.query_ms, t.main_kernel_launches))
        # hvs_query's schedule: 2^18 queries first and last, the rest in equal batches of at most 2^21 -> 4 batches x 4 levels
        # (radices 16, 16, 16, 4 above a level 0 of 19 blocks) + the levels of a retry batch, if any; every launch is timed
        assert t.main_kernel_launches in (4 * 4, 5 * 4), t.main_kernel_launches
        print("retried with a proven threshold:", t.retry_queries, "of", nq)
        e.gen_queries(nq, T.SEED_QUERY, T.GEN_V1, 100, -1, 0)
        e.query_resident(0, nq, 1.0)
        e.sync()
        res = e.download_results(0, nq, want_dists=False)
        assert np.array_equal(ids, res)
        sel = np.arange(0, nq, nq // 48)[:48]
        nodes = e.download_data(0, n)
    # the same call through the multi-GPU context with 8 parts ("virtual ranks" on this GPU): the orchestration an 8-GPU node
    # runs -- slice uploads + peer all-gather of D, 8 host pipelines of 5 x 10^5 queries each writing their slice of the result
    with PKG.Engine(devices=[0] * 8) as m:
        m.reserve(nq)
        m.load_data(nodes)
        ids8 = m.query(queries, 1.0, want_dists=False)
        t8 = m.last_timing()
        assert t8.n_gpus == 8 and t8.nq == nq and t8.fallback_queries == 0
        bad = np.nonzero((ids8 != ids).any(axis=1))[0]
        assert bad.size == 0, "8-part context: first wrong query %d of %d wrong" % (bad[0], bad.size)
        print("8 virtual ranks on one GPU: host->host %.0f ms for %d queries" % (t8.host_ms, nq))
    ref, _ = T.oracle_query(nodes, queries[sel], threads=16)
    T.check_parity(nodes, queries[sel], ids[sel], ref)


def test_d1e8_config4_hbm_sizing():
    """BASELINE configs[4], one GPU's view of it: D = 10^8 rows resident (40.8 GB of rows + the INT8 index) and a rank's REAL share
    of the 10^7-query set on an 8-GPU node -- 1.25 x 10^6 mixed queries -- as ONE hvs_query call from host memory to host memory
    (schedule = hvs_plan_batches(1250000, 1): 2^18 first and last, the rest in between, consecutive batches on two lanes).
    Filter engine, no fallbacks; the ids equal the device-resident path's; sorted distances, predicate / padding properties on
    every answer; bit-equality with the exact-scan engine on 1024 queries and with the oracle on 64 queries per type."""
    n, nq = 100_000_000, 1_250_000
    lib = PKG.library()
    sched = (C.c_uint32 * 16)()
    nb = lib.hvs_plan_batches(nq, 1, sched, 16)
    assert nb >= 3 and sum(sched[:nb]) == nq and sched[0] == sched[nb - 1] == 1 << 18 and max(sched[:nb]) <= 1 << 21, list(sched[:nb])
    with PKG.Engine(0) as e:
        e.reserve(nq)
        e.gen_data(n, T.SEED_DATA, T.GEN_V1, 100)
        e.gen_queries(nq, T.SEED_QUERY + 9, T.GEN_V1, 100, -1, 0)
        queries = e.download_queries(0, nq)
        ids, dists = e.query(queries, 1.0)                         # host -> host, one call
        t = e.last_timing()
        assert t.engine in FILTER_ENGINES and t.fallback_queries == 0 and t.nq == nq
        print("D=1e8: %d queries host->host in %.0f ms (%.0f on the device), %.0f rescored pairs per query, %d retried, %d filter launches"
              % (nq, t.host_ms, t.query_ms, t.rescored_pairs / nq, t.retry_queries, t.main_kernel_launches))
        e.query_resident(0, nq, 1.0)                               # the resident path on the queries the host path left in HBM
        e.sync()
        ids_r, dists_r = e.download_results(0, nq)
        assert np.array_equal(ids, ids_r) and np.array_equal(dists.view(np.uint32), dists_r.view(np.uint32))
        del ids_r, dists_r
        sel = np.arange(0, nq, nq // 1024)[:1024]
        e.set_engine(PKG.ENGINE_EXACT_SCAN)
        ids_x, d_x = e.query(queries[sel], 1.0)
        assert e.last_timing().engine == PKG.ENGINE_EXACT_SCAN
        assert np.array_equal(ids[sel], ids_x) and np.array_equal(dists[sel].view(np.uint32), d_x.view(np.uint32))
        nodes = np.empty((n, 102), np.float32)
        for r0 in range(0, n, 10_000_000):        # 40.8 GB in slices
            nodes[r0:r0 + 10_000_000] = e.download_data(r0, 10_000_000)
    assert ids.max() < n and np.all(np.diff(dists, axis=1) >= 0)
    typ = queries[:, 0].astype(int)
    has_c, has_t = (typ & 1) == 1, (typ & 2) == 2
    notpad = ids < n - 100                                    # padding ids come from the last 100 rows only
    assert np.all((nodes[:, 0][ids[has_c]] == queries[has_c, 1:2])[notpad[has_c]])
    tt = nodes[:, 1][ids[has_t]]
    assert np.all(((tt >= queries[has_t, 2:3]) & (tt <= queries[has_t, 3:4]))[notpad[has_t]])
    for row in ids[typ == 0][::997]:
        assert len(set(row.tolist())) == 100
    pick = np.concatenate([np.nonzero(typ == k)[0][:64] for k in range(4)])           # 256 queries against the oracle
    assert np.array_equal(T.oracle_dists_for_ids(nodes, queries[pick], ids[pick]).view(np.uint32), dists[pick].view(np.uint32))
    ref, _ = T.oracle_query(nodes, queries[pick], threads=32)
    T.check_parity(nodes, queries[pick], ids[pick], ref, got_dists=dists[pick])


def test_more_than_2pow27_rows_run_through_the_filter():
    """Survivor entries carry the block position in 24 bits (22 in round 2, which sent data sets of more than 2^27 rows per GPU to
    the exact engine): 2^27 + 4096 rows generated on the device (54.8 GB of rows + 35 GB of INT8 index), HVS_ENGINE_AUTO answers
    with the filter engine, bit-equal to the exact engine; the returned distances are recomputed on the host from the
    returned rows."""
    n, nq = (1 << 27) + 4096, 2048
    with PKG.Engine(0) as e:
        e.gen_data(n, T.SEED_DATA, T.GEN_V1, 100)
        e.gen_queries(nq, T.SEED_QUERY + 3, T.GEN_V1, 100, -1, 0)
        queries = e.download_queries(0, nq)
        e.query_resident(0, nq, 1.0)
        e.sync()
        t = e.last_timing()
        ids, dists = e.download_results(0, nq)
        assert t.engine in FILTER_ENGINES and t.flags == 0 and t.nq == nq and t.fallback_queries == 0
        assert ids.max() < n and np.all(np.diff(dists, axis=1) >= 0)
        e.set_engine(PKG.ENGINE_EXACT_SCAN)
        e.query_resident(0, 64, 1.0)
        e.sync()
        assert e.last_timing().engine == PKG.ENGINE_EXACT_SCAN
        xi, xd = e.download_results(0, 64)
        assert np.array_equal(xi, ids[:64]) and np.array_equal(xd.view(np.uint32), dists[:64].view(np.uint32))
        for qi in (0, 17, 40, 63, 2047):
            rows = np.stack([e.download_data(int(r), 1)[0] for r in ids[qi]])
            want = T.oracle_dists_for_ids(rows, queries[qi:qi + 1], np.arange(100, dtype=np.uint32)[None, :])
            assert np.array_equal(want.view(np.uint32), dists[qi:qi + 1].view(np.uint32))
            typ = int(queries[qi, 0])
            notpad = ids[qi] < n - 100
            if typ & 1:
                assert np.all((rows[:, 0] == queries[qi, 1])[notpad])
            if typ & 2:
                assert np.all(((rows[:, 1] >= queries[qi, 2]) & (rows[:, 1] <= queries[qi, 3]))[notpad])


@pytest.mark.parametrize("engine", [PKG.ENGINE_EXACT_SCAN] + FILTER_ENGINES, ids=["exact"] + FILTER_IDS)
@pytest.mark.parametrize("path", KGOLDENS, ids=[os.path.basename(p)[:-4] for p in KGOLDENS])
def test_other_k_matches_reference_built_with_that_k(path, engine):
    """hvs_set_k: k = 8, 10 and 256 against output.bin of the reference compiled with that KNN_LIMIT
    (include/optimized_impl.h:26), serial and threaded engine, and against the oracle with the same k."""
    z = np.load(path)
    nodes, queries = _inputs(z)
    k = int(z["k"])
    with PKG.Engine(0) as e:
        e.set_engine(engine)
        e.set_k(k)
        assert e.k == k
        e.load_data(nodes)
        ids, dists = e.query(queries, 1.0)
        t = e.last_timing()
        assert ids.shape == (queries.shape[0], k) and t.engine == engine
    with T.oracle_k(k):
        T.check_parity(nodes, queries, ids, z["ids_optimized"], got_dists=dists)
        T.check_parity(nodes, queries, ids, z["ids_optimized_parallel"])
        ref, refd = T.oracle_query(nodes, queries)
        assert np.array_equal(ids, ref) and np.array_equal(dists.view(np.uint32), refd.view(np.uint32))


@pytest.mark.parametrize("k", [8, 37, 128, 129, 200, 256])
def test_k_sweep_all_engines_agree_with_the_oracle(k):
    """k on both sides of the 128-key list boundary, ragged (37, 129, 200): the filter engines (levels, re-scoring,
    merges) and the exact engine must return the oracle's bits; changing k on a live context drops old results and
    keeps the data; n < k is refused."""
    n, nq = 90_000, 260
    nodes = T.gen_data(n, 600 + k, T.GEN_V1, 60)
    queries = T.gen_queries(nq, 700 + k, T.GEN_V1, 60)
    queries[0, :4] = [3, 7, 0.5, 0.5003]      # a handful of rows: mostly padding
    queries[1, :4] = [1, 999, -1, -1]         # no row at all: k pad rows
    with T.oracle_k(k):
        ref, refd = T.oracle_query(nodes, queries)
        ref_half, refd_half = T.oracle_query(nodes, queries[:64], 0.5)
    with PKG.Engine(0) as e:
        e.load_data(nodes)                     # loaded at the default k = 100 ...
        ids100, _ = e.query(queries[:8], 1.0)
        assert ids100.shape == (8, 100)
        e.set_k(k)                             # ... then k changes under it
        for engine in (PKG.ENGINE_AUTO, PKG.ENGINE_MFMA_FILTER, PKG.ENGINE_EXACT_SCAN):
            e.set_engine(engine)
            ids, d = e.query(queries, 1.0)
            assert e.last_timing().fallback_queries == 0
            assert np.array_equal(ids, ref) and np.array_equal(d.view(np.uint32), refd.view(np.uint32)), (k, engine)
            ids, d = e.query(queries[:64], 0.5)
            assert np.array_equal(ids, ref_half) and np.array_equal(d.view(np.uint32), refd_half.view(np.uint32)), (k, engine)
        with pytest.raises(PKG.HvsError):
            e.set_k(7)
        with pytest.raises(PKG.HvsError):
            e.set_k(257)
    with PKG.Engine(0) as e:
        e.set_k(256)
        with pytest.raises(PKG.HvsError):
            e.load_data(T.gen_data(255))       # n < k: the reference's padding index would underflow
    with PKG.Engine(devices=[0, 0]) as m:      # multi-GPU context: k reaches every part
        m.set_k(k)
        m.load_data(nodes)
        ids, d = m.query(queries, 1.0)
        assert np.array_equal(ids, ref) and np.array_equal(d.view(np.uint32), refd.view(np.uint32))


@pytest.mark.parametrize("k", [16, 200])
def test_data_sharded_mode_with_another_k(k):
    """D-sharded mode with k != 100: partial top-k lists of 3 row shards (padding off), one padding pass
    from the tail of the whole set, merged on the host (sharding.merge_data_shards) = the oracle's answer at that k."""
    sharding = importlib.import_module("project---hybrid-vector-search-queries_amd.sharding")
    n, nq, world = 120_000, 200, 3
    nodes = T.gen_data(n, 141, T.GEN_V1, 40)
    queries = T.gen_queries(nq, 142, T.GEN_V1, 40)
    queries[0, :4] = [3, 7, 0.5, 0.5001]       # a few rows only -> padding from the global tail
    queries[1, :4] = [1, 999, -1, -1]          # no row at all
    parts = []
    for r in range(world):
        r0, r1 = sharding.row_shard_range(n, r, world)
        with PKG.Engine(0) as e:
            e.set_k(k)
            e.set_padding(False)
            e.load_data(nodes[r0:r1])
            ids, dists = e.query(queries, 1.0)
        assert ids.shape == (nq, k)
        parts.append((ids, dists, r0))
    with PKG.Engine(0) as e:
        e.set_k(k)
        e.load_data(nodes[n - k:])               # the last k rows: every query sees all of them
        pad = sharding.tail_pad_dists(lambda q: e.query(q, 1.0), queries)
    ids, dists = sharding.merge_data_shards(parts, n, pad)
    with T.oracle_k(k):
        ref, refd = T.oracle_query(nodes, queries)
    assert np.array_equal(ids, ref) and np.array_equal(dists.view(np.uint32), refd.view(np.uint32))


@pytest.mark.parametrize("devices", [None, [0, 0]], ids=["one_gpu", "two_virtual_ranks"])
def test_format_change_then_a_smaller_call_on_the_same_context(devices):
    """the list of queries a mid-call change of tile format moved aside must die with its call.  A call far
    outside the data's box (FORMAT_CHANGED), then a much smaller call on the same context: the small call's rows -- and nothing
    past them: the result arrays sit inside guarded buffers -- equal the exact engine's."""
    n, nq, small = 200_000, 6000, 700
    nodes = T.gen_data(n, 81, T.GEN_V1, 10)
    inside = T.gen_queries(nq, 82, T.GEN_V1, 10)
    outside = inside.copy()
    outside[:, 4:] *= np.float32(3.0)
    with PKG.Engine(0) as x:
        x.set_engine(PKG.ENGINE_EXACT_SCAN)
        x.load_data(nodes)
        want_out = x.query(outside, 1.0)
        want_small = x.query(inside[:small], 1.0)
    with (PKG.Engine(devices=devices) if devices else PKG.Engine(0)) as e:
        e.load_data(nodes)
        ids, d = e.query(outside, 1.0)
        t = e.last_timing()
        assert t.flags & 2, t.as_dict()
        assert np.array_equal(ids, want_out[0]) and np.array_equal(d.view(np.uint32), want_out[1].view(np.uint32))
        guard = 4096
        buf_i = np.full((small + guard, 100), 0xDEADBEEF, np.uint32)
        buf_d = np.full((small + guard, 100), -7.0, np.float32)
        e.query(inside[:small], 1.0, out_ids=buf_i[:small], out_dists=buf_d[:small])
        t2 = e.last_timing()
        assert t2.flags == 0 and t2.nq == small
        assert np.array_equal(buf_i[:small], want_small[0]) and np.array_equal(buf_d[:small].view(np.uint32), want_small[1].view(np.uint32))
        assert np.all(buf_i[small:] == 0xDEADBEEF) and np.all(buf_d[small:] == -7.0), "a stale list of the earlier call was scattered past the result"


@pytest.mark.parametrize("rotate", ["", "1"], ids=["planner", "rotated_int8"])
def test_one_context_many_calls_fuzz(rotate, monkeypatch):
    """per-context state survives between calls (tile format, guess tables keyed by k, candidate capacities, lists of
    re-run queries).  ONE context answers a sequence of calls with varying nq, k, sample_proportion, engine and API (host /
    resident), with a call far outside the data's box in the middle; every call is compared bit for bit with a fresh exact
    engine."""
    if rotate:
        monkeypatch.setenv("HVS_I8_ROTATE", rotate)   # (read per data set: INT8 tiles of this context are cut from rotated vectors)
    else:
        monkeypatch.delenv("HVS_I8_ROTATE", raising=False)
    n = 150_000
    rng = np.random.default_rng(20261005)
    nodes = T.gen_data(n, 91, T.GEN_V1, 10)
    pool = T.gen_queries(40_000, 92, T.GEN_V1, 10)
    far = pool[:5000].copy()
    far[:, 4:] *= np.float32(3.0)
    plan = []
    for step in range(14):
        nq = int(rng.choice([1, 37, 512, 3000, 9000, 20000]))
        q0 = int(rng.integers(0, len(pool) - nq))
        plan.append(dict(q=pool[q0:q0 + nq], k=int(rng.choice([100, 100, 100, 10, 200])), sp=float(rng.choice([1.0, 1.0, 0.6, 0.3])),
                         engine=int(rng.choice([0, 0, 0, 3, 4, 2, 1])), resident=bool(rng.integers(0, 2))))
    plan.insert(5, dict(q=far, k=100, sp=1.0, engine=0, resident=False))           # FORMAT_CHANGED in mid-sequence
    plan.insert(6, dict(q=pool[100:100 + 300], k=100, sp=1.0, engine=0, resident=False))
    with PKG.Engine(0) as x, PKG.Engine(0) as e:
        x.set_engine(PKG.ENGINE_EXACT_SCAN)
        x.load_data(nodes)
        e.load_data(nodes)
        for i, st in enumerate(plan):
            x.set_k(st["k"])
            want_i, want_d = x.query(st["q"], st["sp"])
            e.set_k(st["k"])
            e.set_engine(st["engine"])
            if st["resident"]:
                e.upload_queries(st["q"])
                e.query_resident(0, len(st["q"]), st["sp"])
                e.sync()
                ids, d = e.download_results(0, len(st["q"]))
            else:
                ids, d = e.query(st["q"], st["sp"])
            t = e.last_timing()
            bad = np.nonzero((ids != want_i).any(axis=1) | (d.view(np.uint32) != want_d.view(np.uint32)).any(axis=1))[0]
            assert bad.size == 0, (i, {k: v for k, v in st.items() if k != "q"}, len(st["q"]), bad[:8], t.as_dict())
            print("call", i, "nq", len(st["q"]), {k: v for k, v in st.items() if k != "q"}, "ran", t.engine, "flags", t.flags,
                  "retried", t.retry_queries, "exact", t.fallback_queries)


@pytest.mark.parametrize("profile", [T.GEN_V1, T.GEN_CLUSTER, T.GEN_PCA, T.GEN_HEAVY, T.GEN_V1_OUT], ids=["v1", "clustered", "pca", "heavy_tails", "v1_out_of_box"])
def test_rotated_int8_tiles_parity(profile):
    """Round 4: INT8 tiles cut from the ROTATED vectors (signs + 128-point Walsh-Hadamard transform, csrc/hvs_filter.h HvsQuant) --
    forced with HVS_I8_ROTATE=1 on every vector law, queries outside the data's box included -- answer bit for bit what the exact
    engine answers; on PCA-like data they re-score fewer pairs than the plain INT8 tiles (why the planner's probe may pick them)."""
    n, nq = 400_000, 6144
    with PKG.Engine(0) as x:
        x.set_engine(PKG.ENGINE_EXACT_SCAN)
        x.gen_data(n, T.SEED_DATA, profile, 100)
        x.gen_queries(nq, T.SEED_QUERY, profile, 100, -1, 0)
        queries = x.download_queries(0, nq)
        nodes = x.download_data(0, n)
        want_i, want_d = x.query(queries, 1.0)
    pairs = {}
    old = os.environ.get("HVS_I8_ROTATE")
    try:
        for rot in ("0", "1"):
            os.environ["HVS_I8_ROTATE"] = rot
            with PKG.Engine(0) as e:
                e.set_engine(PKG.ENGINE_MFMA_I8)
                e.load_data(nodes)
                ids, d = e.query(queries, 1.0)
                t = e.last_timing()
                assert t.engine == PKG.ENGINE_MFMA_I8 and bool(t.flags & 4) == (rot == "1"), (rot, t.as_dict())
                bad = np.nonzero((ids != want_i).any(axis=1) | (d.view(np.uint32) != want_d.view(np.uint32)).any(axis=1))[0]
                assert bad.size == 0, (rot, bad[:8], t.as_dict())
                pairs[rot] = t.rescored_pairs / nq
                print("profile", profile, "rotated" if rot == "1" else "plain", "rescored/query %.0f" % pairs[rot], "retried", t.retry_queries,
                      "exact", t.fallback_queries)
                # a second, smaller call on the same context (resident path)
                e.upload_queries(queries[:777])
                e.query_resident(0, 777, 1.0)
                e.sync()
                ids2, d2 = e.download_results(0, 777)
                assert np.array_equal(ids2, want_i[:777]) and np.array_equal(d2.view(np.uint32), want_d[:777].view(np.uint32))
    finally:
        if old is None:
            os.environ.pop("HVS_I8_ROTATE", None)
        else:
            os.environ["HVS_I8_ROTATE"] = old
    if profile == T.GEN_PCA:
        assert pairs["1"] < 0.7 * pairs["0"], pairs
    ref, _ = T.oracle_query(nodes, queries[:48], threads=8)
    T.check_parity(nodes, queries[:48], want_i[:48], ref, got_dists=want_d[:48])


_LANES_CODE = r"""
import importlib, os, sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import hvs_testlib as T
PKG = importlib.import_module('project---hybrid-vector-search-queries_amd')
n, nq = 300_000, 40_000                     # HVS_MFMA_BATCH=4096: calls of 10 batches, alternating between the two lanes
nodes = T.gen_data(n, 61, T.GEN_V1, 10); queries = T.gen_queries(nq, 62, T.GEN_V1, 10)
with PKG.Engine(0) as x:
    x.set_engine(PKG.ENGINE_EXACT_SCAN); x.load_data(nodes)
    want_i, want_d = x.query(queries, 1.0)
for engine in (PKG.ENGINE_AUTO, PKG.ENGINE_MFMA_I8, PKG.ENGINE_MFMA_F16, PKG.ENGINE_MFMA_FILTER):
    with PKG.Engine(0) as e:
        e.set_engine(engine); e.load_data(nodes)
        for rep in range(2):                # the second call finds both lanes' workspaces (or the fallback decision) in place
            ids, d = e.query(queries, 1.0)
            t = e.last_timing()
            assert np.array_equal(ids, want_i) and np.array_equal(d.view(np.uint32), want_d.view(np.uint32)), (engine, rep)
        e.upload_queries(queries); e.query_resident(1000, 30_000, 1.0); e.sync()
        ri, rd = e.download_results(1000, 30_000)
        assert np.array_equal(ri, want_i[1000:31000]) and np.array_equal(rd.view(np.uint32), want_d[1000:31000].view(np.uint32)), engine
        print('engine', engine, 'ran', t.engine, 'launches', t.main_kernel_launches, 'retried', t.retry_queries)
ref, _ = T.oracle_query(nodes, queries[:64]); T.check_parity(nodes, queries[:64], want_i[:64], ref, got_dists=want_d[:64])
print('SUBPROCESS-OK')
"""


@pytest.mark.parametrize("mode", ["two_lanes", "one_lane", "no_room_for_the_spare_lane"])
def test_calls_of_many_batches_on_two_lanes(mode):
    """Round 4: consecutive batches of a call run on two lanes (stream + workspace each, gated behind each other's last filter
    launches; csrc/hvs.hip HvsLane).  Small batches (HVS_MFMA_BATCH=4096) make every call ten batches long: host path twice,
    resident path on an inner range, all filter engines, bit-equal to the exact engine -- with both lanes, with HVS_LANES=0, and
    when the spare lane's workspace cannot be allocated (HVS_TEST_NO_SPARE=1: the call must continue on one lane, silently)."""
    import subprocess
    import sys
    env = dict(os.environ, HVS_MFMA_BATCH="4096")
    if mode == "one_lane":
        env["HVS_LANES"] = "0"
    if mode == "no_room_for_the_spare_lane":
        env["HVS_TEST_NO_SPARE"] = "1"
    r = subprocess.run([sys.executable, "-c", _LANES_CODE], capture_output=True, text=True, env=env, cwd=T.REPO)
    print(r.stdout[-800:])
    assert "SUBPROCESS-OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_planner_picks_rotated_int8_tiles_for_pca_like_data_at_d1e7():
    """Round 4, the planner end to end at full size: D = 10^7 PCA-like rows (include/hvs_gen.h HVS_GEN_PCA) under HVS_ENGINE_AUTO.  The
    probe finds the plain INT8 band too wide (every probe query overflows its lists), tries the rotated INT8 tiles and the FP16 tiles
    and keeps the rotated ones (hvs_timing.flags & 4); 2^18 mixed queries (1 % outside the data's box) run without exact-engine
    fallbacks beyond a handful, and a sample is bit-equal to the exact engine and confirmed by the oracle."""
    n, nq = 10_000_000, 1 << 18
    with PKG.Engine(0) as e:
        e.gen_data(n, T.SEED_DATA, T.GEN_PCA, 100)
        e.gen_queries(nq, T.SEED_QUERY, T.GEN_PCA, 100, -1, 0)
        queries = e.download_queries(0, nq)
        e.query_resident(0, nq, 1.0)
        e.sync()
        t = e.last_timing()
        ids, dists = e.download_results(0, nq)
        print("PCA-like D=1e7: engine", t.engine, "flags", t.flags, "device ms %.0f" % t.query_ms, "rescored/query %.0f" % (t.rescored_pairs / nq),
              "retried", t.retry_queries, "exact", t.fallback_queries)
        assert t.engine == PKG.ENGINE_MFMA_I8 and (t.flags & 4), t.as_dict()
        assert t.fallback_queries <= nq // 1000
        assert np.all(np.diff(dists, axis=1) >= 0) and ids.max() < n
        sel = np.arange(0, nq, nq // 512)[:512]
        e.set_engine(PKG.ENGINE_EXACT_SCAN)
        ids_x, d_x = e.query(queries[sel], 1.0)
        assert np.array_equal(ids[sel], ids_x) and np.array_equal(dists[sel].view(np.uint32), d_x.view(np.uint32))
        nodes = e.download_data(0, n)
    pick = sel[:32]
    ref, _ = T.oracle_query(nodes, queries[pick], threads=16)
    T.check_parity(nodes, queries[pick], ids[pick], ref, got_dists=dists[pick])
