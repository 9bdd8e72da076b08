#!/usr/bin/env python3
"""Golden vectors for k != 100: the reference's k is the compile-time constant KNN_LIMIT
(include/optimized_impl.h:26), with K = 100 repeated in its file writer (include/io.h:27,54,93-94) and driver
(src/test.cpp:102).  This script makes a throw-away copy of the reference under /tmp, changes those constants with
sed, compiles the serial and the parallel engine with the reference's flags (CMakeLists.txt:8) and runs them on
gen-v1 inputs.  Only DATA is committed (tests/golden/k*.npz: generator parameters, input hashes, output ids); the
modified copy never enters the repository.

Run in the build container only:  python tests/golden/make_goldens_k.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvs_testlib as T  # noqa: E402

REF = "/root/reference"
CASES = {
    # k = 10 with low-selectivity queries (padding path: fewer than 10 matches) and a multi-threaded parallel run
    "k10_300k_x96": dict(k=10, n=300_000, nq=96, ncat=2000),
    # k = 256 > one 128-key list: types 1/3 match fewer than 256 rows -> padding + duplicates
    "k256_50k_x96": dict(k=256, n=50_000, nq=96, ncat=100),
    # k = 8: the smallest k the reference accepts (static_assert(KNN_LIMIT >= 8))
    "k8_5k_x64": dict(k=8, n=5_000, nq=64, ncat=20),
}


def build(k, tmp):
    src = os.path.join(tmp, f"ref_k{k}")
    shutil.copytree(REF, src, ignore=shutil.ignore_patterns(".git", "report", "presentation"))
    subprocess.run(["sed", "-i", f"s/constexpr size_t KNN_LIMIT = 100;/constexpr size_t KNN_LIMIT = {k};/",
                    os.path.join(src, "include", "optimized_impl.h")], check=True)
    subprocess.run(["sed", "-i", f"s/const int K = 100;/const int K = {k};/; s/resize(100)/resize({k})/; s/j < 100/j < {k}/",
                    os.path.join(src, "include", "io.h")], check=True)
    subprocess.run(["sed", "-i", f"s/j < 100/j < {k}/", os.path.join(src, "src", "test.cpp")], check=True)
    assert f"KNN_LIMIT = {k};" in open(os.path.join(src, "include", "optimized_impl.h")).read()
    exes = {}
    for name, impl in (("optimized", 2), ("optimized_parallel", 3)):
        exe = os.path.join(tmp, f"{name}_k{k}.out")
        subprocess.run(["g++", "-std=gnu++20", "-O3", "-DNDEBUG", "-mavx2", f"-I{src}", f"-I{src}/include", f"-DIMPL={impl}",
                        os.path.join(src, "src", "test.cpp"), "-o", exe, "-lpthread"], check=True)
        exes[name] = exe
    return exes


def main():
    T.build_oracle()
    for name, spec in CASES.items():
        k = spec["k"]
        nodes = T.gen_data(spec["n"], T.SEED_DATA, T.GEN_V1, spec["ncat"])
        queries = T.gen_queries(spec["nq"], T.SEED_QUERY, T.GEN_V1, spec["ncat"])
        rec = dict(k=k, n=spec["n"], nq=spec["nq"], profile=T.GEN_V1, ncat=spec["ncat"], force_type=-1,
                   seed_data=T.SEED_DATA, seed_query=T.SEED_QUERY, sha256_data=T.sha256_of(nodes),
                   sha256_queries=T.sha256_of(queries))
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            exes = build(k, tmp)
            dpath, qpath = os.path.join(tmp, "d.bin"), os.path.join(tmp, "q.bin")
            T.write_bin(dpath, nodes)
            T.write_bin(qpath, queries)
            for eng, exe in exes.items():
                opath = os.path.join(tmp, eng + ".bin")
                r = subprocess.run([exe, dpath, qpath, opath], capture_output=True, text=True, check=True)
                thr = [ln for ln in r.stderr.splitlines() if ln.startswith("Using ")]
                rec["ids_" + eng] = np.fromfile(opath, np.uint32).reshape(spec["nq"], k)
                print(f"  {name}/{eng}: {rec['ids_' + eng].shape} {thr[0] if thr else ''}")
        np.savez_compressed(os.path.join(T.GOLDEN_DIR, name + ".npz"), **rec)
        print("wrote", name)


if __name__ == "__main__":
    main()
