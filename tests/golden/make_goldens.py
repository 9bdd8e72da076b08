#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference binaries (oracle/_ref/,
built by oracle/Makefile from /root/reference) on gen-v1 inputs.

Run in the build container only (the reference does not exist on the GPU box):
    python tests/golden/make_goldens.py [--only NAME] [--big]

Each fixture holds data only: generator parameters, SHA-256 of the generated D/Q
rows, and the reference engines' output.bin ids (uint32).  Inputs are regenerated
from the parameters by the tests (tests/hvs_testlib.py gen_data/gen_queries).
"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvs_testlib as T  # noqa: E402

CASES = {
    # BASELINE.json configs[0] size: the "default provided set" shape, all four query types
    "config1_10k_x100": dict(n=10_000, nq=100, profile=T.GEN_V1, ncat=100, force_type=-1,
                             engines=["baseline", "optimized", "optimized_parallel"]),
    # small integer-categorical set: low-selectivity queries fall into the padding path
    # (optimized_parallel.hpp:149-157) and produce duplicate ids
    "pad_2k_x200": dict(n=2_000, nq=200, profile=T.GEN_V1, ncat=50, force_type=-1,
                        engines=["baseline", "optimized", "optimized_parallel"]),
    # the reference generators' own value ranges: types 1/3 match nothing
    "v0_5k_x64": dict(n=5_000, nq=64, profile=T.GEN_V0, ncat=100, force_type=-1,
                      engines=["baseline", "optimized", "optimized_parallel"]),
    # exactly 100 rows: every query is all rows (+ nothing to pad)
    "tiny_100_x16": dict(n=100, nq=16, profile=T.GEN_V1, ncat=4, force_type=-1,
                         engines=["baseline", "optimized", "optimized_parallel"]),
    # multi-threaded reference path (thread_n = min(hw, sn/100000) > 1, optimized_parallel.hpp:76-77)
    "d400k_x48": dict(n=400_000, nq=48, profile=T.GEN_V1, ncat=100, force_type=-1,
                      engines=["optimized", "optimized_parallel"]),
}
BIG_CASES = {
    # BASELINE.json configs[2] data size with a query slice
    "d1m_x256": dict(n=1_000_000, nq=256, profile=T.GEN_V1, ncat=100, force_type=-1,
                     engines=["optimized", "optimized_parallel"]),
}


def run_case(name, spec, outdir):
    nodes = T.gen_data(spec["n"], T.SEED_DATA, spec["profile"], spec["ncat"])
    queries = T.gen_queries(spec["nq"], T.SEED_QUERY, spec["profile"], spec["ncat"], spec["force_type"])
    rec = dict(n=spec["n"], nq=spec["nq"], profile=spec["profile"], ncat=spec["ncat"],
               force_type=spec["force_type"], seed_data=T.SEED_DATA, seed_query=T.SEED_QUERY,
               sha256_data=T.sha256_of(nodes), sha256_queries=T.sha256_of(queries))
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        dpath, qpath = os.path.join(tmp, "d.bin"), os.path.join(tmp, "q.bin")
        T.write_bin(dpath, nodes)
        T.write_bin(qpath, queries)
        for eng in spec["engines"]:
            exe = os.path.join(T.REF_DIR, eng + ".out")
            opath = os.path.join(tmp, eng + ".bin")
            r = subprocess.run([exe, dpath, qpath, opath], capture_output=True, text=True, check=True)
            took = [ln for ln in r.stderr.splitlines() if "Vector Search took" in ln]
            thr = [ln for ln in r.stderr.splitlines() if ln.startswith("Using ")]
            print(f"  {name}/{eng}: {took[0] if took else '?'} {thr[0] if thr else ''}")
            rec["ids_" + eng] = T.read_knn(opath)
            rec["distfile_" + eng] = T.read_dist_file(opath + ".dist")
            assert rec["ids_" + eng].shape == (spec["nq"], T.K)
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only")
    ap.add_argument("--big", action="store_true")
    a = ap.parse_args()
    T.build_oracle()
    cases = dict(CASES)
    if a.big:
        cases.update(BIG_CASES)
    for name, spec in cases.items():
        if a.only and a.only != name:
            continue
        print(name)
        run_case(name, spec, T.GOLDEN_DIR)


if __name__ == "__main__":
    main()
