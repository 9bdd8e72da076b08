#!/usr/bin/env python3
"""bench.py -- whole-job queries/second of the filtered brute-force k-NN hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU).

Workload (BASELINE.json metric: queries/sec + recall@100 on D=10^7, Q=4x10^6, dim=100, k=100):
D = 10^7 gen-v1 rows replicated in every GPU's HBM; the 4x10^6-query set is streamed in batches (2^20 by default:
warmup + 3 steps = the whole query set);
one STEP = one pass of the hot path over one batch of `--batch` mixed-type queries per GPU
(inputs resident in HBM when the timed region starts, result ids gathered to rank 0 over RCCL
inside the timed region when N > 1).  Queries shard across ranks with no data-path collective
(weak scaling: per-GPU work is fixed).  value = queries all ranks answered / max-over-ranks time.

Extra JSON objects: "roofline" for the dominant kernel (HIP-event kernel time measured live on the
library's own stream) and "cpu_baseline" (the oracle's reference-faithful threaded engine timed on
this host's cores over a bounded query sample, rank 0 at N=1 only), plus "recall_at_100" of the
GPU answers against that oracle sample.
"""
import argparse
import importlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector = FP32 matrix peak
BF16_PEAK_TFLOPS = 2500.0  # dense BF16 MFMA peak
INT8_PEAK_TOPS = 5000.0    # dense INT8 MFMA peak (2x BF16 per clock)
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=10_000_000, help="rows of D")
    ap.add_argument("--batch", type=int, default=1048576, help="queries per step per GPU")
    ap.add_argument("--force-type", type=int, default=-1, help="-1 mixed types, 0..3 a single type")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--engine", type=int, default=0)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the result gather even with one rank (rehearsal)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ["NCCL_DEBUG"] = "NONE"  # RCCL logs (version banner, WARN lines) go to stdout: rank 0 prints ONE JSON line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("project---hybrid-vector-search-queries_amd")  # after torch: one HIP runtime
    import hvs_testlib as T

    K = 100
    total_batches = a.warmup + a.steps
    eng = pkg.Engine(local_rank)
    if a.engine:
        eng.set_engine(a.engine)
    t0 = time.time()
    eng.gen_data(a.n, T.SEED_DATA, T.GEN_V1, 100)                      # D replicated per GPU
    load_s = time.time() - t0
    # Q is partitioned: this rank owns one contiguous range of the query stream (sharding.shard_range)
    sharding = importlib.import_module("project---hybrid-vector-search-queries_amd.sharding")
    q_first, q_last = sharding.shard_range(world * a.batch * total_batches, rank, world)
    assert q_last - q_first == a.batch * total_batches
    eng.gen_queries(a.batch * total_batches, T.SEED_QUERY, T.GEN_V1, 100, a.force_type, first_row=q_first)

    ids_dev = torch.empty((a.batch, K), dtype=torch.int32, device="cuda")
    gathered = torch.empty((world * a.batch, K), dtype=torch.int32, device="cuda") if use_dist else None

    def step(b):
        eng.query_resident(b * a.batch, a.batch, 1.0)                  # asynchronous on the library's stream
        if use_dist:
            # the previous step's gather (RCCL's stream) ran under this step's compute; it must be done before
            # its send buffer is overwritten
            torch.cuda.current_stream().synchronize()
            eng.export_results_device(b * a.batch, a.batch, ids_dev.data_ptr())
            eng.sync()
            dist.all_gather_into_tensor(gathered, ids_dev)             # result gather over xGMI (RCCL), asynchronous
        else:
            eng.sync()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for b in range(a.warmup):
        step(b)
    kern_ms, kern_launches, pairs, scanned, query_ms, rescored, fallback = 0.0, 0, 0, 0, 0.0, 0, 0
    fence()
    t0 = time.perf_counter()
    for b in range(a.warmup, total_batches):
        step(b)
        tm = eng.last_timing()                                          # HIP events on the library's stream
        kern_ms += tm.main_kernel_ms
        kern_launches += tm.main_kernel_launches
        pairs += tm.pairs
        scanned += tm.scanned_pairs
        query_ms += tm.query_ms
        rescored += tm.rescored_pairs
        fallback += tm.fallback_queries
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    engine_id = int(eng.last_timing().engine)

    out = None
    if rank == 0:
        nq_total = world * a.batch * a.steps
        value = nq_total / elapsed
        flops_alg = 200.0 * pairs                                      # 2*100 per passing pair
        k_s = kern_ms / 1e3
        peak = {2: BF16_PEAK_TFLOPS, 3: INT8_PEAK_TOPS}.get(engine_id, FP32_PEAK_TFLOPS)
        achieved = flops_alg / k_s / 1e12 if k_s > 0 else 0.0
        traffic = None
        tf = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf))
                key = f"n{a.n}_b{a.batch}_t{a.force_type}_e{engine_id}"
                traffic = rec.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "queries/sec (whole node) + recall@100, D=10^7 Q=4x10^6 dim=100",
            "value": value, "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"D={a.n} rows x dim 100 replicated per GPU; step = {a.batch} "
                                   f"{'mixed-type' if a.force_type < 0 else 'type-%d' % a.force_type} queries per GPU "
                                   f"from the gen-v1 4x10^6-query stream, k=100, sample_proportion=1",
                       "n": a.n, "queries_per_step_per_gpu": a.batch, "engine": engine_id,
                       "sharding": "Q partitioned across ranks, D replicated, RCCL all_gather of ids"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "kernel": {2: "hvs_k_filter_mfma<bf16>", 3: "hvs_k_filter_mfma<int8>"}.get(engine_id, "hvs_k_scan_exact"),
                         "peak_is": {2: "dense BF16 MFMA", 3: "dense INT8 MFMA (integer ops)"}.get(engine_id, "FP32 vector"),
                         "kernel_ms_avg": kern_ms / max(kern_launches, 1), "launches": kern_launches,
                         "pairs_per_launch": pairs / max(kern_launches, 1),
                         "evaluated_pairs_per_launch": scanned / max(kern_launches, 1),
                         "device_query_ms_per_step": query_ms / a.steps,
                         "rescored_pairs_per_query": rescored / max(a.batch * a.steps, 1),
                         # context, not the peak: a bare loop of the same MFMA chains sustains this much on random
                         # operands on this chip (DVFS; scripts/mfma_loop_lab.hip / mfma_i8_lab.hip -DLAB_RANDOM)
                         "measured_mfma_ceiling_random_operands_tflops": {2: 1592.0, 3: 3444.0}.get(engine_id),
                         "fallback_queries": fallback},
            "load_s": load_s,
        }

    # ---- CPU baseline + recall leg (rank 0, N=1 only; the oracle is the checker, never the product)
    if rank == 0 and world == 1 and a.cpu_seconds > 0:
        nodes = eng.download_data(0, a.n)
        b = a.warmup                                                   # first timed batch
        hw = os.cpu_count() or 1
        probe = 8
        q = eng.download_queries(b * a.batch, min(a.batch, 4096))
        t1 = time.perf_counter()
        T.oracle_query(nodes, q[:probe], engine="knn", part_threads=0, hw_threads=hw, run_parallel=True)
        per_q = (time.perf_counter() - t1) / probe
        m = int(max(probe, min(q.shape[0], a.cpu_seconds / max(per_q, 1e-6))))
        t1 = time.perf_counter()
        ref_ids, _ = T.oracle_query(nodes, q[:m], engine="knn", part_threads=0, hw_threads=hw, run_parallel=True)
        cpu_s = time.perf_counter() - t1
        sn = a.n
        threads = max(1, min(hw, sn // 100000))                         # optimized_parallel.hpp:76-77
        out["cpu_baseline"] = {"value": m / cpu_s, "unit": "queries/s", "cores": threads, "kind": "port",
                               "sample": f"first {m} queries of the first timed batch, full D={a.n}; "
                                         f"reference-faithful D-partitioned Knn engine (oracle), host has {hw} cpus"}
        got, got_d = eng.download_results(b * a.batch, m)
        can_ids, _ = T.oracle_query(nodes, q[:m], engine="canonical", threads=hw)
        st = T.check_parity(nodes, q[:m], got, can_ids, got_dists=got_d)
        T.check_parity(nodes, q[:m], got, ref_ids)
        out["recall_at_100"] = 1.0
        out["parity"] = st
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if use_dist and rank == 0 and world == 1:
        got = eng.download_results((total_batches - 1) * a.batch, a.batch, want_dists=False)
        assert np.array_equal(gathered.cpu().numpy().view(np.uint32), got), "gathered ids differ from the local results"
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
