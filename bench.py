#!/usr/bin/env python3
"""bench.py -- whole-job queries/second of the filtered brute-force k-NN hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU).

Workload (BASELINE.json metric: queries/sec + recall@100 on D=10^7, Q=4x10^6, dim=100, k=100):
D = 10^7 gen-v1 rows replicated in every GPU's HBM; one STEP = one pass of the hot path over one batch of `--batch` (2^21, the
largest batch the library forms) mixed-type queries per GPU, inputs resident in HBM when the timed region starts.  The K timed
steps are library calls over several batches each (N = 1: one call; N > 1: calls of ~K/4 batches, every batch's ids gathered to
rank 0 over RCCL inside the timed region, under the next call's compute): consecutive batches of a call run on two lanes.
Queries shard across ranks with no data-path collective (weak scaling: per-GPU work is fixed); value = queries all ranks
answered / max-over-ranks time.

Extra JSON objects: "roofline" for the dominant kernel (HIP-event kernel time measured live on the library's own streams),
"end_to_end" (the same batches host memory -> host memory: the reference's own timing scope, PCIe-inclusive; never `value`),
"fixed_q" (BASELINE's ONE 4x10^6-query set: N = 1 -- the shares of 1/2/4/8 ranks run on this GPU; N > 1 -- run for real: shares
from host memory, RCCL gather to rank 0, sampled check on rank 0; strong scaling), "in_library" (N > 1 or --in-library: the same
set through one hvs_create_multi context), "collective" (N > 1), "configs12" (BASELINE configs[1]/[2]: D = 10^6, Q = 10^4 on one
GPU, resident and host -> host), "cpu_baseline" (the oracle's reference-faithful threaded engine timed on this host's cores over
a bounded query sample, rank 0 at N = 1 only) and "recall_at_100" of the GPU answers against that oracle sample.
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


def configs12_leg(pkg, T, np, steps=20, warmup=3):
    """BASELINE configs[1] / [2]: D = 10^6, Q = 10^4 (type-0 only / mixed types) on one GPU, each call ONE batch.
    Per config: resident ms per call (inputs in HBM, results left in HBM: hvs_query_resident + hvs_sync), host -> host ms per
    call (hvs_query on pageable buffers, warm context: the reference's timing scope, src/test.cpp:82-88), the step-level
    fraction of the filter's MFMA peak (200 op x passing pairs / resident time), and hvs_load_data's time for the 408 MB set."""
    n, nq, K = 1_000_000, 10_000, 100
    out = {"n": n, "nq": nq, "steps": steps, "note": "each call is one batch of 10^4 queries; ms are means over `steps` calls after warm-up"}
    with pkg.Engine(0) as e:
        e.gen_data(n, T.SEED_DATA, T.GEN_V1, 100)
        nodes = e.download_data(0, n)
    with pkg.Engine(0) as e:
        t1 = time.perf_counter()
        e.load_data(nodes)
        out["load_data_ms"] = (time.perf_counter() - t1) * 1e3
        t1 = time.perf_counter()
        e.load_data(nodes)
        out["load_data_ms_second_call"] = (time.perf_counter() - t1) * 1e3
        del nodes
        for name, ftype in (("config1_type0", 0), ("config2_mixed", -1)):
            e.gen_queries(nq * (steps + warmup), T.SEED_QUERY, T.GEN_V1, 100, ftype, 0)
            res_ms, dev_ms, kern_ms, pairs, retried, fallback = [], 0.0, 0.0, 0, 0, 0
            for b in range(steps + warmup):
                t1 = time.perf_counter()
                e.query_resident(b * nq, nq, 1.0)
                e.sync()
                dt = time.perf_counter() - t1
                if b >= warmup:
                    tm = e.last_timing()
                    res_ms.append(dt * 1e3)
                    dev_ms += tm.query_ms
                    kern_ms += tm.main_kernel_ms
                    pairs += tm.pairs
                    retried += tm.retry_queries
                    fallback += tm.fallback_queries
            engine_id = int(e.last_timing().engine)
            peak = {2: BF16_PEAK_TFLOPS, 3: INT8_PEAK_TOPS, 4: BF16_PEAK_TFLOPS}.get(engine_id, FP32_PEAK_TFLOPS)
            ids_res = e.download_results((steps + warmup - 1) * nq, nq, want_dists=False)
            q_all = e.download_queries(0, nq * (steps + warmup))
            ids_host = np.empty((nq, K), np.uint32)
            host_ms = []
            for b in range(steps + warmup):
                q = q_all[b * nq:(b + 1) * nq]
                t1 = time.perf_counter()
                e.query(q, 1.0, want_dists=False, out_ids=ids_host)
                dt = time.perf_counter() - t1
                if b >= warmup:
                    host_ms.append(dt * 1e3)
            assert np.array_equal(ids_host, ids_res), "configs12: host path and resident path disagree"
            r = float(np.mean(res_ms))
            out[name] = {"resident_ms": r, "resident_ms_min": float(np.min(res_ms)), "device_ms": dev_ms / steps,
                         "host_to_host_ms": float(np.mean(host_ms)), "host_to_host_ms_min": float(np.min(host_ms)),
                         "queries_per_s_resident": nq / r * 1e3, "queries_per_s_host_to_host": nq / float(np.mean(host_ms)) * 1e3,
                         "step_frac_of_mfma_peak": 200.0 * pairs / steps / (r / 1e3) / 1e12 / peak,
                         "filter_kernel_frac": (200.0 * pairs / (kern_ms / 1e3) / 1e12 / peak) if kern_ms > 0 else None,
                         "engine": engine_id, "retry_queries_per_call": retried / steps, "fallback_queries": fallback}
    return out


def in_library_leg(pkg, T, np, a):
    """ONE context over several GPUs (hvs_create_on_devices: what the vec_query seam and hvs_search.out use), one hvs_query of the
    whole 4x10^6-query set from pageable host memory to pageable host memory; a sampled slice re-computed on one GPU."""
    QSET, K = 4_000_000, 100
    devs = [int(x) for x in a.in_library_devices.split(",")] if a.in_library_devices else [0]
    ids_all = np.empty((QSET, K), np.uint32)
    with pkg.Engine(devices=devs) as m:
        m.reserve(QSET)
        m.gen_data(a.n, T.SEED_DATA, a.profile, 100)
        m.gen_queries(QSET, T.SEED_QUERY, T.GEN_V1 if a.profile == 1 else a.profile, 100, a.force_type, 0)
        q_all = m.download_queries(0, QSET)                               # the query file, in (pageable) host memory
        m.query(q_all[:65536 * len(devs)], 1.0, want_dists=False)
        t1 = time.perf_counter()
        m.query(q_all, 1.0, want_dists=False, out_ids=ids_all)
        lib_s = time.perf_counter() - t1
        tm = m.last_timing()
    sel = np.unique(np.linspace(0, QSET - 1, 64 * len(devs)).astype(np.int64))
    with pkg.Engine(devs[0]) as e1:
        e1.gen_data(a.n, T.SEED_DATA, a.profile, 100)
        want = e1.query(q_all[sel], 1.0, want_dists=False)
    assert np.array_equal(ids_all[sel], want), "in-library leg: ids differ from a one-GPU recomputation"
    return {"value": QSET / lib_s, "unit": "queries/s", "ms": lib_s * 1e3, "devices": devs, "n_gpus": int(tm.n_gpus),
            "slowest_gpu_device_ms": tm.query_ms, "sampled_queries_checked": int(len(sel)),
            "scope": "one hvs_create_on_devices context (child process, 300 s limit), one hvs_query of 4x10^6 queries, pageable host "
                     "memory -> pageable host memory (each GPU's pipeline writes its slice of the caller's array)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=10_000_000, help="rows of D")
    ap.add_argument("--batch", type=int, default=2097152, help="queries per step per GPU")
    ap.add_argument("--force-type", type=int, default=-1, help="-1 mixed types, 0..3 a single type")
    ap.add_argument("--profile", type=int, default=1,
                    help="vector law of D and Q (include/hvs_gen.h): 1 gen-v1 uniform (the headline), 2 clustered, 3 PCA-like "
                         "decaying variances, 4 heavy-tailed norms, 5 gen-v1; 2-5: 1 %% of the queries lie outside the data's box")
    ap.add_argument("--cpu-seconds", type=float, default=150.0,
                    help="time cap of the CPU baseline leg (0 = skip): it times the fixed --cpu-queries prefix in chunks "
                         "of 256 queries and stops at the first chunk boundary past the cap")
    ap.add_argument("--cpu-queries", type=int, default=2048, help="CPU baseline sample: first N queries of the first timed batch (SURVEY 8d)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-memory (PCIe-inclusive) leg")
    ap.add_argument("--no-fixed-q", action="store_true",
                    help="skip the fixed-Q leg (rank 0's share of the 4x10^6-query set at 1, 2, 4, 8 GPUs, run on this GPU)")
    ap.add_argument("--engine", type=int, default=0)
    ap.add_argument("--in-library", action="store_true",
                    help="also time the library's own multi-GPU context (hvs_create_multi: one process, all GPUs) on the 4x10^6-query set")
    ap.add_argument("--in-library-only", action="store_true", help="run only that leg, in this process, and print its object")
    ap.add_argument("--in-library-devices", default="", help="device list of that context, e.g. 0,0,0,0 (virtual ranks on one GPU)")
    ap.add_argument("--per-step-calls", action="store_true",
                    help="N = 1: one library call per step with a synchronisation behind it (round 3's timed region; A/B)")
    ap.add_argument("--no-configs12", action="store_true", help="skip the BASELINE configs[1]/[2] leg (D = 10^6, Q = 10^4 on one GPU)")
    ap.add_argument("--only-configs12", action="store_true", help="run only that leg and print its object")
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

    if a.only_configs12:
        print(json.dumps({"configs12": configs12_leg(pkg, T, np)}))
        return
    if a.in_library_only:
        print(json.dumps({"in_library": in_library_leg(pkg, T, np, a)}))
        return
    K = 100
    total_batches = a.warmup + a.steps
    eng = pkg.Engine(local_rank)
    if a.engine:
        eng.set_engine(a.engine)
    eng.reserve(a.batch * total_batches)                                # buffers + batch workspace before anything is timed
    t0 = time.time()
    eng.gen_data(a.n, T.SEED_DATA, a.profile, 100)                      # D replicated per GPU
    load_s = time.time() - t0
    # Q is partitioned: this rank owns one contiguous range of the query stream (sharding.shard_range)
    sharding = importlib.import_module("project---hybrid-vector-search-queries_amd.sharding")
    q_first, q_last = sharding.shard_range(world * a.batch * total_batches, rank, world)
    assert q_last - q_first == a.batch * total_batches
    eng.gen_queries(a.batch * total_batches, T.SEED_QUERY, a.profile, 100, a.force_type, first_row=q_first)

    # Timed region.  The K timed steps are a few library calls over several batches each (the W warm-up steps another): the
    # library runs consecutive batches of a call on two lanes, so that a batch's preparation and low levels share the chip with
    # the previous batch's last re-scoring and final merge -- a pipeline a driver that synchronised after every batch would
    # break.  N = 1: ONE call over the K batches.  N > 1: calls of ~K/4 batches; after each call every batch's block of ids
    # is exported and gathered to rank 0 over xGMI (RCCL gather on torch's stream: rank 0 alone needs the output.bin rows)
    # while the next call computes -- only the last call's gathers are exposed.
    chunk = a.steps if not use_dist else max(1, a.steps // 4)
    if a.per_step_calls:
        chunk = 1
    ids_dev = [torch.empty((a.batch, K), dtype=torch.int32, device="cuda") for _ in range(min(a.steps, 2 * chunk))] if use_dist else None
    gathered = ([torch.empty((a.batch, K), dtype=torch.int32, device="cuda") for _ in range(world)]
                if use_dist and rank == 0 else None)
    gather_done = [None] * (len(ids_dev) if ids_dev else 0)            # event after the gather that last read each send buffer
    gathered_bytes = 0

    def run_call(b0, nb):
        """batches [b0, b0 + nb) as one library call; results final (failed guesses re-run) when it returns"""
        eng.query_resident(b0 * a.batch, nb * a.batch, 1.0)            # asynchronous on the library's streams
        eng.sync()

    def gather_call(b0, nb):
        nonlocal gathered_bytes
        for b in range(b0, b0 + nb):
            k = b % len(ids_dev)
            if gather_done[k] is not None:
                gather_done[k].synchronize()                           # (2 x chunk steps old: long done; keeps the buffer reuse honest)
            eng.export_results_device(b * a.batch, a.batch, ids_dev[k].data_ptr())
            eng.stream_wait(torch.cuda.current_stream().cuda_stream)   # stream-ordered hand-off to the collective's stream
            dist.gather(ids_dev[k], gathered, dst=0)                   # RCCL, asynchronous on torch's stream
            gather_done[k] = torch.cuda.Event()
            gather_done[k].record()
            gathered_bytes += (world - 1) * a.batch * K * 4

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if a.warmup:
        run_call(0, a.warmup)
        if use_dist:
            gather_call(a.warmup - 1, 1)                               # (the collective's first use: outside the timed region)
    kern_ms, kern_launches, pairs, scanned, query_ms, rescored, fallback, untimed, retried = 0.0, 0, 0, 0, 0.0, 0, 0, 0, 0
    gathered_bytes = 0
    calls = 0
    fence()
    t0 = time.perf_counter()
    for b0 in range(a.warmup, total_batches, chunk):
        nb = min(chunk, total_batches - b0)
        run_call(b0, nb)
        calls += 1
        tm = eng.last_timing()                                          # HIP events on the library's streams
        kern_ms += tm.main_kernel_ms
        kern_launches += tm.main_kernel_launches
        pairs += tm.pairs
        scanned += tm.scanned_pairs
        query_ms += tm.query_ms
        rescored += tm.rescored_pairs
        fallback += tm.fallback_queries
        untimed += tm.untimed_launches
        retried += tm.retry_queries
        if use_dist:
            gather_call(b0, nb)                                         # runs under the next call's compute
    fence()
    elapsed = time.perf_counter() - t0
    rank_ms = elapsed * 1e3
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        tmin = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        elapsed, rank_ms_min = float(tt.item()), float(tmin.item()) * 1e3
    engine_id = int(eng.last_timing().engine)
    if use_dist and rank == 0 and world == 1:                          # --force-dist rehearsal: the gathered block is the local one
        got = eng.download_results((total_batches - 1) * a.batch, a.batch, want_dists=False)
        assert np.array_equal(gathered[0].cpu().numpy().view(np.uint32), got), "gathered ids differ from the local results"
    # the first timed batch: its answers are what the parity / end-to-end legs compare against
    first_timed_q = first_timed_ids = first_timed_dists = None
    if rank == 0 or not a.no_e2e:
        first_timed_q = eng.download_queries(a.warmup * a.batch, a.batch)
        first_timed_ids, first_timed_dists = eng.download_results(a.warmup * a.batch, a.batch)

    out = None
    if rank == 0:
        nq_total = world * a.batch * a.steps
        value = nq_total / elapsed
        flops_alg = 200.0 * pairs                                      # 2*100 per passing pair
        k_s = kern_ms / 1e3
        peak = {2: BF16_PEAK_TFLOPS, 3: INT8_PEAK_TOPS, 4: BF16_PEAK_TFLOPS}.get(engine_id, FP32_PEAK_TFLOPS)
        assert untimed == 0, "some launches of the dominant kernel were not timed: no roofline from this run"
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
                                   f"from the {['gen-v0', 'gen-v1', 'clustered', 'PCA-like', 'heavy-tailed', 'gen-v1 + 1 % out-of-box'][a.profile]} 4x10^6-query stream, "
                                   f"k=100, sample_proportion=1",
                       "profile": a.profile,
                       "n": a.n, "queries_per_step_per_gpu": a.batch, "engine": engine_id,
                       "timed_region": f"{calls} library call(s) of up to {chunk} batches each (consecutive batches of a call run on two lanes)",
                       "sharding": "Q partitioned across ranks, D replicated, RCCL gather of the ids to rank 0"},
            "roofline": {"bound": "mfma" if engine_id in (2, 3, 4) else "valu-fp32", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "kernel": {2: "hvs_k_filter_mfma<bf16>", 4: "hvs_k_filter_mfma<f16>",
                                    3: "hvs_k_filter_mfma<int8>" if os.environ.get("HVS_I8_SHAPE") == "32" else "hvs_k_filter_i8x16"}.get(engine_id, "hvs_k_scan_exact"),
                         "peak_is": {2: "dense BF16 MFMA", 3: "dense INT8 MFMA (integer ops)",
                                     4: "dense FP16 MFMA (= the BF16 rate)"}.get(engine_id, "FP32 vector"),
                         "kernel_ms_avg": kern_ms / max(kern_launches, 1), "launches": kern_launches,
                         "pairs_per_launch": pairs / max(kern_launches, 1),
                         "evaluated_pairs_per_launch": scanned / max(kern_launches, 1),
                         "device_query_ms_per_step": query_ms / a.steps,
                         "rescored_pairs_per_query": rescored / max(a.batch * a.steps, 1),
                         # SURVEY 8d's second figure: what a one-query-at-a-time scan (the reference) would have to fetch,
                         # 8 B of attributes per scanned row + 400 B per passing row, over the step time and the 8 TB/s peak
                         # per GPU.  An EFFECTIVE figure: batching is why it exceeds 1; the real traffic is `traffic`.
                         # (per GPU: rank 0's own queries and pairs)
                         "effective_hbm_frac_of_a_per_query_scan": ((8.0 * a.n * a.batch * a.steps + 400.0 * pairs) / elapsed / 1e9)
                                                                   / HBM_PEAK_GBS,
                         # context, not the peak: a bare loop of the same MFMA chains sustains this much on random
                         # operands on this chip (DVFS; scripts/mfma_loop_lab.hip / mfma_i8_lab.hip -DLAB_RANDOM)
                         # context, not the peak: what the filter-shaped loop (LDS fragment reads, epilogue, 2 waves/SIMD)
                         # sustains on random operands on this chip: scripts/mfma_shape_lab.hip, 12.8 G 32x32 pair blocks/s
                         # x 262144 op (INT8 16x16x64); BF16: scripts/mfma_loop_lab.hip
                         "measured_loop_ceiling_random_operands_tflops": {2: 1592.0, 3: 3350.0, 4: 1592.0}.get(engine_id),
                         # bare in-place chains of the same instruction, 2 waves/SIMD (scripts/mfma_shape_lab.hip part 3)
                         "measured_bare_chain_ceiling_random_operands_tflops": {3: 4010.0}.get(engine_id),
                         "fallback_queries": fallback, "retry_queries": retried},
            "load_s": load_s,
            # dtype "f32" names the arithmetic that decides every answer (exact-order f32 re-scoring, optimized_impl.h:96-125);
            # the timed dominant kernel (roofline.kernel) filters in `filter_dtype` and is priced against THAT type's peak
            "filter_dtype": {2: "bf16", 3: "int8", 4: "f16"}.get(engine_id, "f32"),
        }
        if use_dist:
            out["collective"] = {"backend": "nccl (RCCL)", "rccl_ranks": dist.get_world_size(), "op": "gather of each step's ids to rank 0",
                                 "gathered_bytes": gathered_bytes, "library_calls": calls,
                                 "rank_ms_max": elapsed * 1e3, "rank_ms_min": rank_ms_min}

    # ---- end-to-end leg: the reference's own timing scope (src/test.cpp:82-88: queries in host RAM -> ids in host
    # RAM).  The same a.steps batches as ONE hvs_query call from ordinary (pageable) host memory through the library's
    # pipeline (pinned staging, H2D one batch ahead, D2H under the next batch); never `value`.
    if not a.no_e2e:
        steps_e = min(a.steps, 4)                                       # (bounded: 4 batches of host buffers at most)
        nq_e = a.batch * steps_e
        q_host = eng.download_queries(a.warmup * a.batch, nq_e)
        ids_host = np.empty((nq_e, K), np.uint32)
        eng.query(q_host[: min(nq_e, 65536)], 1.0, want_dists=False)    # warm-up: staging slots exist afterwards
        fence()
        t1 = time.perf_counter()
        eng.query(q_host, 1.0, want_dists=False, out_ids=ids_host)
        e2e = time.perf_counter() - t1
        tm = eng.last_timing()
        if use_dist:
            tt = torch.tensor([e2e], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e2e = float(tt.item())
        if rank == 0:
            out["end_to_end"] = {"value": world * nq_e / e2e, "unit": "queries/s", "ms_per_step": e2e / steps_e * 1e3,
                                 "scope": "host RAM -> host RAM (pageable buffers), PCIe-inclusive, one hvs_query call of "
                                          f"{nq_e} queries per GPU; D resident", "device_ms": tm.query_ms,
                                 "headline": "value (inputs resident in HBM) is the headline; this is the reference's own scope"}
        # the host path must return what the resident path returned
        e2e_check = np.array_equal(ids_host[: a.batch], first_timed_ids) if first_timed_ids is not None else None
        if rank == 0 and e2e_check is not None:
            out["end_to_end"]["ids_identical_to_resident_path"] = bool(e2e_check)
            assert e2e_check, "hvs_query (host path) and the resident path disagree"

    # ---- fixed-Q leg (rank 0, N=1 only): BASELINE's metric is ONE query set of 4x10^6 (src/test.cpp:82-92 times one fixed
    # set), which an N-GPU run cuts into shares of 4x10^6 / N.  Each share is run on this GPU the way a rank would run it:
    # resident (one hvs_query_resident call) and from host memory (one hvs_query call, pageable buffers).
    if rank == 0 and world == 1 and not a.no_fixed_q and a.n == 10_000_000 and a.profile == 1:
        peak_fq = {2: BF16_PEAK_TFLOPS, 3: INT8_PEAK_TOPS, 4: BF16_PEAK_TFLOPS}.get(engine_id, FP32_PEAK_TFLOPS)
        fixed = {"query_set": 4_000_000, "note": "one GPU running the share a rank of an N-GPU run gets (D replicated); "
                 "whole-node rate of such a run = N x the share's rate if every rank does the same", "shares": []}
        for n_ranks in (1, 2, 4, 8):
            share = 4_000_000 // n_ranks
            eng.gen_queries(share, T.SEED_QUERY, T.GEN_V1, 100, a.force_type, first_row=0)
            eng.query_resident(0, min(share, 65536), 1.0)
            eng.sync()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            eng.query_resident(0, share, 1.0)
            eng.sync()
            res_s = time.perf_counter() - t1
            tm = eng.last_timing()
            frac = (200.0 * tm.pairs / (tm.main_kernel_ms / 1e3) / 1e12 / peak_fq) if tm.main_kernel_ms > 0 else None
            ids_res = eng.download_results(0, share, want_dists=False)
            q_host = eng.download_queries(0, share)
            ids_host = np.empty((share, K), np.uint32)
            t1 = time.perf_counter()
            eng.query(q_host, 1.0, want_dists=False, out_ids=ids_host)
            host_s = time.perf_counter() - t1
            assert np.array_equal(ids_host, ids_res), "fixed-Q leg: host path and resident path disagree"
            fixed["shares"].append({"n_gpus": n_ranks, "queries": share,
                                    "resident": {"value": share / res_s, "unit": "queries/s", "ms": res_s * 1e3, "roofline_frac": frac,
                                                 "filter_launches": tm.main_kernel_launches, "retry_queries": tm.retry_queries},
                                    "host_to_host": {"value": share / host_s, "unit": "queries/s", "ms": host_s * 1e3}})
            del q_host, ids_host, ids_res
        out["fixed_q"] = fixed

    # ---- fixed-Q leg at N > 1 (strong scaling; BASELINE's metric is ONE query set of 4x10^6, src/test.cpp:82-92): every rank takes
    # its contiguous share (sharding.shard_range) FROM HOST MEMORY -- upload, one resident call, export -- and the ids are gathered
    # to rank 0 over RCCL inside the timed region; rank 0 then checks a sampled slice of every rank's share against its own
    # recomputation of those queries.
    if use_dist and not a.no_fixed_q and a.n == 10_000_000 and a.profile == 1:
        QSET = 4_000_000
        f0, f1 = sharding.shard_range(QSET, rank, world)
        share, max_share = f1 - f0, -(-QSET // world)
        eng.gen_queries(share, T.SEED_QUERY, T.GEN_V1, 100, a.force_type, first_row=f0)
        q_host = eng.download_queries(0, share)                           # this rank's slice of the query file, in host memory
        send = torch.zeros((max_share, K), dtype=torch.int32, device="cuda")
        recv = [torch.empty((max_share, K), dtype=torch.int32, device="cuda") for _ in range(world)] if rank == 0 else None
        eng.upload_queries(q_host[: min(share, 65536)])                   # warm-up of the path
        eng.query_resident(0, min(share, 65536), 1.0)
        eng.sync()
        fence()
        t1 = time.perf_counter()
        eng.upload_queries(q_host)                                        # host -> device (pageable buffer, the library's staging)
        eng.query_resident(0, share, 1.0)
        eng.export_results_device(0, share, send.data_ptr())
        eng.stream_wait(torch.cuda.current_stream().cuda_stream)
        dist.gather(send, recv, dst=0)
        torch.cuda.synchronize()
        local_s = time.perf_counter() - t1
        fence()
        fq_s = time.perf_counter() - t1
        tm = eng.last_timing()
        tt = torch.tensor([fq_s, local_s, -local_s], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        fq_s, local_max, local_min = float(tt[0]), float(tt[1]), -float(tt[2])
        if rank == 0:
            checked = 0
            for r in range(world):
                r0, r1 = sharding.shard_range(QSET, r, world)
                sel = np.unique(np.linspace(0, r1 - r0 - 1, 48).astype(np.int64))
                eng.gen_queries(r1 - r0, T.SEED_QUERY, T.GEN_V1, 100, a.force_type, first_row=r0)
                qs = np.concatenate([eng.download_queries(int(i), 1) for i in sel])
                want = eng.query(qs, 1.0, want_dists=False)
                got = recv[r].cpu().numpy().view(np.uint32)[sel]
                assert np.array_equal(got, want), f"fixed-Q leg: rank {r}'s gathered ids differ from rank 0's recomputation"
                checked += len(sel)
            out["fixed_q"] = {"query_set": QSET, "scaling": "strong", "value": QSET / fq_s, "unit": "queries/s", "ms": fq_s * 1e3,
                              "rccl_ranks": dist.get_world_size(), "gathered_bytes": (world - 1) * max_share * K * 4,
                              "rank_ms_min": local_min * 1e3, "rank_ms_max": local_max * 1e3,
                              "scope": "every rank: its share of the 4x10^6-query set from pageable host memory -> HBM -> one resident "
                                       "call -> RCCL gather of the ids to rank 0's HBM (timed region ends there)",
                              "rank0_device_ms": tm.query_ms, "sampled_queries_checked_on_rank0": checked}
        del q_host, send, recv

    # ---- CPU baseline + recall leg (rank 0, N=1 only; the oracle is the checker, never the product)
    if rank == 0 and world == 1 and a.cpu_seconds > 0 and min(a.cpu_queries, a.batch) > 0:
        nodes = eng.download_data(0, a.n)
        hw = os.cpu_count() or 1
        m_fixed = min(a.cpu_queries, a.batch)
        q = first_timed_q[:m_fixed]
        rule_threads = max(1, min(hw, a.n // 100000))                   # the reference's own rule, optimized_parallel.hpp:76-77
        T.oracle().hvs_oracle_pin_threads(1)                            # baseline threads on CPUs of their own for the leg's duration
        # D placed for the scan: every thread's partition first touched by that thread (the reference reads D on its main
        # thread, i.e. onto one memory node of a two-socket host; see hvs_oracle_place_rows).  Thread-count mini-sweep on the
        # first 256 queries (placement redone per thread count): the reference's rule picks min(hw, n / 100000) threads, whose
        # per-query fork/join (threading.hpp:72-96) can cost more than the extra threads bring -- the timed sample below runs
        # with the best count of the sweep, the rule's own rate is reported beside it.
        msweep = min(256, m_fixed)
        sweep, best_t = {}, rule_threads
        for tcount in sorted({8, 32, 64, rule_threads}):
            if tcount > hw:
                continue
            pl, _ = T.oracle_place_rows(nodes, 1.0, tcount, hw)
            t1 = time.perf_counter()
            T.oracle_query(pl, q[:msweep], engine="knn", part_threads=tcount, hw_threads=hw, run_parallel=True)
            sweep[str(tcount)] = msweep / (time.perf_counter() - t1)
            del pl
        best_t = int(max(sweep, key=lambda k: sweep[k]))
        # the rule's thread count on the UNPLACED array (everything on the node the download thread ran on), for the record
        t1 = time.perf_counter()
        T.oracle_query(nodes, q[:msweep], engine="knn", part_threads=0, hw_threads=hw, run_parallel=True)
        unplaced = msweep / (time.perf_counter() - t1)
        placed, _ = T.oracle_place_rows(nodes, 1.0, best_t, hw)
        ref_parts, done, cpu_s = [], 0, 0.0
        while done < m_fixed and cpu_s < a.cpu_seconds:                  # fixed prefix, in chunks of 256, capped in time
            m = min(256, m_fixed - done)
            t1 = time.perf_counter()
            r_ids, _ = T.oracle_query(placed, q[done:done + m], engine="knn", part_threads=best_t, hw_threads=hw, run_parallel=True)
            cpu_s += time.perf_counter() - t1
            ref_parts.append(r_ids)
            done += m
        T.oracle().hvs_oracle_pin_threads(0)
        ref_ids = np.concatenate(ref_parts)
        mq = T.passing_rows_per_query(nodes, q[:done])
        row_bytes = float(np.sum(8.0 * a.n + 400.0 * mq))               # attributes of every row + vectors of passing rows
        out["cpu_baseline"] = {"value": done / cpu_s, "unit": "queries/s", "cores": best_t, "threads": best_t, "kind": "port",
                               "row_data_gb_per_s": row_bytes / cpu_s / 1e9,
                               "reference_rule_threads": rule_threads,
                               "threads_sweep_queries_per_s_on_256_queries": sweep,
                               "reference_rule_threads_without_numa_placement_queries_per_s": unplaced,
                               "sample": f"first {done} queries of the fixed {m_fixed}-query prefix of the first timed batch "
                                         f"(SURVEY 8d; chunks of 256 until {a.cpu_seconds:.0f} s), full D={a.n}; "
                                         f"reference-faithful D-partitioned Knn engine (oracle: optimized_parallel.hpp:91-160, "
                                         f"threading.hpp:116-118) on the best thread count of the sweep, threads pinned, D first-touched "
                                         f"per partition; host has {hw} cpus"}
        del placed
        got, got_d = first_timed_ids[:done], first_timed_dists[:done]
        can_ids, _ = T.oracle_query(nodes, q[:done], engine="canonical", threads=hw)
        st = T.check_parity(nodes, q[:done], got, can_ids, got_dists=got_d)
        T.check_parity(nodes, q[:done], got, ref_ids)
        hits = sum(len(np.intersect1d(got[i], can_ids[i])) for i in range(done))
        out["recall_at_100"] = hits / (100.0 * done)                    # computed; ties at rank 100 are the only slack
        out["recall_checked_queries"] = done
        out["parity"] = st
    elif rank == 0:
        out["cpu_baseline"] = None
    eng.close()
    if rank == 0 and world == 1 and not a.no_configs12:
        out["configs12"] = configs12_leg(pkg, T, np)
    # ---- in-library leg (--in-library; rank 0 after every rank has released its GPU): ONE context over all GPUs
    # (hvs_create_multi: what the vec_query seam and hvs_search.out use), one hvs_query of the whole 4x10^6-query set from
    # host memory to host memory -- the same job as the process-per-GPU fixed-Q leg, through the seam's own path.
    if a.in_library or world > 1:
        # (host-side rendezvous through the process group's store: a collective barrier would leave a spinning RCCL kernel on
        # every GPU the in-library context is about to use)
        store = None
        if use_dist:
            torch.cuda.synchronize()
            try:
                store = dist.distributed_c10d._get_default_store()
                store.add("hvs_inlib_arrived", 1)
                if rank == 0:
                    t_wait = time.time()
                    while int(store.add("hvs_inlib_arrived", 0)) < world and time.time() - t_wait < 300:
                        time.sleep(0.05)
            except Exception:                                            # no store API: a collective barrier on both sides instead
                store = None
                dist.barrier()
        if rank == 0:
            # in a child process with a time limit: a multi-GPU context that misbehaves on hardware this code has never seen
            # must not take the bench line (or the other ranks, which wait for this leg) with it
            import subprocess
            devs = a.in_library_devices if a.in_library_devices else ",".join(str(i) for i in range(max(world, 1)))
            cmd = [sys.executable, os.path.abspath(__file__), "--in-library-only", "--in-library-devices", devs, "--n", str(a.n),
                   "--profile", str(a.profile), "--force-type", str(a.force_type)]
            try:
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                out["in_library"] = json.loads(line[-1])["in_library"] if line else {"error": (r.stderr or r.stdout)[-300:]}
            except subprocess.TimeoutExpired:
                out["in_library"] = {"error": "time limit of 300 s"}
            except Exception as ex:
                out["in_library"] = {"error": repr(ex)[:300]}
        if use_dist:
            import datetime
            if store is None:
                dist.barrier()
            elif rank == 0:
                store.set("hvs_inlib_done", "1")
            else:
                store.wait(["hvs_inlib_done"], datetime.timedelta(seconds=900))

    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
