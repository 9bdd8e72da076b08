"""Multi-GPU plan of the path: queries are independent (reference
include/optimized_parallel.hpp:91 carries no state between iterations), so Q is cut into one
contiguous range per rank, D is replicated in every GPU's HBM and the only exchange is the gather
of the result ids (RCCL all_gather over xGMI when the tensors live on GPUs; any torch.distributed
backend works, the CPU tests use gloo)."""
from __future__ import annotations

import numpy as np


def shard_range(nq: int, rank: int, world: int):
    """Contiguous, balanced [q0, q1) of rank `rank`; the first nq % world ranks get one more query."""
    base, rem = divmod(nq, world)
    q0 = rank * base + min(rank, rem)
    return q0, q0 + base + (1 if rank < rem else 0)


def gather_ids(local_ids, nq: int, group=None):
    """All ranks contribute their [q0,q1) x 100 block; every rank returns the full nq x 100 array in
    query order (the layout of output.bin, reference include/io.h:23-36).  `local_ids` is a torch
    tensor (int32, on the device of the backend)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per = -(-nq // world)  # ranks pad to the largest shard so that one fixed-size all_gather suffices
    pad = torch.zeros((per, local_ids.shape[1]), dtype=local_ids.dtype, device=local_ids.device)
    pad[: local_ids.shape[0]] = local_ids
    out = torch.empty((world * per, local_ids.shape[1]), dtype=local_ids.dtype, device=local_ids.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = []
    for r in range(world):
        q0, q1 = shard_range(nq, r, world)
        parts.append(out[r * per: r * per + (q1 - q0)])
    return torch.cat(parts, 0)


def gather_ids_to_root(local_ids, nq: int, dst: int = 0, group=None):
    """The result gather of the multi-GPU plan: every rank's [q0,q1) x k block of ids travels to rank
    `dst` only (RCCL gather over xGMI on GPUs: rank 0 alone writes output.bin, so the other ranks need nothing --
    an all_gather would move world x the bytes).  Returns the nq x k array on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = -(-nq // world)  # ranks pad to the largest shard so that one fixed-size gather suffices
    pad = torch.zeros((per, local_ids.shape[1]), dtype=local_ids.dtype, device=local_ids.device)
    pad[: local_ids.shape[0]] = local_ids
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        q0, q1 = shard_range(nq, r, world)
        parts.append(bufs[r][: q1 - q0])
    return torch.cat(parts, 0)


def run_sharded(answer, queries: np.ndarray, group=None, device="cpu"):
    """answer(q_rows) -> (n_local x 100) uint32 ids for this rank's queries.  Returns all ids."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    q0, q1 = shard_range(queries.shape[0], rank, world)
    ids = np.ascontiguousarray(answer(queries[q0:q1]), np.uint32)
    t = torch.from_numpy(ids.view(np.int32)).to(device)
    return gather_ids(t, queries.shape[0], group).cpu().numpy().view(np.uint32)


# ----------------------------------------------------------------------------------------------
# D-sharded mode: for data sets larger than one GPU's HBM the ROWS are partitioned
# instead of the queries.  Every rank answers all queries against its contiguous row range with
# padding switched off (hvs_set_padding(ctx, 0)); the partial top-100 lists are exchanged and merged
# -- the multi-GPU counterpart of Knn::merge (reference include/optimized_impl.h:337-385) -- and the
# padding of include/optimized_parallel.hpp:149-157 is applied once, from the tail of the whole set.
# ----------------------------------------------------------------------------------------------

EMPTY_ID = np.uint32(0xFFFFFFFF)
_EMPTY_KEY = np.uint64(0xFFFFFFFFFFFFFFFF)


def row_shard_range(n: int, rank: int, world: int):
    """Contiguous row range of a rank; every shard must keep >= 100 rows (hvs_load_data's precondition)."""
    return shard_range(n, rank, world)


def _keys(ids, dists, row0):
    ids = np.asarray(ids, np.uint32)
    d = np.ascontiguousarray(dists, np.float32).view(np.uint32).astype(np.uint64)
    k = (d << np.uint64(32)) | (ids.astype(np.uint64) + np.uint64(row0))
    return np.where(ids == EMPTY_ID, _EMPTY_KEY, k)


def merge_data_shards(parts, n_total: int, pad_dists):
    """parts: list of (ids_local [nq,100] u32, dists [nq,100] f32, row0) -- one per shard, padding off.
    pad_dists: [nq,100] f32, pad_dists[q, s] = exact-order distance of query q to row n_total-1-s.
    Returns (ids [nq,100] u32 global, dists [nq,100] f32) in the canonical order (dist asc, id asc)."""
    k = np.asarray(parts[0][0]).shape[1]            # 100 unless the contexts were given another k (hvs_set_k)
    allk = np.concatenate([_keys(i, d, r0) for i, d, r0 in parts], axis=1)
    allk.sort(axis=1)
    best = allk[:, :k].copy()
    m = (best != _EMPTY_KEY).sum(axis=1)
    pad_ids = (np.uint64(n_total - 1) - np.arange(k, dtype=np.uint64))[None, :]
    padk = (np.ascontiguousarray(pad_dists, np.float32).view(np.uint32).astype(np.uint64) << np.uint64(32)) | pad_ids
    for q in np.nonzero(m < k)[0]:
        need = k - int(m[q])
        row = np.concatenate([best[q, : m[q]], padk[q, :need]])  # rows n-1, n-2, ... regardless of duplicates
        row.sort()
        best[q] = row
    ids = (best & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    dists = (best >> np.uint64(32)).astype(np.uint32).view(np.float32)
    return ids, dists


def tail_pad_dists(answer_tail, queries):
    """Distances of every query to the last k (= 100 by default) rows of the whole data set.  `answer_tail(q_rows)`
    answers against a data set made of exactly those k rows (ids 0..k-1 = rows n-k..n-1); the
    queries' predicates are stripped so that all k rows are returned."""
    q = np.array(queries, np.float32, copy=True)
    q[:, 0] = 0.0
    q[:, 1:4] = -1.0
    ids, dists = answer_tail(q)
    out = np.empty(dists.shape, np.float32)
    # row n-1-s is tail row k-1-s
    np.put_along_axis(out, (ids.shape[1] - 1 - ids.astype(np.int64)), dists, axis=1)
    return out


def run_data_sharded(answer_shard, row0, n_total, queries, pad_dists, group=None, device="cpu", engine=None):
    """Every rank: ids/dists of ALL queries on its row shard (padding off).  All ranks return the merged
    global answer.  `pad_dists` must be the same on every rank (broadcast it from the tail's owner).
    With `engine` (an Engine on this rank's GPU, tensors on that GPU) the gathered lists are merged by the
    device kernel (hvs_merge_shards_device); otherwise on the host (merge_data_shards)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    ids, dists = answer_shard(queries)
    ids_t = torch.from_numpy(np.ascontiguousarray(ids, np.uint32).view(np.int32)).to(device)
    d_t = torch.from_numpy(np.ascontiguousarray(dists, np.float32)).to(device)
    r0_t = torch.tensor([row0], dtype=torch.int64, device=device)
    nq = ids_t.shape[0]
    # concatenated (not stacked) outputs: the layout every backend's all_gather_into_tensor accepts
    ids_all = torch.empty((world * nq, ids_t.shape[1]), dtype=ids_t.dtype, device=device)
    d_all = torch.empty((world * nq, d_t.shape[1]), dtype=d_t.dtype, device=device)
    r0_all = torch.empty((world,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(ids_all, ids_t, group=group)
    dist.all_gather_into_tensor(d_all, d_t, group=group)
    dist.all_gather_into_tensor(r0_all, r0_t, group=group)
    if engine is not None:
        pad_t = torch.from_numpy(np.ascontiguousarray(pad_dists, np.float32)).to(device)
        out_i = torch.empty((nq, ids_t.shape[1]), dtype=torch.int32, device=device)
        out_d = torch.empty((nq, ids_t.shape[1]), dtype=torch.float32, device=device)
        torch.cuda.synchronize()
        engine.merge_shards_device(ids_all.data_ptr(), d_all.data_ptr(), [int(r) for r in r0_all.cpu()], nq, n_total,
                                   pad_t.data_ptr(), out_i.data_ptr(), out_d.data_ptr())
        engine.sync()
        return out_i.cpu().numpy().view(np.uint32), out_d.cpu().numpy()
    ids_all = ids_all.view(world, nq, -1)
    d_all = d_all.view(world, nq, -1)
    parts = [(ids_all[r].cpu().numpy().view(np.uint32), d_all[r].cpu().numpy(), int(r0_all[r])) for r in range(world)]
    return merge_data_shards(parts, n_total, pad_dists)
