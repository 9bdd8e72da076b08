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


def run_sharded(answer, queries: np.ndarray, group=None, device="cpu"):
    """answer(q_rows) -> (n_local x 100) uint32 ids for this rank's queries.  Returns all ids."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    q0, q1 = shard_range(queries.shape[0], rank, world)
    ids = np.ascontiguousarray(answer(queries[q0:q1]), np.uint32)
    t = torch.from_numpy(ids.view(np.int32)).to(device)
    return gather_ids(t, queries.shape[0], group).cpu().numpy().view(np.uint32)
