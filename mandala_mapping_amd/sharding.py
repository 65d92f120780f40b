"""Multi-GPU host logic: independent scan-pair registrations sharded over the ranks of one node.

The path shards naturally: pairs never exchange data, so there is NO collective in the
data path. One process drives one GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests); the only collective is the final gather of the 4x4 poses + status words
(a few KB for 64 pairs — latency-bound, one call per batch, never per iteration).
"""
from typing import Callable, List, Sequence, Tuple

import numpy as np


def lpt_assign(costs: Sequence[float], n_ranks: int, groups: Sequence[int] = None, capacity: int = None) -> List[List[int]]:
    """Longest-processing-time-first assignment of work items to ranks.
    capacity (optional): at most this many items per rank (weak scaling: every GPU gets the same number of pairs per step).

    costs[i]: estimated cost of pair i (N_src + N_tgt is a good proxy: both the bucketing and the NN search
    are linear in the cloud sizes). groups[i] (optional): pairs with the same group id share a target
    cloud and are kept on one rank so that cloud is bucketed once. Returns the item indices of every rank,
    each list in ascending order (deterministic for equal costs).
    """
    n = len(costs)
    if groups is None:
        groups = list(range(n))
    units = {}
    for i in range(n):
        units.setdefault(groups[i], []).append(i)
    order = sorted(units.values(), key=lambda u: (-sum(costs[i] for i in u), u[0]))
    load = [0.0] * n_ranks
    out: List[List[int]] = [[] for _ in range(n_ranks)]
    for u in order:
        fits = [k for k in range(n_ranks) if capacity is None or len(out[k]) + len(u) <= capacity]
        if not fits:
            raise ValueError("lpt_assign: the items do not fit the ranks' capacity")
        r = min(fits, key=lambda k: (load[k], k))
        out[r].extend(u)
        load[r] += sum(costs[i] for i in u)
    return [sorted(x) for x in out]


def gather_results_start(local_idx: Sequence[int], local_T: np.ndarray, local_status: Sequence[int], n_total: int, dist, device=None):
    """Enqueue the all-gather of this rank's poses / status words and return a ticket for gather_results_finish.
    ONE host->device copy and ONE collective; nothing here waits for the other ranks, so a caller that pipelines its
    batches (bench.py) is not forced into lock-step with the slowest rank at every batch."""
    import torch
    world = dist.get_world_size()
    cap = n_total   # upper bound on a shard; a record is 144 bytes, so 64 pairs x 8 ranks is ~74 KB in total
    # the record block is assembled on the host (element-wise writes into a device tensor cost a launch each)
    rec_h = np.full((cap, 18), -1.0, dtype=np.float64)
    for j, i in enumerate(local_idx):
        rec_h[j, 0] = float(i)
        rec_h[j, 1] = float(local_status[j])
        rec_h[j, 2:] = np.asarray(local_T[j], np.float64).reshape(16)
    rec = torch.from_numpy(rec_h).to(device) if device is not None else torch.from_numpy(rec_h)
    out = torch.empty((world * cap, 18), dtype=torch.float64, device=rec.device)
    work = dist.all_gather_into_tensor(out, rec, async_op=True)
    return (work, out, rec, n_total)


def gather_results_finish(ticket) -> Tuple[np.ndarray, np.ndarray]:
    """Wait for the collective of gather_results_start and unpack it into pair order: ([n_total,4,4], [n_total])."""
    work, out, _rec, n_total = ticket
    work.wait()
    a = out.cpu().numpy()   # ONE device->host copy
    T = np.zeros((n_total, 4, 4))
    st = np.full(n_total, -1, np.int64)
    rows = a[a[:, 0] >= 0]
    idx = rows[:, 0].astype(np.int64)
    T[idx] = rows[:, 2:].reshape(-1, 4, 4)
    st[idx] = rows[:, 1].astype(np.int64)
    return T, st


def gather_results(local_idx: Sequence[int], local_T: np.ndarray, local_status: Sequence[int], n_total: int,
                   dist=None, device=None) -> Tuple[np.ndarray, np.ndarray]:
    """All-gather the poses ([k,4,4]) and status words of every rank into pair order.

    Each rank contributes a fixed-size record block (padded to the largest shard) of
    [pair index, status, 16 pose floats]; one `all_gather` moves everything. Returns ([n_total,4,4], [n_total]).
    """
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        T = np.zeros((n_total, 4, 4))
        st = np.full(n_total, -1, np.int64)
        for j, i in enumerate(local_idx):
            T[i] = local_T[j]
            st[i] = local_status[j]
        return T, st
    return gather_results_finish(gather_results_start(local_idx, local_T, local_status, n_total, dist, device))


def register_sharded(pairs: Sequence, costs: Sequence[float], register_local: Callable[[List[int]], Tuple[np.ndarray, List[int]]],
                     dist=None, device=None, groups: Sequence[int] = None):
    """Shard `pairs` over the ranks (LPT), run `register_local(indices)` on this rank's shard — in production
    that is Registrar.align_batch on the rank's GPU — and gather all poses. Every rank returns the full result."""
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    shards = lpt_assign(costs, world, groups)
    mine = shards[rank]
    T_local, st_local = register_local(mine) if mine else (np.zeros((0, 4, 4)), [])
    return gather_results(mine, T_local, st_local, len(pairs), dist, device)
