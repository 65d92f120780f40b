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
    each list heaviest item first (deterministic for equal costs).
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
    if capacity is not None and len(units) == n:
        _balance_by_swaps(out, load, costs)
    # every rank's items HEAVIEST FIRST (ties: ascending index — deterministic): a batch's workgroups are dispatched pair after pair, and a crowded pair that comes last is the
    # tail of every launch (the eight LPT shards of config 4, heaviest pair first / last: 1.5 - 4 % apart, profiles/r06_pair_order.txt)
    return [sorted(x, key=lambda i: (-costs[i], i)) for x in out]


def table_costs(table: dict, n: int) -> Tuple[List[float], str]:
    """(costs, source) of the first n pairs of a tabulated workload (mandala_mapping_amd/config4_costs.json): the MEASURED additive per-pair costs when the
    table holds them ("measured_ms": least-squares fit of the step times of random 8-pair batches on an MI355X, scripts/measure_pair_costs.py), else the
    a-priori estimate density(source) + density(target) ("costs")."""
    m = table.get("measured_ms")
    if m and len(m) >= n:
        return [float(x) for x in m[:n]], "measured_ms (scripts/measure_pair_costs.py)"
    return [float(x) for x in table["costs"][:n]], "a-priori density estimate"


def _balance_by_swaps(out, load, costs, rounds: int = 1000):
    """Greedy LPT under an equal-count capacity leaves the ranks' estimated loads several per cent apart (config 4: 9 %), and at N ranks the step
    takes what the HEAVIEST rank takes. Local search on top of it: swap one item of the heaviest rank against one of another rank whenever that lowers
    the larger of the two loads the most; stop when no swap helps. Counts per rank never change; deterministic (first best swap in index order)."""
    for _ in range(rounds):
        a = max(range(len(out)), key=lambda k: (load[k], -k))
        best = None
        for b in range(len(out)):
            if b == a:
                continue
            for i in out[a]:
                for j in out[b]:
                    d = costs[i] - costs[j]
                    if d <= 0:
                        continue
                    new_max = max(load[a] - d, load[b] + d)
                    if new_max < load[a] - 1e-12 and (best is None or new_max < best[0] - 1e-12):
                        best = (new_max, b, i, j)
        if best is None:
            return
        _, b, i, j = best
        out[a].remove(i); out[b].remove(j)
        out[a].append(j); out[b].append(i)
        d = costs[i] - costs[j]
        load[a] -= d; load[b] += d


class PoseGather:
    """The one collective of the path, with everything it touches allocated ONCE: a ring of pinned host record blocks, their
    device twins, the gathered block on the device and its pinned host copy, and a stream of its own. start() fills a pinned block
    on the host and enqueues copy -> all_gather -> copy back on that stream without blocking the host (no pageable copy — those are
    synchronous — and nothing on torch's default stream, which the registrations of a step do not use either); finish() waits for the
    slot's event and unpacks. With gloo (CPU tensors, the tests and the one-GPU rehearsal) the same calls run on host memory."""

    def __init__(self, n_total: int, dist, device=None, slots: int = 4):
        import torch
        self.torch, self.dist, self.device, self.n_total = torch, dist, device, n_total
        self.world = dist.get_world_size()
        self.cap = n_total   # upper bound on a shard; a record is 144 bytes, so 64 pairs x 8 ranks is ~74 KB in total
        self.on_gpu = device is not None
        self.slots, self.next = [], 0
        for _ in range(slots):
            rec_h = torch.empty((self.cap, 18), dtype=torch.float64)
            out_h = torch.empty((self.world * self.cap, 18), dtype=torch.float64)
            if self.on_gpu:
                rec_h, out_h = rec_h.pin_memory(), out_h.pin_memory()
                slot = dict(rec_h=rec_h, out_h=out_h, rec_d=torch.empty_like(rec_h, device=device), out_d=torch.empty_like(out_h, device=device),
                            ev=torch.cuda.Event(), busy=False)
            else:
                slot = dict(rec_h=rec_h, out_h=out_h, rec_d=rec_h, out_d=out_h, ev=None, busy=False)
            slot["rec_np"] = rec_h.numpy()   # (shares the pinned memory)
            self.slots.append(slot)
        self.stream = torch.cuda.Stream(device=device) if self.on_gpu else None

    def start(self, local_idx: Sequence[int], local_T: np.ndarray, local_status: Sequence[int]):
        s = self.slots[self.next]
        self.next = (self.next + 1) % len(self.slots)
        if s["busy"]:
            raise RuntimeError("PoseGather: more gathers in flight than slots (finish the oldest first)")
        a = s["rec_np"]
        a.fill(-1.0)
        k = len(local_idx)
        if k:
            a[:k, 0] = np.asarray(local_idx, np.float64)
            a[:k, 1] = np.asarray(local_status, np.float64)
            a[:k, 2:] = np.asarray(local_T, np.float64).reshape(k, 16)
        if self.on_gpu:
            torch = self.torch
            with torch.cuda.stream(self.stream):
                s["rec_d"].copy_(s["rec_h"], non_blocking=True)
                work = self.dist.all_gather_into_tensor(s["out_d"], s["rec_d"], async_op=True)
                work.wait()                                     # orders THIS stream behind the collective; the host does not block
                s["out_h"].copy_(s["out_d"], non_blocking=True)
                s["ev"].record(self.stream)
            s["work"] = work
        else:
            s["work"] = self.dist.all_gather_into_tensor(s["out_d"], s["rec_d"], async_op=True)
        s["busy"] = True
        return s

    def finish(self, s) -> Tuple[np.ndarray, np.ndarray]:
        if self.on_gpu:
            s["ev"].synchronize()
        else:
            s["work"].wait()
        s["busy"] = False
        a = s["out_h"].numpy()
        T = np.zeros((self.n_total, 4, 4))
        st = np.full(self.n_total, -1, np.int64)
        rows = a[a[:, 0] >= 0]
        idx = rows[:, 0].astype(np.int64)
        T[idx] = rows[:, 2:].reshape(-1, 4, 4)
        st[idx] = rows[:, 1].astype(np.int64)
        return T, st


_GATHERS = {}


def _gather_for(n_total, dist, device):
    key = (n_total, str(device), dist.get_world_size(), dist.get_rank())
    g = _GATHERS.get(key)
    if g is None:
        g = _GATHERS[key] = PoseGather(n_total, dist, device)
    return g


def gather_results_start(local_idx: Sequence[int], local_T: np.ndarray, local_status: Sequence[int], n_total: int, dist, device=None):
    """Enqueue the all-gather of this rank's poses / status words and return a ticket for gather_results_finish.
    ONE pinned host->device copy, ONE collective, ONE copy back, all on the gather's own stream (PoseGather); nothing here waits for
    the other ranks or for the device, so a caller that pipelines its batches (bench.py) is not forced into lock-step with the slowest
    rank at every batch."""
    g = _gather_for(n_total, dist, device)
    return (g, g.start(local_idx, local_T, local_status))


def gather_results_finish(ticket) -> Tuple[np.ndarray, np.ndarray]:
    """Wait for the collective of gather_results_start and unpack it into pair order: ([n_total,4,4], [n_total])."""
    g, slot = ticket
    return g.finish(slot)


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
