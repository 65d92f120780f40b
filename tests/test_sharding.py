"""The N > 1 host path on CPU: LPT sharding + the single pose all_gather, world_size 2 over gloo.
The per-rank registration function is the CPU oracle here (tests may use it); on the GPU box bench.py runs
the same sharding code with Registrar.align_batch per rank over RCCL."""
import os
import socket

import numpy as np
import pytest

from mandala_mapping_amd import sharding


def test_lpt_assignment_properties():
    costs = [100, 90, 80, 10, 10, 10, 5, 5]
    sh = sharding.lpt_assign(costs, 3)
    assert sorted(i for s in sh for i in s) == list(range(8))
    loads = [sum(costs[i] for i in s) for s in sh]
    assert max(loads) - min(loads) <= 20
    # pairs sharing a target stay together
    sh = sharding.lpt_assign([10] * 6, 2, groups=[0, 0, 1, 1, 2, 2])
    for s in sh:
        for gpair in ((0, 1), (2, 3), (4, 5)):
            assert (gpair[0] in s) == (gpair[1] in s)
    assert sharding.lpt_assign([], 4) == [[], [], [], []]
    assert sharding.lpt_assign([1.0], 1) == [[0]]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mandala_mapping_amd import abi, synth
    from oracle import orc
    p = abi.Params.make(leaf=0.25, iterations=5, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    pairs, costs = [], []
    for k in range(5):
        tgt = synth.planes_cloud(1500 + 400 * k, 50 + k)
        Tg = synth.random_T(np.random.default_rng(k), 2.0, 0.1)
        src = synth.apply_T(synth.inv_T(Tg), synth.planes_cloud(1200 + 300 * k, 80 + k)).astype(np.float32)
        pairs.append((src, tgt))
        costs.append(len(src) + len(tgt))

    def local(idx):
        Ts, sts = [], []
        for i in idx:
            T, st, _ = orc.align(p, orc.Cloud(p, pairs[i][0]), orc.Cloud(p, pairs[i][1]))
            Ts.append(T)
            sts.append(st.status)
        return np.stack(Ts), sts

    T, st = sharding.register_sharded(pairs, costs, local, dist)
    mine = sharding.lpt_assign(costs, world)[rank]
    # the pipelined form bench.py uses: two batches' gathers in flight, read one batch later
    t1 = sharding.gather_results_start([rank], np.eye(4)[None] * (rank + 1), [rank], world, dist)
    t2 = sharding.gather_results_start([rank], np.eye(4)[None] * (rank + 10), [rank + 5], world, dist)
    Ta, sa = sharding.gather_results_finish(t1)
    Tb, sb = sharding.gather_results_finish(t2)
    assert [Ta[r, 0, 0] for r in range(world)] == [r + 1 for r in range(world)] and list(sa) == list(range(world))
    assert [Tb[r, 0, 0] for r in range(world)] == [r + 10 for r in range(world)] and list(sb) == [r + 5 for r in range(world)]
    q.put((rank, T, st, mine))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_over_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    (_, T0, st0, m0), (_, T1, st1, m1) = res
    assert np.array_equal(T0, T1) and np.array_equal(st0, st1)          # every rank holds the full result
    assert sorted(m0 + m1) == list(range(5)) and m0 and m1               # work was really split
    assert (st0 >= 0).all()
    # and it equals the single-process result
    from mandala_mapping_amd import abi, synth
    from oracle import orc
    p = abi.Params.make(leaf=0.25, iterations=5, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    k = 3
    tgt = synth.planes_cloud(1500 + 400 * k, 50 + k)
    Tg = synth.random_T(np.random.default_rng(k), 2.0, 0.1)
    src = synth.apply_T(synth.inv_T(Tg), synth.planes_cloud(1200 + 300 * k, 80 + k)).astype(np.float32)
    T, _, _ = orc.align(p, orc.Cloud(p, src), orc.Cloud(p, tgt))
    assert np.array_equal(T0[k], T)


def test_lpt_assignment_with_capacity():
    """Weak scaling: every rank gets the same number of pairs; the expensive ones are spread first."""
    costs = [9.0, 1.0, 1.0, 1.0, 8.0, 1.0, 1.0, 7.0]
    sh = sharding.lpt_assign(costs, 2, capacity=4)
    assert sorted(sum(sh, [])) == list(range(8)) and [len(x) for x in sh] == [4, 4]
    assert 0 in sh[0] and 4 in sh[1]                       # the two most expensive pairs never share a rank
    loads = sorted(sum(costs[i] for i in x) for x in sh)
    assert loads == [12.0, 17.0]                            # the best any 4 + 4 split of these costs can do
    assert sharding.lpt_assign(costs, 2, capacity=4) == sh  # deterministic: every rank derives the same table
    import pytest
    with pytest.raises(ValueError):
        sharding.lpt_assign(costs, 2, capacity=3)


def test_config4_cost_table_matches_the_generator():
    """bench.py derives every rank's shard of BASELINE config 4 from mandala_mapping_amd/config4_costs.json instead of ray-casting all 64 pairs on every rank:
    the table must be what the generator gives (three pairs recomputed here), and give every rank 8 pairs at N = 8."""
    import json
    import os
    from mandala_mapping_amd import synth
    tab = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mandala_mapping_amd", "config4_costs.json")))
    assert tab["azimuth"] == 3125 and len(tab["costs"]) == 64
    for k in (0, 17, 63):
        s, t, _ = synth.config4_pair(k, 3125)
        assert abs(tab["costs"][k] - (synth.crowdedness(s) + synth.crowdedness(t))) < 1e-4
    sh = sharding.lpt_assign(tab["costs"], 8, capacity=8)
    assert sorted(sum(sh, [])) == list(range(64)) and all(len(x) == 8 for x in sh)
    loads = [sum(tab["costs"][i] for i in x) for x in sh]
    # one pair of the 64 costs twice what the others do (22.6 against 11.3 - 18): with 8 pairs per rank the rank that holds it cannot get below that pair
    # plus the seven cheapest ones — the assignment must reach that bound (LPT + swap refinement), and the other ranks must lie below it
    c = sorted(tab["costs"])
    assert max(loads) <= c[-1] + sum(c[:7]) + 0.2 and max(loads) / min(loads) < 1.09


@pytest.mark.gpu
def test_pose_gather_on_the_gpu_under_a_single_rank_nccl_group():
    """The pose gather's GPU path (pinned record blocks, its own stream, RCCL all_gather_into_tensor, non-blocking copy back, event) has only ever a
    one-GPU box to run on: a single-rank `nccl` group exercises every call of it — several gathers in flight, results in pair order."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n = 8
        tickets = []
        for step in range(3):
            T = np.stack([np.eye(4) * (step + 1) + k for k in range(n)])
            tickets.append(sharding.gather_results_start(list(range(n))[::-1], T[::-1], [k + step for k in range(n)][::-1], n, dist, dev))
        for step, tk in enumerate(tickets):
            Tg, st = sharding.gather_results_finish(tk)
            assert [Tg[k, 0, 0] for k in range(n)] == [step + 1 + k for k in range(n)] and list(st) == [k + step for k in range(n)]
        g = sharding._gather_for(n, dist, dev)
        assert g.on_gpu and g.slots[0]["rec_h"].is_pinned() and g.slots[0]["out_h"].is_pinned()
        tt = torch.tensor([1.25], dtype=torch.float64, device=dev)      # bench.py's max-over-ranks of a block time
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.barrier()
        assert float(tt.item()) == 1.25
    finally:
        dist.destroy_process_group()
