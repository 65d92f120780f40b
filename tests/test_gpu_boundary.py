"""GPU tests of the boundary itself (SURVEY.md §8 rows b / e): the one-process multi-device entry, the exception guard with a real
handle, cloud lifetime across handles, per-stage timers, and the torch.distributed host path with the HIP kernels underneath."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from mandala_mapping_amd import abi, synth

pytestmark = pytest.mark.gpu


def _pairs(n, az=500):
    out = []
    for k in range(n):
        src, tgt, Tgt = synth.hdl32_pair(az + 37 * k, 400 + k, 500 + k, dx=0.2 + 0.02 * k, dy=0.1, dyaw_deg=1.5 + 0.2 * k)
        out.append((src, tgt, synth.perturb(Tgt, np.random.default_rng(k), 0.5, 0.05)))
    return out


def test_multi_context_equals_single_handle_batch(reg, orc):
    """m3dreg_multi_align over devices {0, 0} (two handles, two streams, LPT shards) returns exactly what m3dreg_align_batch
    returns on one handle, which is exactly what the oracle returns."""
    p = abi.Params.make(leaf=0.2, iterations=8, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    pairs = _pairs(5)
    M = reg.MultiRegistrar(p, devices=(0, 0))
    Tm, stm, dev = M.align(pairs)
    R = reg.Registrar(p)
    cl = R.clouds([a for s, t, _ in pairs for a in (s, t)], source_only=[i % 2 == 0 for i in range(2 * len(pairs))])
    Tb, stb = R.align_batch([(cl[2 * i], cl[2 * i + 1], pairs[i][2]) for i in range(len(pairs))])
    assert np.array_equal(Tm, Tb)
    for a, b in zip(stm, stb):
        assert (a.status, a.iterations, a.n_corr, a.rms) == (b.status, b.iterations, b.n_corr, b.rms)
    assert list(dev) == [0] * len(pairs)
    for i in (0, 3):
        To, sto, _ = orc.align(p, orc.Cloud(p, pairs[i][0]), orc.Cloud(p, pairs[i][1]), pairs[i][2])
        assert np.array_equal(Tm[i], To)
    # twice in a row on the same context (pooled blocks, counters back at zero), and a one-pair batch
    Tm2, _, _ = M.align(pairs)
    assert np.array_equal(Tm2, Tm)
    T1, _, _ = M.align(pairs[:1])
    assert np.array_equal(T1[0], Tm[0])
    M.close()


def test_target_groups_are_co_located_and_bucketed_once(reg, orc):
    """m3dreg_pair_desc.target_group (ABI 5; SURVEY 8e: "pairs sharing the same reference cloud are co-located so bucketing is done once"): six
    loop-closure candidates against two submaps over two device contexts — each group stays on one context, its target is uploaded and bucketed
    once (8 clouds instead of 12), and every pose equals the ungrouped call's and the oracle's, bit for bit. Pairs of one group that name different
    payloads are refused."""
    p = abi.Params.make(leaf=0.2, iterations=8, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    tg = []
    pairs, groups = [], []
    for g in range(2):
        _, tgt, _ = synth.hdl32_pair(400, 900 + g, 950 + g, base=(2.0 * g, 1.0, 10.0 * g))
        tg.append(tgt)
        for k in range(3):
            src, _, Tgt = synth.hdl32_pair(400, 900 + g, 960 + 10 * g + k, dx=0.1 + 0.1 * k, dy=0.05 * k, dyaw_deg=1.0 + k, base=(2.0 * g, 1.0, 10.0 * g))
            pairs.append((src, tgt, None)); groups.append(g + 1)
    M = reg.MultiRegistrar(p, devices=[0, 0])
    Tg, stg, _ = M.align(pairs, source_only=True, groups=groups)
    assert M.clouds_bucketed() == 8
    Tu, stu, _ = M.align(pairs, source_only=True)
    assert M.clouds_bucketed() == 12
    assert np.array_equal(Tg, Tu)
    for i, (src, tgt, _) in enumerate(pairs):
        To, sto, _ = orc.align(p, orc.Cloud(p, src, source_only=True), orc.Cloud(p, tgt))
        assert np.array_equal(Tg[i], To) and stg[i].n_corr == sto.n_corr == stu[i].n_corr, i
    # one group larger than a context's share (4 of 6 pairs, capacity 3): still together, still the same bits
    g2 = [1, 1, 1, 0, 0, 0]
    pr2 = [(pairs[i][0], tg[0], None) for i in range(3)] + pairs[3:]
    T2, _, _ = M.align(pr2 + [(pairs[0][0], tg[0], None)], source_only=True, groups=g2 + [1])
    assert M.clouds_bucketed() == 7 + 1 + 3 and np.array_equal(T2[:6], Tg) and np.array_equal(T2[6], Tg[0])
    # different payloads under one group id
    descs, keep = M.describe(pairs[:2], source_only=True)
    descs[0].target_group = descs[1].target_group = 5
    with pytest.raises(abi.M3dregError) as ei:
        M.align_described(descs)
    assert ei.value.code == abi.ERR_INVALID_ARG
    M.close()


def test_multi_align_survives_allocation_failures_and_takes_pinned_payloads(reg):
    """a std::bad_alloc anywhere inside m3dreg_multi_align (the caller's thread or a device thread, before or after
    something was enqueued) comes back as M3DREG_ERR_OUT_OF_MEMORY with every device's handle idle and every cloud released — the
    next call on the same context works and gives the same poses. Pinned payloads (m3dreg_host_alloc: asynchronous DMA) and pageable
    ones (staged through the device threads' pinned blocks) give identical results."""
    p = abi.Params.make(leaf=0.2, iterations=6, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    pairs = _pairs(4)
    L = reg.lib()
    M = reg.MultiRegistrar(p, devices=(0, 0))
    T_ok, st_ok, _ = M.align(pairs, source_only=True)
    failed = 0
    for nth in range(1, 12):
        L.m3dreg_debug_fail_alloc(nth)
        try:
            try:
                T, _, _ = M.align(pairs, source_only=True)
                assert np.array_equal(T, T_ok)          # (nth beyond the allocation points this call reaches)
            except abi.M3dregError as e:
                assert e.code == abi.ERR_OUT_OF_MEMORY, (nth, e)
                failed += 1
        finally:
            L.m3dreg_debug_fail_alloc(0)
        T2, _, _ = M.align(pairs, source_only=True)     # the context is usable again at once: nothing pending, nothing leaked
        assert np.array_equal(T2, T_ok), nth
    assert failed >= 3
    descs, keep = M.describe(pairs, source_only=True, pinned=True)
    Tp, stp, _ = M.align_described(descs)
    assert np.array_equal(Tp, T_ok)
    assert [(a.status, a.iterations, a.n_corr) for a in stp] == [(a.status, a.iterations, a.n_corr) for a in st_ok]
    M.close()


def test_allocation_failure_is_an_error_code_not_an_exception(reg):
    """std::bad_alloc inside m3dreg_cloud_create_batch comes back as M3DREG_ERR_OUT_OF_MEMORY; the handle stays usable."""
    p = abi.Params.make(leaf=0.25, iterations=5, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    R = reg.Registrar(p)
    src, tgt, _ = synth.config1(2000)
    L = reg.lib()
    T_ok, _ = R.align(R.cloud(src), R.cloud(tgt))
    for nth in (1, 2, 3):
        L.m3dreg_debug_fail_alloc(nth)
        try:
            with pytest.raises(abi.M3dregError) as e:
                R.clouds([src, tgt, src])
            assert e.value.code == abi.ERR_OUT_OF_MEMORY
        finally:
            L.m3dreg_debug_fail_alloc(0)
    T2, _ = R.align(R.cloud(src), R.cloud(tgt))
    assert np.array_equal(T2, T_ok)


def test_cloud_freed_right_after_a_foreign_handle_enqueued_its_use(reg, orc):
    """A's clouds are registered on B's stream (enqueue only), freed at once and A buckets new clouds into the recycled blocks:
    A's stream must wait for B's reads (last-use event), or B would register against half-overwritten clouds."""
    p = abi.Params.make(leaf=0.2, iterations=10, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    A, B = reg.Registrar(p), reg.Registrar(p)
    ref = {}
    for seed in range(3):
        src, tgt, _ = synth.hdl32_pair(700, 30 + seed, 40 + seed, dx=0.2, dy=0.1, dyaw_deg=2.0)
        ref[seed] = (src, tgt, orc.align(p, orc.Cloud(p, src), orc.Cloud(p, tgt))[0])
    for rep in range(3):
        for seed in range(3):
            src, tgt, To = ref[seed]
            cs, ct = A.clouds([src, tgt], wait=False)
            arr = B._pairs([(cs, ct, None)])
            B.align_batch_async(arr, 1)
            cs.free(); ct.free()                                   # blocks go back to A's pool while B's batch is in flight
            junk = A.clouds([ref[(seed + 1) % 3][1], ref[(seed + 2) % 3][0]], wait=False)   # same sizes: the pool hands the blocks out again
            T, st = B.batch_wait(1)
            assert np.array_equal(T[0], To), (rep, seed)
            A.synchronize()                                        # (junk's host payloads must outlive their copies)
            del junk
    # destroying through the other handle returns the block to its owner all the same; the owner may close first
    cs, ct = A.clouds([ref[0][0], ref[0][1]])
    reg.lib().m3dreg_cloud_destroy(B._h, cs._p); cs._p = None
    A.close()
    g = abi.GridInfo()
    assert reg.lib().m3dreg_cloud_grid_info(B._h, ct._p, 0, C.byref(g)) == 0 and g.n_valid > 0   # A lives on until its last cloud is gone
    assert reg.lib().m3dreg_cloud_destroy(B._h, ct._p) == 0; ct._p = None


def test_stage_timers(reg):
    p = abi.Params.make(leaf=0.2, iterations=6, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    R = reg.Registrar(p)
    R.profile_enable(True, every=1)
    src, tgt, _ = synth.hdl32_pair(700, 1, 2)
    cs, ct = R.clouds([src, tgt])
    R.align(cs, ct)
    n = {}; ms = {}
    for what in range(4):
        n[what], ms[what] = R.profile_read(what=what, reset=True)
    assert n[0] == 6 and n[1] == 6 and n[3] == 6 and n[2] == 1
    assert all(ms[w] > 0 for w in range(4))
    assert abs((ms[1] + ms[3]) - ms[0]) < 0.05 * ms[0] + 0.05     # correspondence step + reduce/solve = iteration


def test_per_batch_brackets_can_be_sampled(reg):
    """m3dreg_profile_batches (ABI 8): the two events around a bucketing batch and the two around a batch's chain of iterations on every n-th batch only (bench.py: every 4th —
    an event record is a barrier packet on the stream); the default stays every batch, the results do not depend on it."""
    p = abi.Params.make(leaf=0.2, iterations=5, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5, eps_rot=0.0, eps_trans=0.0)
    src, tgt, _ = synth.hdl32_pair(700, 1, 2)
    out = {}
    for every in (1, 3):
        R = reg.Registrar(p)
        R.profile_batches(every)
        R.profile_enable(True, every=1 << 30)      # (no iteration brackets: only the per-batch ones)
        Ts = []
        for _ in range(7):
            cs, ct = R.clouds([src, tgt], source_only=[True, False])
            T, st = R.align(cs, ct)
            Ts.append(T.tobytes() + bytes(st))
            cs.free(); ct.free()
        nb, msb = R.profile_read(2, reset=True)
        nc, msc = R.profile_read(4, reset=True)
        out[every] = (nb, nc, Ts)
        assert msb > 0 and msc > 0 and len(set(Ts)) == 1
        R.close()
    assert out[1][0] == 7 and out[1][1] == 7 * 5          # every batch: 7 bucketing brackets, 7 chains of 5 iterations
    assert out[3][0] == 3 and out[3][1] == 3 * 5          # batches 0, 3 and 6
    assert out[1][2] == out[3][2]
    R = reg.Registrar(p)
    assert reg.lib().m3dreg_profile_batches(R._h, 0) == abi.ERR_INVALID_ARG and reg.lib().m3dreg_profile_batches(None, 2) == abi.ERR_INVALID_ARG


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mandala_mapping_amd import binding, sharding
    p = abi.Params.make(leaf=0.2, iterations=8, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    pairs = _pairs(5)
    R = binding.Registrar(p, device=0)          # every rank shares GPU 0 (one-GPU box): the HIP path under a process group

    def local(idx):
        cl = R.clouds([a for i in idx for a in (pairs[i][0], pairs[i][1])], source_only=[j % 2 == 0 for j in range(2 * len(idx))])
        T, st = R.align_batch([(cl[2 * j], cl[2 * j + 1], pairs[i][2]) for j, i in enumerate(idx)])
        return T, [s.status for s in st]

    costs = [len(s) + len(t) for s, t, _ in pairs]
    T, st = sharding.register_sharded(pairs, costs, local, dist)
    q.put((rank, T, st, sharding.lpt_assign(costs, world)[rank]))
    dist.barrier()
    dist.destroy_process_group()


def test_hip_path_under_a_two_rank_process_group(reg, orc):
    """sharding.register_sharded with Registrar.align_batch per rank, two ranks over gloo sharing the one GPU: the N > 1 host path
    with the real kernels underneath (the 8-GPU run differs only in the backend name and the device index)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    res.sort(key=lambda r: r[0])
    (_, T0, st0, m0), (_, T1, st1, m1) = res
    assert np.array_equal(T0, T1) and sorted(m0 + m1) == list(range(5)) and m0 and m1
    p = abi.Params.make(leaf=0.2, iterations=8, max_corr_dist=0.6, metric=abi.POINT_TO_PLANE, normal_leaf=0.5)
    pairs = _pairs(5)
    for i in (1, 4):
        To, _, _ = orc.align(p, orc.Cloud(p, pairs[i][0]), orc.Cloud(p, pairs[i][1]), pairs[i][2])
        assert np.array_equal(T0[i], To)


@pytest.mark.parametrize("extra,pts", [((), 100000), (("--azimuth", "1000", "--pairs-per-gpu", "3"), 32000)])
def test_bench_launcher_runs_two_ranks_end_to_end(extra, pts):
    """the REAL `bench.py --gpus 2` — its launcher (a fresh process that never touches the GPU) starts two rank processes, which
    rendezvous (gloo; both on this box's one GPU: --share-gpu), take their LPT shards of config 4, run the timed steps, gather the poses and
    per-rank times, and rank 0 prints the one line. What the driver's first SCALE run does, minus RCCL and the second device."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "M3D_BENCH_RANK_PROCESS", "M3D_BENCH_FULL_LINE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--no-extra", "--no-cpu-baseline", "--min-seconds", "0", "--launch-timeout", "110"] + list(extra), capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6000, r.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["steps"] == 2 and d["scaling"] == "weak"
    pr = d["per_rank"]
    B = d["config"]["pairs_per_gpu"]   # (second case: 1000 azimuth steps — no tabulated costs for that, so every rank buckets all pairs once and reads the LPT costs from the DEVICE: m3dreg_cloud_density)
    assert len(pr) == 2 and sorted(p["rank"] for p in pr) == [0, 1] and all(len(p["pairs"]) == B and p["own_work_ms_median"] > 0 for p in pr)
    assert not set(pr[0]["pairs"]) & set(pr[1]["pairs"]) and sorted(pr[0]["pairs"] + pr[1]["pairs"]) == list(range(2 * B))
    assert abs(d["config"]["points_per_cloud"] - pts) < 0.1 * pts
    assert d["max_rot_err_deg"] < 0.2 and d["max_trans_err_m"] < 0.02


def test_bench_line_survives_an_rccl_group_that_does_not_come_up():
    """Round 6: the N > 1 line must not depend on RCCL coming up (it has never run with two ranks: no multi-GPU box in six rounds). Two ranks on this box's ONE GPU
    with the default backend: RCCL refuses ("Duplicate GPU detected"), every rank takes the same decision, the run's three control-plane collectives (barrier, max
    of the ranks' times, gather) go through gloo on host tensors, rank 0 prints the line and the line says so. Under the real launcher (torch.distributed.run:
    the agent hosts the rendezvous store)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "M3D_BENCH_RANK_PROCESS", "M3D_BENCH_FULL_LINE")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
                        os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--azimuth", "800", "--pairs-per-gpu", "3",
                        "--no-extra", "--no-cpu-baseline", "--min-seconds", "0"], capture_output=True, text=True, timeout=170, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6000, r.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(d["per_rank"]) == 2
    assert "gloo (RCCL group failed)" in d["config"]["parallelism"], d["config"]["parallelism"]
    assert "the RCCL process group failed" in r.stderr
