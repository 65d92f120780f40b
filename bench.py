#!/usr/bin/env python3
"""bench.py — scan-pair registrations/s on synthetic Velodyne-HDL-32-shaped 100k-point clouds.

A "step" = one batch of --pairs-per-gpu independent scan-pair registrations per GPU (BASELINE config 4:
64 loop-closure pairs over 8 GPUs = 8 pairs per GPU; at N=1 the same 8-pair batch on one GPU). For every
pair the timed region covers the whole job on data already resident in HBM as PointCloud2 payloads:
decode + exact AABB + voxel bucketing + normals of BOTH clouds, then --iters point-to-plane Gauss-Newton
iterations (0.1 m voxel NN), then the pose read-back; with N > 1 the poses of all ranks are gathered
over RCCL (one all_gather per step — the only collective; pairs never exchange data).

Prints ONE JSON line (rank 0). `roofline` prices the dominant kernel (k_nn_iter: source transform, NN-certificate
check and exact 27-voxel nearest-neighbour search of every query of every pair of the batch) by its algorithmic
bytes (NN part: 12 N + 12 M + 8 C_occ per pair) over its average launch duration, taken from
hipEvents the library records on its stream around launches of that kernel inside the timed region (every seventh
launch: an event record is a barrier packet that costs ~6 us on the queue, bracketing all of them cost 4 % of the
throughput being measured). `cpu_baseline` is the CPU oracle (OpenMP build)
timed on a bounded sample of the same workload on this host's cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-gpu", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20, help="fixed Gauss-Newton iterations per registration")
    ap.add_argument("--azimuth", type=int, default=3125, help="azimuth steps per sweep (x32 beams = points)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--from-host", action="store_true",
                    help="hand over HOST PointCloud2 buffers every step (PCIe-inclusive rate; reported in DESIGN.md, never the headline)")
    ap.add_argument("--converge", action="store_true",
                    help="terminate on eps 1e-5 (max --iters) instead of running a fixed iteration count; secondary figure, see DESIGN.md")
    ap.add_argument("--inflight", type=int, default=3,
                    help="steps in flight: step i runs on handle/stream i %% D, the host enqueues step i+D-1 before waiting for step i (1 = strictly serial steps)")
    ap.add_argument("--queue-depth", type=int, default=2,
                    help="steps QUEUED per stream (the first of them runs, the others wait behind it on the same stream): with 2 a stream never "
                         "drains while the host collects a finished step and enqueues the next one")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real thing); gloo + --share-gpu = rehearsal of the N > 1 flow on a single-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--event-every", type=int, default=7, help="bracket every n-th iteration with hipEvents (n should not divide --iters)")
    ap.add_argument("--no-events", action="store_true", help="do not record hipEvents in the timed region (A/B of their cost; roofline then reads 0)")
    ap.add_argument("--trace-host", action="store_true", help="print the host-side time of every enqueue (bucketing / launch) and wait of the timed region to stderr")
    ap.add_argument("--pair-offset", type=int, default=None,
                    help="index of this rank's first scan pair (default rank * pairs-per-gpu): lets one GPU run the shard another rank gets at N > 1")
    ap.add_argument("--pair-list", default=None, help="comma-separated indices of this rank's scan pairs (diagnostics: run any shard of any N on one GPU)")
    ap.add_argument("--shard", default="lpt", choices=["lpt", "consecutive"],
                    help="N > 1: which pairs a rank gets. lpt = the world_size * pairs-per-gpu pairs dealt longest-processing-time-first by an "
                         "a-priori cost estimate (synth.crowdedness of both clouds), the same number to every rank (SURVEY 8d config 4: "
                         "LPT-sharded); consecutive = rank r takes pairs r * B ... r * B + B - 1")
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the cpu_baseline leg (0 = min(cores, 64))")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from mandala_mapping_amd import abi, binding, sharding, synth
    from mandala_mapping_amd.pointcloud2 import encode_xyz

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else None   # where the collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    B, K, W = args.pairs_per_gpu, args.steps, args.warmup
    # fixed iteration count: eps = 0 never triggers, so every launch of the dominant kernel does full work
    params = abi.Params.make(leaf=0.1, iterations=args.iters, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE,
                             normal_leaf=0.4, eps_rot=1e-5 if args.converge else 0.0, eps_trans=1e-5 if args.converge else 0.0)
    # --inflight D HIP streams; step i runs entirely (bucketing + iterations) on stream i % D.
    # The host enqueues the next steps while step i is still iterating, so D 8-pair chains share the GPU:
    # every iteration kernel is latency-bound, a second chain fills the idle CUs. Every step still
    # does all of its own work inside the timed region; steps merely overlap in time.
    D = max(1, args.inflight)
    Q = max(1, args.queue_depth)
    streams = [torch.cuda.Stream(device=dev) for _ in range(D)]
    # D * Q handles: handle j runs on stream j % D, so up to Q steps queue up behind each other on a stream (a handle holds the
    # state of one batch). Nothing in a step's chain — bucketing (m3dreg_cloud_create_batch_async) or iterations — waits for the
    # host, so a queued step starts the moment the one ahead of it on its stream ends.
    regs = [binding.Registrar(params, device=local_rank, stream=C.c_void_p(streams[j % D].cuda_stream)) for j in range(D * Q)]
    bregs = regs
    reg = regs[0]

    # ---- synthetic workload: this rank's B pairs, resident in HBM as PointCloud2 payloads -------------
    payloads, gts, host_pairs, host_msgs = [], [], [], []
    if args.pair_list is not None:
        pair_ids, generated = [int(x) for x in args.pair_list.split(",")], {}
        assert len(pair_ids) == B, "--pair-list must name --pairs-per-gpu pairs"
    elif args.pair_offset is not None:
        pair_ids, generated = [args.pair_offset + i for i in range(B)], {}
    elif world > 1 and args.shard == "lpt":
        # every rank generates all pairs and derives the same assignment: no communication, deterministic
        generated = {k: synth.config4_pair(k, args.azimuth) for k in range(world * B)}
        costs = [synth.crowdedness(generated[k][0]) + synth.crowdedness(generated[k][1]) for k in range(world * B)]
        pair_ids = sharding.lpt_assign(costs, world, capacity=B)[rank]
        generated = {k: generated[k] for k in pair_ids}
    else:
        pair_ids, generated = [rank * B + i for i in range(B)], {}
    for i in range(B):
        src, tgt, Tgt = generated[pair_ids[i]] if pair_ids[i] in generated else synth.config4_pair(pair_ids[i], args.azimuth)
        ms, mt = encode_xyz(src), encode_xyz(tgt)
        ds = torch.frombuffer(bytearray(ms.data), dtype=torch.uint8).to(dev)
        dt = torch.frombuffer(bytearray(mt.data), dtype=torch.uint8).to(dev)
        payloads.append((ds, ms.n, dt, mt.n))
        host_msgs += [ms, mt]
        gts.append(Tgt)
        if rank == 0 and i < 8:
            host_pairs.append((src, tgt))
    torch.cuda.synchronize()

    last = {}

    def make_clouds(r):
        """decode + AABB + bucketing + normals of this rank's 2B clouds, one batched pipeline on r's stream"""
        if args.from_host:
            cl = r.clouds(host_msgs, wait=False, source_only=[True, False] * B)   # host buffers (they outlive the step) cross PCIe inside the timed region
        else:
            items = []
            for ds, ns, dt, nt in payloads:
                items += [(ds.data_ptr(), ns), (dt.data_ptr(), nt)]
            cl = r.clouds_from_device(items, wait=False, source_only=[True, False] * B)   # no host synchronisation anywhere in a step's chain; sources: no normals
        return [(cl[2 * i], cl[2 * i + 1]) for i in range(len(payloads))]

    def finish(T, st, clouds):
        if world > 1:   # the only collective of the path: one all_gather of poses + status per step (RCCL over xGMI).
            # It is enqueued here and read one step later (or at the end of the timed region): the ranks exchange every
            # step's results without falling into lock-step at every step.
            ticket = sharding.gather_results_start(pair_ids, T, [x.status for x in st], world * B, dist, cdev)
            drain_gather()
            last["pending_gather"] = ticket
        last["T"], last["st"] = T, st

    def drain_gather():
        if last.get("pending_gather") is not None:
            last["all"] = sharding.gather_results_finish(last.pop("pending_gather"))

    host_log = []

    def enqueue(i):
        r = regs[i % len(regs)]
        ta = time.perf_counter()
        clouds = make_clouds(bregs[i % len(bregs)])
        tb = time.perf_counter()
        r.align_batch_async(r._pairs([(s_, t_, None) for s_, t_ in clouds]), B)
        host_log.append(("enq", i, 1e3 * (tb - ta), 1e3 * (time.perf_counter() - tb)))
        return clouds

    def run_steps(k, keep_clouds=False):
        """k steps, at most D * Q enqueued: a step's bucketing and iterations queue up on its stream before the host waits for
        an earlier one. A finished step's clouds go back to their handle's block pool BEFORE the next step is enqueued on it,
        so the pools reach their steady state with the first step of every handle (no hipMalloc inside the timed region)."""
        pending, nxt = [], 0
        while nxt < min(len(regs), k):
            pending.append((nxt, enqueue(nxt))); nxt += 1
        clouds = None
        for i in range(k):
            idx, clouds = pending.pop(0)
            ta = time.perf_counter()
            T, st = regs[idx % len(regs)].batch_wait(B)
            host_log.append(("wait", idx, 1e3 * (time.perf_counter() - ta), 0.0))
            finish(T, st, clouds)
            if not (keep_clouds and i == k - 1):
                for s_, t_ in clouds:
                    s_.free(); t_.free()
                clouds = None
            if nxt < k:
                pending.append((nxt, enqueue(nxt))); nxt += 1
        drain_gather()   # the last step's poses are on every rank before the timed region ends
        return clouds

    def step():
        return run_steps(1, keep_clouds=True)

    # algorithmic bytes of one launch of the dominant kernel, from the real clouds
    clouds0 = step()
    alg_bytes = 0       # NN kernel: source xyz + target xyz + occupied-voxel table, each touched once
    alg_bytes_iter = 0  # whole linearisation: + target normals (point-to-plane)
    for (s, t) in clouds0:
        g = t.grid_info()
        alg_bytes += 12 * s.n + 12 * t.n + 8 * g.n_cells
        alg_bytes_iter += 12 * s.n + 12 * t.n + 8 * g.n_cells + 12 * t.n
    n_pts = int(np.mean([s.n for s, _ in clouds0]))
    del clouds0
    last.clear()
    run_steps(max(W - 1, len(regs) if W > 0 else 0))   # untimed; at least one step per handle so that every pool is allocated
    last.clear()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for r in regs:
        r.profile_enable(not args.no_events, every=args.event_every)   # 7 does not divide the 20 iterations of a step: every iteration index gets sampled
        r.profile_read(0, reset=True)
        r.profile_read(1, reset=True)
    import gc
    gc.collect()
    gc.disable()   # the host thread only enqueues and collects; a generation-2 collection in the middle of the region is a 40 ms stall (measured: -9 % at 300 steps)
    barrier()
    host_log.clear()
    t0 = time.perf_counter()
    run_steps(K)
    barrier()
    t1 = time.perf_counter()
    gc.enable()
    if args.trace_host and rank == 0:
        for what, i, a, b in host_log:
            print(f"[host] {what} step {i}: {a:.3f} ms" + (f" bucketing, {b:.3f} ms enqueue of the iterations" if what == "enq" else ""), file=sys.stderr)
    launches = kern_ms = iters_timed = iter_ms = 0
    for r in regs:
        a, b = r.profile_read(1, reset=True)       # k_nn_iter alone
        c, d = r.profile_read(0, reset=True)       # search + reduction of one linearisation
        launches, kern_ms, iters_timed, iter_ms = launches + a, kern_ms + b, iters_timed + c, iter_ms + d
        r.profile_enable(False)
    # After the timed region, untimed: the same kernel with NOTHING else on the GPU (one step, one handle, every launch bracketed).
    # With several chains sharing the GPU a launch takes longer although more launches complete per second; this is the kernel's own
    # duration, reported beside the contract's figure as roofline.alone.
    alone_ms = 0.0
    if not args.no_events and world == 1:
        last_T, last_st = last.get("T"), last.get("st")
        regs[0].profile_enable(True, every=1)
        regs[0].profile_read(1, reset=True)
        c_ = make_clouds(regs[0])
        regs[0].align_batch_async(regs[0]._pairs([(s_, t_, None) for s_, t_ in c_]), B)
        regs[0].batch_wait(B)
        a_, b_ = regs[0].profile_read(1, reset=True)
        regs[0].profile_enable(False)
        alone_ms = b_ / max(1, a_)
        del c_
        last["T"], last["st"] = last_T, last_st
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev if cdev is not None else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # sanity of the timed work: poses against the generator's ground truth
    errs = [synth.pose_error(last["T"][i], gts[i]) for i in range(B)]
    max_rot, max_tr = max(e[0] for e in errs), max(e[1] for e in errs)

    if rank == 0:
        total_regs = world * B * K
        value = total_regs / elapsed
        avg_launch_s = (kern_ms / 1e3) / max(1, launches)
        achieved = alg_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("k_nn_iter", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "scan-pair registrations/sec (100k-pt clouds, point-to-plane, 0.1 m voxel NN)",
            "value": value, "unit": "registrations/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (int64 fixed-point sums, f64 solve)", "data": "synthetic",
            "config": {"workload": f"BASELINE config 4 shard: {B} HDL-32-shaped scan pairs per GPU per step, "
                                   f"{n_pts} pts/cloud (100000 rays per sweep, as m3d_aggregator publishes them: its +-1 m self-filter box applied), "
                                   f"point-to-plane, leaf 0.1 m, {args.iters} fixed iterations, "
                                   "decode + bucketing of both clouds and the normals of the target (the source is sorted only: m3dreg_cloud_desc.source_only) inside the timed region",
                       "pairs_per_gpu": B, "points_per_cloud": n_pts, "iterations": args.iters,
                       "parallelism": f"pairs sharded over {world} GPU(s)" + (f" ({args.shard})" if world > 1 else "") + ", one all_gather of poses per step",
                       "overlap": ("none (serial steps)" if D == 1 else f"{D} steps run concurrently, one HIP stream each") +
                                  (f"; {Q} steps queued per stream" if Q > 1 else "")},
            "ms_per_icp_iter_batch": iter_ms / max(1, iters_timed),
            "ms_per_icp_iter_per_pair": iter_ms / max(1, iters_timed) / B,
            "iteration_algorithmic_GBps": (alg_bytes_iter / (iter_ms / max(1, iters_timed) / 1e3) / 1e9) if iter_ms > 0 else 0.0,
            "max_rot_err_deg": max_rot, "max_trans_err_m": max_tr,
            "iterations_executed_pair0": int(last["st"][0].iterations),
            "roofline": {"bound": "hbm", "kernel": "k_nn_iter", "concurrent_chains": D, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": 1e3 * avg_launch_s, "launches_timed": launches,
                         "alone": ({"avg_launch_ms": alone_ms, "achieved": alg_bytes / (alone_ms / 1e3) / 1e9, "frac": alg_bytes / (alone_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                                    "note": "the same kernel with nothing else on the GPU: one extra, untimed step after the timed region, every launch bracketed"}
                                   if alone_ms > 0 else None)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params, host_pairs, args.iters, args.cpu_threads)
        if args.from_host:
            out["data"] = "synthetic (host PointCloud2 buffers: PCIe-inclusive, not the headline configuration)"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(params, host_pairs, iters, threads=0):
    """The CPU oracle (OpenMP build: normals and the per-point NN/accumulate loop are parallel, the sort
    and the 6x6 solve are serial) on a bounded sample of the same pairs."""
    from oracle import orc
    threads = threads or min(os.cpu_count() or 1, 64)   # beyond ~64 threads the per-iteration fork/join outweighs the work
    orc.build()
    threads = orc.set_threads(threads)   # torch has already initialised libgomp: the env var alone would be ignored
    done, t0 = 0, time.perf_counter()
    for src, tgt in host_pairs:
        cs, ct = orc.Cloud(params, src, omp=True), orc.Cloud(params, tgt, omp=True)
        orc.align(params, cs, ct)
        done += 1
        if time.perf_counter() - t0 > 20.0:
            break
    dt = time.perf_counter() - t0
    kd = None
    try:
        kd = cpu_kdtree_baseline(params, host_pairs[:2], iters, threads, orc)
    except Exception as e:   # scipy missing on the box: the figure is optional
        kd = {"error": repr(e)}
    return {"value": done / dt, "unit": "registrations/s", "cores": threads, "kind": "port", "kdtree": kd,
            "sample": f"{done} of the same scan pairs (bucketing + normals of BOTH clouds — the oracle has no source-only mode — + {iters} iterations each), "
                      f"oracle/m3d_oracle.c built with -O2 -fopenmp, {threads} threads"}


def cpu_kdtree_baseline(params, host_pairs, iters, threads, orc):
    """What pcl::IterativeClosestPoint with a point-to-plane estimator would do (the reference includes pcl/registration/icp.h but
    never calls it: m3d_calibration_sa.cpp:22): k-d tree on the target (scipy.spatial.cKDTree, all cores for the queries), 1-NN per
    source point with the max-distance reject, linearised point-to-plane Gauss-Newton in float64. The target normals are the
    oracle's (its bucketing is inside the timed region). Not bit-comparable with anything: a second CPU figure beside the port."""
    import numpy as np
    from scipy.spatial import cKDTree
    dmax = float(params.max_corr_dist[0])
    done, t0 = 0, time.perf_counter()
    for src, tgt in host_pairs:
        ct = orc.Cloud(params, tgt, omp=True)
        e = ct.export(0)
        q_xyz, q_nrm = e["sorted_xyz"].astype(np.float64), e["normals"].astype(np.float64)
        tree = cKDTree(q_xyz)
        p = src[np.isfinite(src).all(axis=1)].astype(np.float64)
        T = np.eye(4)
        for _ in range(iters):
            u = p @ T[:3, :3].T + T[:3, 3]
            d, j = tree.query(u, k=1, distance_upper_bound=dmax, workers=threads)
            ok = np.isfinite(d)
            n = q_nrm[j[ok]]
            ok2 = (n * n).sum(1) > 0.5
            uu, qq, nn = u[ok][ok2], q_xyz[j[ok]][ok2], n[ok2]
            r = ((uu - qq) * nn).sum(1)
            J = np.concatenate([np.cross(uu, nn), nn], axis=1)
            x = np.linalg.solve(J.T @ J, -J.T @ r)
            w, v = x[:3], x[3:]
            th = np.linalg.norm(w)
            K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            R = np.eye(3) + (np.sin(th) / th if th > 1e-12 else 1.0) * K + ((1 - np.cos(th)) / th ** 2 if th > 1e-12 else 0.5) * (K @ K)
            D = np.eye(4); D[:3, :3] = R; D[:3, 3] = v
            T = D @ T
        done += 1
        if time.perf_counter() - t0 > 15.0:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "registrations/s", "cores": threads,
            "sample": f"{done} of the same scan pairs, scipy cKDTree 1-NN ({threads} workers) + point-to-plane Gauss-Newton in numpy, {iters} iterations each"}


if __name__ == "__main__":
    main()
