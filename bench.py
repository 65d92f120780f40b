#!/usr/bin/env python3
"""bench.py — scan-pair registrations/s on synthetic Velodyne-HDL-32-shaped 100k-point clouds.

A "step" = one batch of --pairs-per-gpu independent scan-pair registrations per GPU (BASELINE config 4:
64 loop-closure pairs over 8 GPUs = 8 pairs per GPU; at N=1 the same 8-pair batch on one GPU). For every
pair the timed region covers the whole job on data already resident in HBM as PointCloud2 payloads:
decode + exact AABB of both clouds, the sort of the source (it is only ever a source: m3dreg_cloud_desc.source_only),
voxel bucketing + tile images + normals of the target, then --iters point-to-plane Gauss-Newton iterations (0.1 m voxel NN),
then the pose read-back; with N > 1 the poses of all ranks are gathered over RCCL (one all_gather per step — the only
collective; pairs never exchange data).

Prints ONE JSON line (rank 0). `roofline` prices the correspondence step of an iteration (k_nn_iter: source transform,
NN-certificate check, binning [the rare query that cannot be binned is walked by the reduction pass, or by k_nn_fallback on handles that saw many]; k_nn_tiles: the binned searches from LDS-staged target tiles) by its algorithmic
bytes (NN part: 12 N + 12 M + 8 C_occ per pair) over its average duration, taken from hipEvents the library
records on its stream around that step (m3dreg_profile_read): `frac` from inside the timed region, where four chains share
the GPU (every seventh iteration is bracketed: an event record is a barrier packet, bracketing all of them cost 4 % of the
throughput being measured), `alone` from one extra, untimed step with nothing else on the GPU (what a rocprofv3 kernel trace of
serial steps shows), `iteration` = a whole linearisation of the shipped schedule. From the 8th iteration of a level on the library
runs search and reduction as ONE launch (k_icp_late) wherever nobody asked for a bracket around the correspondence step; a
bracketed iteration runs as the two-launch chain (same bits),
so `roofline` always prices k_nn_iter (+ k_nn_tiles) — the fused launches show up in the step time, not in the bracket.
`cpu_baseline`: the port of the voxel algorithm (oracle/m3d_oracle.c) and a from-scratch k-d tree
ICP (oracle/m3d_kdtree_icp.c), each -O3 -march=native at 1 thread and at all cores, warm-up + median, on a bounded sample of
the same pairs. `legs`: the other BASELINE configurations and the §8d variants of the headline (serial steps, host payloads,
convergence-terminated), each measured by a child run of this script right after the headline (--no-extra skips them).
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
    ap.add_argument("--steps", type=int, default=100)   # (the driver passes --steps 20 --warmup 5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-gpu", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20, help="fixed Gauss-Newton iterations per registration")
    ap.add_argument("--azimuth", type=int, default=3125, help="azimuth steps per sweep (x32 beams = points)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--from-host", action="store_true",
                    help="hand over HOST PointCloud2 buffers every step (PCIe-inclusive rate; reported in DESIGN.md, never the headline)")
    ap.add_argument("--converge", action="store_true",
                    help="terminate on eps 1e-5 (max --iters) instead of running a fixed iteration count; secondary figure, see DESIGN.md")
    ap.add_argument("--inflight", type=int, default=4,
                    help="steps in flight: step i runs on handle/stream i %% D, the host enqueues step i+D-1 before waiting for step i (1 = strictly serial steps)")
    ap.add_argument("--queue-depth", type=int, default=2,
                    help="steps QUEUED per stream (the first of them runs, the others wait behind it on the same stream): with 2 a stream never "
                         "drains while the host collects a finished step and enqueues the next one")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real thing); gloo + --share-gpu = rehearsal of the N > 1 flow on a single-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--event-every", type=int, default=13, help="bracket every n-th iteration with hipEvents (n should not divide --iters: every iteration index is then sampled over the steps). "
                    "Round 6: 13 (rounds 1-5 and profiles/r06_v1/v2: 7) — an event record is a barrier packet, the brackets of every 7th iteration + four per-batch ones cost the "
                    "headline 5 %% of what it measures without any (profiles/r06_event_density.txt)")
    ap.add_argument("--batch-brackets-every", type=int, default=4, help="the per-batch brackets (bucketing, whole chain of iterations) on every n-th batch of a handle only (m3dreg_profile_batches)")
    ap.add_argument("--no-events", action="store_true", help="do not record hipEvents in the timed region (A/B of their cost; roofline then reads 0)")
    ap.add_argument("--trace-host", action="store_true", help="print the host-side time of every enqueue (bucketing / launch) and wait of the timed region to stderr")
    ap.add_argument("--pair-offset", type=int, default=None,
                    help="index of this rank's first scan pair (default rank * pairs-per-gpu): lets one GPU run the shard another rank gets at N > 1")
    ap.add_argument("--pair-list", default=None, help="comma-separated indices of this rank's scan pairs (diagnostics: run any shard of any N on one GPU)")
    ap.add_argument("--shard", default="lpt", choices=["lpt", "consecutive"],
                    help="N > 1: which pairs a rank gets. lpt = the world_size * pairs-per-gpu pairs dealt longest-processing-time-first by an "
                         "a-priori cost estimate (synth.crowdedness of both clouds), the same number to every rank (SURVEY 8d config 4: "
                         "LPT-sharded); consecutive = rank r takes pairs r * B ... r * B + B - 1")
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the cpu_baseline 'all cores' legs (0 = every core of the host)")
    ap.add_argument("--workload", default="config4", choices=["config4", "config3", "config2", "config5", "loop64"],
                    help="config4 (default, the headline): B loop-closure pairs per GPU per step; config3: the single 100k pair, point-to-plane, leaf 0.1; "
                         "config2: the single 70k pair, point-to-point, leaf 0.2, eps 1e-5 / 30 iterations; config5: 100k live scan against the 2 M-point map, "
                         "leaves 0.4 / 0.2 / 0.1 (own code path: run_config5)")
    ap.add_argument("--multi-devices", default=None, help="single-process multi-device mode (m3dreg_multi_*: ONE process drives the listed devices through the C ABI, host "
                                                         "payloads in, a different measurement from the contract's line): comma-separated device ordinals, e.g. 0,0 to rehearse "
                                                         "two contexts on one GPU. Never implied: --gpus N > 1 without torchrun starts N rank processes instead")
    ap.add_argument("--pageable", action="store_true", help="--multi-devices / --from-host: hand over PAGEABLE host payloads (default: pinned, m3dreg_host_alloc)")
    ap.add_argument("--spawn", action="store_true", help="start the rank processes from this one even for --gpus 1 (what --gpus N > 1 does when torchrun did not): checks that "
                                                         "the launcher adds nothing to the measurement")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="--gpus N without torchrun: seconds after which the launcher ends its rank processes")
    ap.add_argument("--async-calls", action="store_true", help="serial steps (--inflight 1 --queue-depth 1) through m3dreg_align_batch_async + m3dreg_batch_wait instead of the synchronous "
                                                               "m3dreg_align_batch: one launch chain per step, no internal chains (the figure rounds 1-4 called `serial`)")
    ap.add_argument("--no-latency-mode", action="store_true", help="serial synchronous steps WITHOUT m3dreg_set_latency_mode (the serial caller's statement that its batches have the GPU to themselves, ABI 7)")
    ap.add_argument("--no-extra", action="store_true", help="headline only: do not run the other configurations / variants as child runs")
    ap.add_argument("--all-legs", action="store_true", help="also the legs the default run leaves out to stay short (batch16 / batch32 / batch64 with two chains, from_host_converge)")
    ap.add_argument("--verify-steps", action="store_true",
                    help="compare EVERY step's poses and statistics (timed steps included) byte for byte with the first result of the same shard; the default run "
                         "verifies --verify-extra untimed steps of the same schedule right after the timed region instead")
    ap.add_argument("--verify-extra", type=int, default=None, help="untimed steps of the same schedule whose results are checked byte for byte after the timed region (default: 64 in the "
                                                                     "full run — the driver's command —, 0 in the child legs and with --no-extra / --no-cpu-baseline: scripts that count launches per step)")
    ap.add_argument("--rotate-pairs", action="store_true",
                    help="step k registers shard k mod 8 of the 64 config-4 pairs (the eight LPT shards bench.py --gpus 8 forms) instead of the same 8 pairs every step: "
                         "64 pairs' payloads resident, inputs no longer repeat from step to step")
    ap.add_argument("--min-seconds", type=float, default=1.5,
                    help="the block of --steps timed steps is repeated (each block bracketed by barrier + synchronize on both sides, exactly --steps steps) "
                         "until the blocks add up to this much wall time; the MEDIAN block is reported (0 = one block)")
    ap.add_argument("--max-blocks", type=int, default=200)
    a = ap.parse_args()
    if a.verify_extra is None:
        a.verify_extra = 64 if (a.workload == "config4" and not a.no_extra and not a.no_cpu_baseline) else 0
    return a


def run_multi(args):
    """`python bench.py --multi-devices 0,1,...` (explicit only): ONE process drives the listed devices through the C ABI's m3dreg_multi_*
    (include/m3dreg.h) — what a single gpu_6dslam_node on a multi-GPU host would do. A step = N x pairs-per-gpu pairs handed over as HOST
    PointCloud2 payloads: LPT assignment in C++, upload, bucketing and registration on every device, poses gathered in pair order, all
    inside the timed call — a PCIe-inclusive figure by construction (the contract's N > 1 line is the torchrun path above; this is the
    deployment the library offers beside it). --multi-devices 0,0 rehearses it on one GPU."""
    from mandala_mapping_amd import abi, binding, synth
    devices = [int(x) for x in args.multi_devices.split(",")]
    params = abi.Params.make(leaf=0.1, iterations=args.iters, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    n = args.pairs_per_gpu * len(devices)
    pairs, gts = [], []
    for k in range(n):
        src, tgt, Tgt = synth.config4_pair(k, args.azimuth)
        pairs.append((src, tgt, None)); gts.append(Tgt)
    M = binding.MultiRegistrar(params, devices=devices)
    descs, keep = M.describe(pairs, source_only=True, pinned=not args.pageable)   # pinned host payloads: asynchronous DMA; --pageable: staged through the device threads' pinned blocks
    for _ in range(max(1, args.warmup)):
        T, st, dev = M.align_described(descs)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        T, st, dev = M.align_described(descs)
    dt = time.perf_counter() - t0
    errs = [synth.pose_error(T[i], gts[i]) for i in range(n)]
    emit({
        "metric": "scan-pair registrations/sec (100k-pt clouds, point-to-plane, 0.1 m voxel NN)", "value": n * args.steps / dt, "unit": "registrations/s",
        "n_gpus": len(devices), "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32 (int64 fixed-point sums, f64 solve)", "data": "synthetic",
        "config": {"workload": f"BASELINE config 4: {args.pairs_per_gpu} HDL-32-shaped scan pairs per device per step, ONE process, m3dreg_multi_align over devices {devices} "
                               "(host PointCloud2 payloads in, poses out: upload, LPT assignment, bucketing and 20 iterations inside the timed call, strictly serial steps)",
                   "pairs_per_gpu": args.pairs_per_gpu, "iterations": args.iters, "parallelism": f"pairs LPT-sharded over {len(devices)} device context(s) by the library, no collective"},
        "pairs_per_device_ordinal": {str(d): int((dev == d).sum()) for d in sorted(set(devices))},
        "max_rot_err_deg": max(e[0] for e in errs), "max_trans_err_m": max(e[1] for e in errs),
        "roofline": None, "cpu_baseline": None})


def launch_ranks(args):
    """`python bench.py --gpus N` WITHOUT torchrun: this process — which has not touched the GPU and never will (it imports neither torch
    nor the library) — starts N fresh rank processes of this same script, one per device, with the environment torchrun would give them
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), so that every rank runs EXACTLY the per-rank code of the N = 1 line:
    device-resident payloads, the same steps in flight, rendezvous over RCCL, max-over-ranks timing, rank 0 prints the one JSON line.
    Children are started (subprocess), never exec'ed into. All of them are polled together: the first rank that exits with an error ends
    the others (they would otherwise sit in the rendezvous / a barrier until the collective's timeout) and its code is this run's; an
    overall --launch-timeout does the same for a run that hangs."""
    import socket
    import subprocess
    # the port stays bound (SO_REUSEADDR, never listening) until the ranks have been started: nobody else is handed it by bind(0) in between
    so = socket.socket()
    so.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    procs = []
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), M3D_BENCH_RANK_PROCESS="1")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))   # stdout / stderr inherited: rank 0's line is this run's line
    finally:
        so.close()
    rc, deadline = 0, time.monotonic() + args.launch_timeout
    live = list(procs)
    while live and rc == 0:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0:
                rc = abs(r) or 1
                print(f"[bench] rank process {procs.index(p)} exited with code {r}: ending the other ranks", file=sys.stderr)
                break
        if rc == 0 and live:
            if time.monotonic() > deadline:
                rc = 124
                print(f"[bench] --launch-timeout {args.launch_timeout:.0f} s reached: ending the rank processes", file=sys.stderr)
            else:
                time.sleep(0.05)
    if rc:   # the ranks this launcher started (exact PIDs), first politely
        for p in live:
            p.terminate()
        t_end = time.monotonic() + 10.0
        for p in live:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        raise SystemExit(rc)


def main():
    args = parse()
    if args.workload == "config5":
        return run_config5(args)
    if args.workload == "loop64":
        return run_loop64(args)
    if args.multi_devices:
        return run_multi(args)
    if (args.gpus > 1 or args.spawn) and "M3D_BENCH_RANK_PROCESS" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return launch_ranks(args)
    if os.environ.get("M3D_BENCH_DRYRUN"):   # tests/test_bench_launcher.py: what a rank process is started with (no GPU, no torch)
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")} | {"argv": sys.argv[1:]}), flush=True)
        fail_rank = os.environ.get("M3D_BENCH_DRYRUN_FAIL_RANK")   # (the launcher's failure path: that rank dies, the others "hang in the rendezvous")
        if fail_rank is not None:
            if os.environ.get("RANK") == fail_rank:
                raise SystemExit(3)
            time.sleep(300)
        return
    pregen = {}
    if args.rotate_pairs:
        if args.workload != "config4" or int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.pairs_per_gpu != 8 or args.pair_list is not None or args.from_host:
            raise SystemExit("--rotate-pairs: the N = 1 config-4 headline with 8 pairs per step and device-resident payloads only")
        import multiprocessing as mp
        from mandala_mapping_amd import synth as synth_
        with mp.get_context("fork").Pool(min(host_cores(), 16)) as pool:
            got = pool.starmap(synth_.config4_pair, [(i, args.azimuth) for i in range(64)])
        pregen = dict(enumerate(got))
    elif args.workload == "config4" and int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.pairs_per_gpu >= 16 and args.pair_list is None:
        # a big batch's clouds are ray-cast by a pool of forked workers BEFORE this process touches the GPU (64 pairs: 20 s of numpy on one core)
        import multiprocessing as mp
        from mandala_mapping_amd import synth as synth_
        first = args.pair_offset or 0
        with mp.get_context("fork").Pool(min(host_cores(), 16, args.pairs_per_gpu)) as pool:
            got = pool.starmap(synth_.config4_pair, [(first + i, args.azimuth) for i in range(args.pairs_per_gpu)])
        pregen = {first + i: g for i, g in enumerate(got)}
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
    collectives = "none (one rank)"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        collectives = args.dist_backend
        if args.dist_backend == "nccl":
            # RCCL for the run's three control-plane collectives (barrier, max of the ranks' times, the pose / per-rank gather): nothing of a step's data path is a
            # collective. Should the RCCL group not come up on this node (it has never run with two ranks: no multi-GPU box in six rounds), the SAME
            # three collectives go through gloo on host tensors instead of the run ending without its line; every rank takes the same decision (the eager
            # communicator set-up below fails or succeeds for the group), the line says which backend carried them.
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                assert int(probe.item()) == world
            except Exception as e:   # noqa: BLE001
                print(f"[bench] rank {rank}: the RCCL process group failed ({e!r}); the barrier / max / gather go through gloo (no collective is on the data path)", file=sys.stderr, flush=True)
                try:
                    dist.destroy_process_group()
                except Exception:   # noqa: BLE001
                    pass
                dist.init_process_group("gloo", rank=rank, world_size=world)   # (the same rendezvous store — under torchrun the agent hosts it —, a fresh key prefix)
                cdev, collectives = None, "gloo (RCCL group failed)"
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    if args.workload in ("config2", "config3"):   # one pair per step (the latency of ONE registration is the point)
        args.pairs_per_gpu = 1
    B, K, W = args.pairs_per_gpu, args.steps, args.warmup
    # fixed iteration count: eps = 0 never triggers, so every launch of the dominant kernel does full work
    if args.workload == "config2":   # 70 016 rays, point-to-point, leaf 0.2 m, d_max 1.0 m, eps 1e-5 or 30 iterations
        # coarse to fine FROM IDENTITY (the pair is 0.5 m / 3 deg apart): 0.8 / 0.4 / 0.2 m, the last level with a correspondence distance below the ring
        # spacing — what point-to-point needs on ring-structured sweeps (tests/test_gpu_parity.py: CONFIG2, 0.076 deg / 1.0 cm)
        args.iters, args.converge = 150, True
        params = abi.Params.make(leaf=(0.8, 0.4, 0.2), iterations=(30, 30, 150), max_corr_dist=(2.0, 0.6, 0.2), metric=abi.POINT_TO_POINT, eps_rot=1e-5, eps_trans=1e-5)
        gen_pair = lambda k: synth.config2()
        init_of = lambda Tgt: None   # identity
    else:
        params = abi.Params.make(leaf=0.1, iterations=args.iters, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE,
                                 normal_leaf=0.4, eps_rot=1e-5 if args.converge else 0.0, eps_trans=1e-5 if args.converge else 0.0)
        gen_pair = (lambda k: synth.config3()) if args.workload == "config3" else (lambda k: pregen.pop(k) if k in pregen else synth.config4_pair(k, args.azimuth))
        init_of = lambda Tgt: None   # identity
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
    payloads, gts, host_pairs, host_msgs, inits = [], [], [], [], []
    if args.pair_list is not None:
        pair_ids, generated = [int(x) for x in args.pair_list.split(",")], {}
        assert len(pair_ids) == B, "--pair-list must name --pairs-per-gpu pairs"
    elif args.pair_offset is not None:
        pair_ids, generated = [args.pair_offset + i for i in range(B)], {}
    elif world > 1 and args.shard == "lpt":
        # every rank derives the same assignment from the same a-priori costs: no communication, deterministic. The costs of the 64 standard pairs are
        # tabulated (mandala_mapping_amd/config4_costs.json; ray-casting all of them takes half a minute per rank) — anything else is computed here
        costs, generated = None, {}
        tab = os.path.join(ROOT, "mandala_mapping_amd", "config4_costs.json")
        if args.workload == "config4" and os.path.exists(tab):
            t_ = json.load(open(tab))
            if t_.get("azimuth") == args.azimuth and len(t_["costs"]) >= world * B:
                costs, _ = sharding.table_costs(t_, world * B)
        if costs is None:   # any other workload: the cost estimate comes from the DEVICE (m3dreg_cloud_density: the bucketing pipeline's own sum of squared
            # voxel populations) — every rank buckets all world * B pairs once, untimed, and reads the same numbers
            generated = {k: gen_pair(k) for k in range(world * B)}
            costs = []
            for k in range(world * B):
                cs_, ct_ = reg.clouds([generated[k][0], generated[k][1]], source_only=[True, False])
                costs.append(cs_.density() + ct_.density())
                cs_.free(); ct_.free()
        pair_ids = sharding.lpt_assign(costs, world, capacity=B)[rank]
        generated = {k: generated[k] for k in pair_ids if k in generated}
    else:
        pair_ids, generated = [rank * B + i for i in range(B)], {}
    # shards[s] = the pairs step k registers when k % len(shards) == s: ONE shard (this rank's pairs) unless --rotate-pairs, which cycles through the
    # eight LPT shards `bench.py --gpus 8` forms out of config 4's 64 pairs
    shard_ids = [list(pair_ids)]
    if args.rotate_pairs:
        t_ = json.load(open(os.path.join(ROOT, "mandala_mapping_amd", "config4_costs.json")))
        if t_.get("azimuth") == args.azimuth:
            c64, shard_source = sharding.table_costs(t_, 64)
            shard_ids = sharding.lpt_assign(c64, 8, capacity=8)
        else:
            shard_ids, shard_source = [list(range(8 * s_, 8 * s_ + 8)) for s_ in range(8)], "consecutive (no cost table for this azimuth)"
        pair_ids = shard_ids[0]
    shard_payloads, shard_gts = [], []
    for si, ids in enumerate(shard_ids + ([list(range(B))] if args.rotate_pairs else [])):   # (--rotate-pairs: pairs 0 .. 7, the N = 1 line's own workload, as a ninth list: timed beside the shards, never part of the rotation)
        pl_, gt_ = [], []
        for i in range(B):
            src, tgt, Tgt = generated[ids[i]] if ids[i] in generated else gen_pair(ids[i])
            ms, mt = encode_xyz(src), encode_xyz(tgt)
            ds = torch.frombuffer(bytearray(ms.data), dtype=torch.uint8).to(dev)
            dt = torch.frombuffer(bytearray(mt.data), dtype=torch.uint8).to(dev)
            pl_.append((ds, ms.n, dt, mt.n))
            gt_.append(Tgt)
            if si == 0:
                host_msgs += [ms, mt]
                inits.append(init_of(Tgt))
                if rank == 0 and i < 8:
                    host_pairs.append((src, tgt))
        shard_payloads.append(pl_); shard_gts.append(gt_)
    payloads, gts = shard_payloads[0], shard_gts[0]
    torch.cuda.synchronize()
    if args.from_host and not args.pageable:   # the payloads a producer hands over live in pinned memory: the copies are asynchronous DMA, not staged
        pinned_views = [np.frombuffer(m.data, np.uint8) for m in host_msgs]
        for v in pinned_views:
            if binding.lib().m3dreg_host_register(C.c_void_p(v.ctypes.data), C.c_size_t(v.nbytes)) != 0:
                raise SystemExit("m3dreg_host_register failed")

    last = {}

    n_rot = len(shard_ids)
    shard_items = []
    for pl_ in shard_payloads:
        items = []
        for ds, ns, dt, nt in pl_:
            items += [(ds.data_ptr(), ns), (dt.data_ptr(), nt)]
        shard_items.append(items)

    def make_clouds(r, shard=0):
        """decode + AABB + bucketing + normals of this rank's 2B clouds, one batched pipeline on r's stream"""
        if args.from_host:
            cl = r.clouds(host_msgs, wait=False, source_only=[True, False] * B)   # host buffers (they outlive the step) cross PCIe inside the timed region
        else:
            cl = r.clouds_from_device(shard_items[shard], wait=False, source_only=[True, False] * B)   # no host synchronisation anywhere in a step's chain; sources: no normals
        return [(cl[2 * i], cl[2 * i + 1]) for i in range(B)]

    # --verify-steps / the untimed verification leg: every step's poses + statistics against the FIRST result of the same shard, byte for byte
    verify = {"on": bool(args.verify_steps), "ref": {}, "checked": 0, "mismatches": 0, "first_bad": None}

    def check_step(idx, shard, T, st):
        sig = np.asarray(T, np.float64).tobytes() + b"".join(bytes(x) for x in st)
        ref = verify["ref"].setdefault(shard, sig)
        verify["checked"] += 1
        if sig != ref:
            verify["mismatches"] += 1
            if verify["first_bad"] is None:
                verify["first_bad"] = {"step": int(idx), "shard": int(shard)}

    def finish(T, st, clouds, idx=0, shard=0):
        if verify["on"]:
            check_step(idx, shard, T, st)
        if world > 1:   # the only collective of the path: one all_gather of poses + status per step (RCCL over xGMI).
            # It is enqueued here and read one step later (or at the end of the timed region): the ranks exchange every
            # step's results without falling into lock-step at every step.
            ticket = sharding.gather_results_start(pair_ids, T, [x.status for x in st], world * B, dist, cdev)
            drain_gather()
            last["pending_gather"] = ticket
        last["T"], last["st"], last["shard"] = T, st, shard

    def drain_gather():
        if last.get("pending_gather") is not None:
            last["all"] = sharding.gather_results_finish(last.pop("pending_gather"))

    host_log = []
    # one step at a time (--inflight 1 --queue-depth 1): the caller is serial — it makes the synchronous call, like the ROS node
    serial_calls = len(regs) == 1 and not args.async_calls
    sync_result = [None]
    if serial_calls and not args.no_latency_mode:   # what a serial caller — the ROS node — tells the library (include/m3dreg.h, ABI 7); never in the headline: its handles share the GPU
        regs[0].set_latency_mode(True)

    force_shard = [None]   # per-shard timing of --rotate-pairs: every step registers this shard

    def shard_of(i):
        return force_shard[0] if force_shard[0] is not None else i % n_rot

    def enqueue(i):
        r = regs[i % len(regs)]
        ta = time.perf_counter()
        clouds = make_clouds(bregs[i % len(bregs)], shard_of(i))
        tb = time.perf_counter()
        arr = r._pairs([(s_, t_, inits[j]) for j, (s_, t_) in enumerate(clouds)])
        if serial_calls:   # a serial caller's call: the synchronous m3dreg_align_batch 
            sync_result[0] = r.align_batch_arr(arr, B)
        else:
            r.align_batch_async(arr, B)
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
            T, st = sync_result[0] if serial_calls else regs[idx % len(regs)].batch_wait(B)
            host_log.append(("wait", idx, 1e3 * (time.perf_counter() - ta), 0.0))
            finish(T, st, clouds, idx, shard_of(idx))
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
        r.profile_batches(args.batch_brackets_every)
        r.profile_enable(not args.no_events and not serial_calls, every=args.event_every)   # 13 (7) does not divide the 20 iterations of a step: every iteration index gets sampled (synchronous calls: no brackets — a bracketed batch runs as ONE chain; `alone` is measured below)
        r.profile_read(0, reset=True)
        r.profile_read(1, reset=True)
    import gc
    gc.collect()
    gc.disable()   # the host thread only enqueues and collects; a generation-2 collection in the middle of the region is a 40 ms stall (measured: -9 % at 300 steps)
    # The block of exactly K timed steps (barrier + synchronize on both sides) is repeated until the blocks add up to --min-seconds: 20 steps
    # are 23 ms, too short for anything outside this process (the driver's smi sampler) to see a busy GPU, and one block's value moves by a
    # few per cent with whatever else the box does in those milliseconds. Reported: the MEDIAN block (steps = K as asked).
    blocks, blocks_local = [], []
    for r in regs:
        r.profile_read(4, reset=True)
    while True:
        barrier()
        host_log.clear()
        t0 = time.perf_counter()
        run_steps(K)
        torch.cuda.synchronize()
        blocks_local.append(time.perf_counter() - t0)   # this rank's OWN work: until its K steps are done, before it waits for the others in the closing barrier
        barrier()
        t1 = time.perf_counter()
        bt = t1 - t0
        if world > 1:   # the same block list on every rank (max over ranks), so every rank takes the same decision to go on
            tt = torch.tensor([bt], dtype=torch.float64, device=cdev if cdev is not None else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            bt = float(tt.item())
        blocks.append(bt)
        if sum(blocks) >= args.min_seconds or len(blocks) >= args.max_blocks:
            break
    gc.enable()
    if args.trace_host and rank == 0:
        for what, i, a, b in host_log:
            print(f"[host] {what} step {i}: {a:.3f} ms" + (f" bucketing, {b:.3f} ms enqueue of the iterations" if what == "enq" else ""), file=sys.stderr)
    launches = kern_ms = iters_timed = iter_ms = buck_n = buck_ms = chain_n = chain_ms = 0
    for r in regs:
        a, b = r.profile_read(1, reset=True)       # the correspondence step (k_nn_iter + k_nn_tiles) alone
        c, d = r.profile_read(0, reset=True)       # correspondence step + reduction + solve of one linearisation
        e_, f_ = r.profile_read(2, reset=True)     # one bucketing batch
        g_, h_ = r.profile_read(4, reset=True)     # ALL iterations of a batch as they ship (fused late launches included): iterations, ms
        launches, kern_ms, iters_timed, iter_ms, buck_n, buck_ms = launches + a, kern_ms + b, iters_timed + c, iter_ms + d, buck_n + e_, buck_ms + f_
        chain_n, chain_ms = chain_n + g_, chain_ms + h_
    # ---- untimed: the SAME schedule (handles, streams, queue depth, event brackets, recycled blocks), every step's 8 poses + statistics compared byte
    # for byte with the first result of its shard. The timed region keeps only its last step's poses (checked against ground truth below); this
    # leg is what says that the pipelined schedule returns the same bits step after step (tests/test_gpu_pipelined.py holds it against the single-handle result).
    if args.verify_extra > 0:
        was = verify["on"]
        verify["on"] = True
        run_steps(args.verify_extra)
        torch.cuda.synchronize()
        verify["on"] = was
    if verify["mismatches"]:
        print(f"[bench] VERIFY FAILED: {verify['mismatches']} of {verify['checked']} steps returned other bits than the first step of their shard (first: {verify['first_bad']})", file=sys.stderr, flush=True)
        raise SystemExit(3)
    scale_ceiling = None
    if args.rotate_pairs:
        # every LPT shard of the N = 8 job as the whole workload of this one GPU, same schedule, back to back on this box; and pairs 0 .. 7, what the N = 1 line runs.
        # At N = 8 a step of the job lasts as long as its slowest shard's: value(8) <= 64 / max_s t_s, value(1) = 8 / t_1.
        per = []
        for sh in range(len(shard_items)):
            force_shard[0] = sh
            run_steps(len(regs))
            torch.cuda.synchronize()
            ts_ = []
            for _ in range(3):
                t0 = time.perf_counter(); run_steps(K); torch.cuda.synchronize(); ts_.append((time.perf_counter() - t0) / K)
            per.append(1e3 * sorted(ts_)[1])
        force_shard[0] = None
        t_sh, t_1 = per[:n_rot], per[n_rot]
        cm = [sum(c64[k] for k in ids) for ids in shard_ids] if t_.get("azimuth") == args.azimuth else None
        scale_ceiling = {"status": "UNMEASURED on 8 GPUs: one GPU ran every shard in turn", "per_shard_ms_per_step": t_sh, "n1_workload_ms_per_step": t_1,
                         "n8_over_n1_measured_shards": 8.0 * t_1 / max(t_sh), "balance_mean_over_max": sum(t_sh) / len(t_sh) / max(t_sh),
                         "n8_over_n1_cost_model": (8.0 * (sum(cm) / len(cm)) / max(cm)) if cm else None,
                         "shards": [[int(k) for k in ids] for ids in shard_ids], "shard_source": shard_source}
    for r in regs:
        for w_ in (0, 1, 2, 4):
            r.profile_read(w_, reset=True)
        r.profile_enable(False)
    # After the timed region, untimed: the same kernel with NOTHING else on the GPU (one step, one handle, every launch bracketed).
    # With several chains sharing the GPU a launch takes longer although more launches complete per second; this is the kernel's own
    # duration, reported beside the contract's figure as roofline.alone.
    alone_ms = alone_iter_ms = alone_bucket_ms = alone_chain_ms = 0.0
    gather_model = None
    if not args.no_events and world == 1:
        last_T, last_st = last.get("T"), last.get("st")
        # (this step HAS the GPU to itself, and says so — m3dreg_set_latency_mode: since round 6 the statement also selects the one-XCD-per-pair map that a lone batch wants;
        #  the timed region's handles never make it)
        alone_stated = not serial_calls and not args.no_latency_mode and not os.environ.get("M3D_BENCH_ALONE_DEFAULT")   # (the variable: A/B of the two maps on the lone step)
        if alone_stated:
            regs[0].set_latency_mode(True)
        regs[0].profile_batches(1)   # (the per-batch brackets of EVERY batch again: this leg is a handful of batches)
        regs[0].profile_enable(True, every=1)
        for w_ in range(4):
            regs[0].profile_read(w_, reset=True)
        c_ = make_clouds(regs[0])
        for s_, t_ in c_:   # (a lone caller's clouds come out of the synchronous creation calls, which read their counts back: the dense-level schedule is then decided on the host —
            s_.status(); t_.status()   # no empty k_nn_coop launch behind the bracketed late iterations, which the enqueue-only clouds of the timed region get: include/m3dreg.h "Threading")
        regs[0].align_batch_async(regs[0]._pairs([(s_, t_, inits[j]) for j, (s_, t_) in enumerate(c_)]), B)
        regs[0].batch_wait(B)
        a_, b_ = regs[0].profile_read(1, reset=True)
        ai_, bi_ = regs[0].profile_read(0, reset=True)
        ab_, bb_ = regs[0].profile_read(2, reset=True)
        regs[0].profile_enable(False)
        alone_ms = b_ / max(1, a_)
        alone_iter_ms, alone_bucket_ms = bi_ / max(1, ai_), bb_ / max(1, ab_)
        # the shipped schedule alone (no per-iteration brackets: the late iterations run fused): one more untimed step, one bracket around its chain
        regs[0].profile_enable(True, every=1 << 30)
        regs[0].profile_read(4, reset=True)
        regs[0].align_batch_async(regs[0]._pairs([(s_, t_, inits[j]) for j, (s_, t_) in enumerate(c_)]), B)
        regs[0].batch_wait(B)
        ac_, bc_ = regs[0].profile_read(4, reset=True)
        regs[0].profile_enable(False)
        if alone_stated:
            regs[0].set_latency_mode(False)
        alone_chain_ms = bc_ / max(1, ac_)
        # SURVEY 8d "gather-model bytes": 12 N k + 8 * 27 N per pair, k = mean number of candidates the spec names per query (all points of its 27
        # voxels; measured at the initial and at the final pose — the searches prune most of them, the certificates skip most searches)
        if host_pairs and args.workload != "config5":
            try:
                gm = {}
                T_end = last_T[0] if last_T is not None else np.eye(4)
                T_ini = inits[0] if inits[0] is not None else np.eye(4)
                tcl = regs[0].cloud(host_pairs[0][1])
                nlv = tcl.levels() if hasattr(tcl, "levels") else 1
                for nm, Tq in (("initial_pose", T_ini), ("final_pose", T_end)):
                    q = synth.apply_T(Tq, host_pairs[0][0]).astype(np.float32)
                    kbar = float(np.mean(tcl.candidates(q, level=nlv - 1)))
                    gm[nm] = {"mean_candidates_per_query": kbar, "bytes_per_launch": int(sum((12.0 * kbar + 8 * 27) * s_.n for s_, _ in c_))}
                gather_model = gm
                del tcl
            except Exception as ex:   # diagnostics only: never fail the bench line over them
                gather_model = {"error": repr(ex)}
        del c_
        last["T"], last["st"] = last_T, last_st   # (last["shard"] still names the shard of that step)
    elapsed = sorted(blocks)[len(blocks) // 2]   # the median block (already the max over ranks)
    per_rank = None
    if world > 1:   # every rank's own block times beside the max-over-ranks figure: the loss to the slowest shard is visible in the line itself
        mine = {"rank": rank, "pairs": [int(x) for x in pair_ids], "own_work_ms_median": 1e3 * sorted(blocks_local)[len(blocks_local) // 2],
                "own_work_ms_min": 1e3 * min(blocks_local), "own_work_ms_max": 1e3 * max(blocks_local)}
        try:   # (diagnostics: a failure here must not cost the run its line. A fixed-size float64 tensor through the SAME collective the timed region used every step —
            # all_gather_object pickles through a second code path (byte tensors, size exchange) that has never run on this fabric)
            rec = torch.full((4 + 16,), -1.0, dtype=torch.float64)
            rec[0], rec[1], rec[2], rec[3] = rank, mine["own_work_ms_median"], mine["own_work_ms_min"], mine["own_work_ms_max"]
            for q_, pid in enumerate(mine["pairs"][:16]):
                rec[4 + q_] = pid
            rec = rec.to(cdev) if cdev is not None else rec
            allr = torch.empty((world * rec.numel(),), dtype=torch.float64, device=rec.device)
            dist.all_gather_into_tensor(allr, rec)
            allr = allr.cpu().numpy().reshape(world, -1)
            per_rank = [{"rank": int(a[0]), "pairs": [int(x) for x in a[4:] if x >= 0], "own_work_ms_median": float(a[1]), "own_work_ms_min": float(a[2]), "own_work_ms_max": float(a[3])} for a in allr]
        except Exception as ex:
            per_rank = [dict(mine, error="per-rank gather failed: " + repr(ex)[:120])]

    # sanity of the timed work: poses against the generator's ground truth
    errs = [synth.pose_error(last["T"][i], shard_gts[last.get("shard", 0)][i]) for i in range(B)]
    max_rot, max_tr = max(e[0] for e in errs), max(e[1] for e in errs)

    if rank == 0:
        total_regs = world * B * K
        value = total_regs / elapsed
        avg_launch_s = (kern_ms / 1e3) / max(1, launches)
        achieved = alg_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        traffic = l2_hit = traffic_unc = None
        traffic_source = None
        t_ph = time.perf_counter()
        live = measure_traffic(args) if (world == 1 and not args.no_extra and not args.no_cpu_baseline and args.workload == "config4") else None
        print(f"[bench] traffic passes: {time.perf_counter() - t_ph:.1f} s of wall clock", file=sys.stderr, flush=True)
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if live is not None:
            traffic, traffic_unc = live["correspondence_step_bytes_per_launch"], live["correspondence_step_bytes_per_launch_fetch_uncorrected"]
            traffic_source = ("measured in this run: two child runs of this script under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, serial steps); `traffic` = FETCH_SIZE x 2 + "
                              "WRITE_SIZE (the guide's gfx950 correction, calibrated for wide coalesced streaming reads), `traffic_fetch_uncorrected` = FETCH_SIZE + WRITE_SIZE: these kernels "
                              "mix 16-B streams with 16-B gathers, for which the guide gives no calibration — the true figure lies between the two")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if live is None:
                    traffic = sum(j.get(k, {}).get("hbm_bytes_per_iteration", 0.0) for k in ("k_nn_iter", "k_nn_tiles")) or None
                    traffic_unc = sum(j.get(k, {}).get("hbm_bytes_per_iteration_uncorrected", 0.0) for k in ("k_nn_iter", "k_nn_tiles")) or None
                    traffic_source = "committed: profiles/pmc_summary.json (rocprofv3 --pmc passes of scripts/profile_gpu.sh on the builder's box, NOT measured in this run: rocprofv3 was not usable here)"
                l2_hit = {k: j[k]["l2_hit_rate"] for k in ("k_nn_iter", "k_nn_tiles", "k_accumulate_matches", "k_icp_late") if k in j and "l2_hit_rate" in j[k]} or None
            except Exception:
                if live is None:
                    traffic = traffic_unc = None
                l2_hit = None
        in_region = achieved
        alone = alg_bytes / (alone_ms / 1e3) / 1e9 if alone_ms > 0 else 0.0
        # one whole linearisation of the SHIPPED schedule (correspondence step + residuals + reduction + solve; fused k_icp_late launches included):
        # mean over every iteration of every batch of the timed region, one event bracket per batch (M3DREG_PROFILE_CHAIN)
        chain_iter_ms = chain_ms / max(1, chain_n)
        iteration_gbps = alg_bytes_iter / (chain_iter_ms / 1e3) / 1e9 if chain_iter_ms > 0 else 0.0
        alone_iteration_gbps = alg_bytes_iter / (alone_chain_ms / 1e3) / 1e9 if alone_chain_ms > 0 else 0.0
        workload = {"config4": f"BASELINE config 4 shard: {B} HDL-32-shaped scan pairs per GPU per step, ",
                    "config3": "BASELINE config 3: the single HDL-32-shaped scan pair (seeds 100 / 101), one registration per step, ",
                    "config2": "BASELINE config 2: the single 70 016-ray HDL-32-shaped scan pair (seeds 100 / 101), one registration per step from identity, "}[args.workload]
        out = {
            "metric": "scan-pair registrations/sec (100k-pt clouds, point-to-plane, 0.1 m voxel NN)",
            "value": value, "unit": "registrations/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (int64 fixed-point sums, f64 solve)", "data": "synthetic",
            "blocks": {"n": len(blocks), "steps_per_block": K, "median_ms": 1e3 * elapsed, "min_ms": 1e3 * min(blocks), "max_ms": 1e3 * max(blocks),
                       "note": "the block of `steps` timed steps is repeated until the blocks add up to --min-seconds; value / ms_per_step are the MEDIAN block's"},
            "config": {"workload": workload +
                                   f"{n_pts} pts/cloud (as m3d_aggregator publishes the sweeps: its +-1 m self-filter box applied), " +
                                   ("point-to-point, leaves 0.8 / 0.4 / 0.2 m, eps 1e-5 / at most 30 + 30 + 150 iterations, " if args.workload == "config2" else
                                    f"point-to-plane, leaf 0.1 m, {args.iters} " + ("iterations at most (eps 1e-5), " if args.converge else "fixed iterations, ")) +
                                   "decode of both clouds, sort of the source (m3dreg_cloud_desc.source_only), bucketing + tile images + normals of the target inside the timed region",
                       "payload": "host (pinned PointCloud2 buffers cross PCIe inside the timed region)" if args.from_host else "hbm-resident",
                       "pipelining": ("none: one synchronous call per step" if serial_calls else f"caller-side: {len(regs)} handles / {D} streams" + (", the same 8 pairs every step" if n_rot == 1 else f", step k registers shard k mod {n_rot}")) +
                                     ("; no event brackets" if (args.no_events or serial_calls) else f"; brackets: iteration {args.event_every}, batch {args.batch_brackets_every}"),
                       "pairs_per_gpu": B, "points_per_cloud": n_pts, "iterations": args.iters,
                       "parallelism": f"pairs sharded over {world} GPU(s)" + (f" ({args.shard})" if world > 1 else "") + ", one all_gather of poses per step" + (f"; collectives: {collectives}" if world > 1 else ""),
                       "overlap": ((("none (serial steps, synchronous calls" + ("" if args.no_latency_mode else ", m3dreg_set_latency_mode on: the caller says its batches have the GPU to themselves") + ")") if serial_calls else "none (serial steps, one chain)") if D == 1 else f"{D} steps run concurrently, one HIP stream each") +
                                  (f"; {Q} steps queued per stream" if Q > 1 else "")},
            "ms_per_icp_iter_batch": chain_iter_ms if chain_iter_ms > 0 else iter_ms / max(1, iters_timed),
            "ms_per_icp_iter_per_pair": (chain_iter_ms if chain_iter_ms > 0 else iter_ms / max(1, iters_timed)) / B,
            "ms_per_icp_iter_batch_alone": alone_chain_ms if alone_chain_ms > 0 else alone_iter_ms,
            "ms_per_icp_iter_batch_bracketed_chain": iter_ms / max(1, iters_timed),
            "ms_bucketing_batch": buck_ms / max(1, buck_n), "ms_bucketing_batch_alone": alone_bucket_ms,
            "max_rot_err_deg": max_rot, "max_trans_err_m": max_tr,
            "verify": {"steps_checked": verify["checked"], "mismatches": verify["mismatches"], "timed_steps_checked": bool(args.verify_steps),
                       "what": "poses + statistics of every checked step == the first result of the same shard, byte for byte, under the timed region's schedule"},
            "scale_ceiling": scale_ceiling,
            "iterations_executed_pair0": int(last["st"][0].iterations),
            # roofline_definition_version 2 (rounds 3-4): frac = the correspondence step INSIDE the timed region (what the timed region ran, several chains
            # sharing the GPU, sampled brackets); `alone` = the same bracket with nothing else on the GPU; `iteration` = a whole linearisation of the
            # shipped schedule (fused late launches included) on its algorithmic bytes. Version 1 (rounds 1-2) printed the ALONE bracket as `frac`, and
            # `value` was one timed region instead of the median block: compare frac across rounds 2 -> 3 through `alone`, not through `frac`.
            "roofline_definition_version": 2,
            "roofline": {"bound": "hbm", "kernel": "k_nn_iter + k_nn_tiles: the correspondence step of one Gauss-Newton iteration, one event bracket (a bracketed iteration "
                                                    "runs as the launch chain; un-bracketed iterations >= 8 of a level run fused with the reduction as k_icp_late: see `iteration`)",
                         "achieved": in_region, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": in_region / HBM_PEAK_GBS,
                         "avg_launch_ms": 1e3 * avg_launch_s, "launches_timed": launches, "concurrent_chains": D,
                         "frac_source": "hipEvent brackets inside the timed region (every --event-every-th iteration), on the library's stream",
                         "traffic": traffic, "traffic_fetch_uncorrected": traffic_unc,
                         "traffic_source": traffic_source, "traffic_step": ({k: live[k] for k in ("step_bytes", "step_bytes_fetch_uncorrected", "per_kernel_MB_per_step", "note")} if live else None),
                         "l2_hit_rate": l2_hit,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "gather_model": gather_model,
                         "alone": {"avg_launch_ms": alone_ms, "achieved": alone, "frac": alone / HBM_PEAK_GBS,
                                   "source": "one extra, untimed step after the timed region with nothing else on the GPU (the handle in latency mode, as a lone caller's would be), every iteration bracketed: what a rocprofv3 kernel trace of serial steps shows (profiles/)"},
                         "iteration": {"what": "one whole linearisation of the shipped schedule: correspondence step + residuals + 29-term reduction + solve (k_icp_late where it runs)",
                                       "algorithmic_bytes": alg_bytes_iter, "avg_ms": chain_iter_ms, "achieved": iteration_gbps, "frac": iteration_gbps / HBM_PEAK_GBS,
                                       "iterations_timed": chain_n,
                                       "alone": {"avg_ms": alone_chain_ms, "achieved": alone_iteration_gbps, "frac": alone_iteration_gbps / HBM_PEAK_GBS}},
                         # kept for continuity with rounds 1-2 (same numbers as roofline.frac / roofline.alone):
                         "in_region": {"concurrent_chains": D, "avg_launch_ms": 1e3 * avg_launch_s, "achieved": in_region, "frac": in_region / HBM_PEAK_GBS, "launches_timed": launches}},
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
            out["per_rank_note"] = ("own_work_ms = a rank's K timed steps until ITS results are in (before the closing barrier); value divides by the block time = the "
                                    "max over ranks, so (max - mean) / max of own_work_ms_median is the throughput lost to the slowest shard")
        if world == 1 and not args.no_cpu_baseline:
            t_ph = time.perf_counter()
            out["cpu_baseline"] = cpu_baseline(params, host_pairs, args.iters, args.cpu_threads)
            print(f"[bench] cpu_baseline: {time.perf_counter() - t_ph:.1f} s of wall clock", file=sys.stderr, flush=True)
        if args.from_host:
            out["data"] = "synthetic (host PointCloud2 buffers: PCIe-inclusive, not the headline configuration)"
        if world == 1 and not args.no_extra and not args.no_cpu_baseline and args.workload == "config4" and not (args.from_host or args.converge):
            out["legs"] = extra_legs(args)
            # what a caller gets, beside the pipelined headline: one synchronous call at a time, host payloads, one 64-pair call, rotating inputs
            for key, leg in (("value_serial", "serial"), ("value_from_host", "from_host"), ("value_batch64", "batch64_one_chain"), ("value_rotate_pairs", "rotate_pairs"), ("value_unbracketed", "unbracketed")):
                v_ = (out["legs"].get(leg) or {}).get("value")
                if v_ is not None:
                    out[key] = v_
            sc = (out["legs"].get("rotate_pairs") or {}).get("scale_ceiling")
            if sc is not None:
                out["scale_ceiling"] = sc
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


LINE_BUDGET = 6000   # bytes: the driver parses the LAST stdout line and keeps only a few KB of tail (round 4's 27 KB line came back parsed = null)


def _r(x, sig=6):
    """floats to `sig` significant digits (the line is read by people and a parser, not fed back into arithmetic)"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def compact_line(full):
    """The ONE line of the contract, built from the full result: the contract's keys, a numbers-only `roofline`, a compact `cpu_baseline`,
    `legs` as {name: {value, ms_per_step, frac_alone, ...}}. Every definition and every prose field lives in DESIGN.md section 6 (keyed by
    `roofline_definition_version`) and in the full result (gpurun_out/bench_result_full.json + stderr). Always < LINE_BUDGET bytes
    (tests/test_bench_launcher.py builds it from a canned full result and from one eight times as wordy)."""
    pick = lambda d, keys: {k: d[k] for k in keys if isinstance(d, dict) and k in d}
    line = pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    for k, cap in (("metric", 120), ("unit", 24), ("scaling", 16), ("dtype", 48), ("data", 100)):
        if isinstance(line.get(k), str):
            line[k] = line[k][:cap]
    line.update(pick(full, ("value_serial", "value_from_host", "value_batch64", "value_rotate_pairs", "value_unbracketed")))
    if isinstance(full.get("verify"), dict):
        line["verify"] = pick(full["verify"], ("steps_checked", "mismatches", "timed_steps_checked"))
    if isinstance(full.get("scale_ceiling"), dict):
        line["scale_ceiling"] = pick(full["scale_ceiling"], ("n8_over_n1_measured_shards", "n8_over_n1_cost_model", "status"))
    cfg = full.get("config") or {}
    line["config"] = pick(cfg, ("workload", "payload", "pipelining", "pairs_per_gpu", "points_per_cloud", "iterations", "parallelism", "overlap"))
    for k in ("workload", "payload", "pipelining", "parallelism", "overlap"):
        if isinstance(line["config"].get(k), str):
            line["config"][k] = line["config"][k][:480 if k == "workload" else 120]
    line.update(pick(full, ("ms_per_icp_iter_batch", "ms_per_icp_iter_per_pair", "ms_per_icp_iter_batch_alone", "ms_bucketing_batch", "ms_bucketing_batch_alone",
                            "max_rot_err_deg", "max_trans_err_m", "iterations_executed_pair0", "roofline_definition_version",
                            "map_points", "bucket_map_ms", "registration_ms")))
    if isinstance(full.get("blocks"), dict):
        line["blocks"] = pick(full["blocks"], ("n", "median_ms", "min_ms", "max_ms"))
    rf = full.get("roofline")
    if isinstance(rf, dict):
        r = pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_fetch_uncorrected", "algorithmic_bytes_per_launch", "avg_launch_ms",
                      "launches_timed", "concurrent_chains"))
        r["bound"], r["unit"] = str(rf.get("bound", "hbm"))[:4], str(rf.get("unit", "GB/s"))[:8]
        r["kernel"] = str(rf.get("kernel", ""))[:40]
        if isinstance(rf.get("traffic_source"), str):
            r["traffic_source"] = rf["traffic_source"].split(":")[0][:24]   # "measured in this run" / "committed"
        if isinstance(rf.get("traffic_step"), dict):
            r["traffic_step_bytes"] = rf["traffic_step"].get("step_bytes")
            r["traffic_step_bytes_fetch_uncorrected"] = rf["traffic_step"].get("step_bytes_fetch_uncorrected")
        if isinstance(rf.get("alone"), dict):
            r["alone"] = pick(rf["alone"], ("avg_launch_ms", "achieved", "frac"))
        if isinstance(rf.get("iteration"), dict):
            it = pick(rf["iteration"], ("algorithmic_bytes", "avg_ms", "achieved", "frac"))
            if isinstance(rf["iteration"].get("alone"), dict):
                it["alone"] = pick(rf["iteration"]["alone"], ("avg_ms", "achieved", "frac"))
            r["iteration"] = it
        line["roofline"] = r
    elif "roofline" in full:
        line["roofline"] = None
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = pick(cb, ("value", "unit", "cores", "kind"))
        c["unit"], c["kind"] = str(cb.get("unit", ""))[:24], str(cb.get("kind", ""))[:9]
        c["sample"] = str(cb.get("sample_short") or cb.get("sample", ""))[:200]
        for impl in ("port", "kdtree"):
            if isinstance(cb.get(impl), dict):
                c[impl] = {t: cb[impl][t].get("registrations_per_s") for t in ("threads_1", "threads_all") if isinstance(cb[impl].get(t), dict)}
        line["cpu_baseline"] = c
    elif "cpu_baseline" in full:
        line["cpu_baseline"] = None
    if isinstance(full.get("legs"), dict):
        legs = {}
        for name, d in list(full["legs"].items())[:16]:
            if not isinstance(d, dict) or "error" in d:
                legs[name] = {"error": str((d or {}).get("error", "?"))[-80:]} if isinstance(d, dict) else None
                continue
            e = pick(d, ("value", "ms_per_step", "registration_ms", "bucket_map_ms", "generation_ms", "pairs", "accepted", "frac"))
            al = ((d.get("roofline") or {}).get("alone") or {}).get("frac")
            if al is not None:
                e["frac_alone"] = al
            itf = ((d.get("roofline") or {}).get("iteration") or {}).get("frac")
            if itf is not None:
                e["frac_iteration"] = itf
            if isinstance(d.get("levels"), list):
                e["level_ms_per_icp_iter"] = [lv.get("ms_per_icp_iter") for lv in d["levels"][:4]]
            legs[name] = e
        line["legs"] = legs
    if isinstance(full.get("per_rank"), list):
        line["per_rank"] = [pick(p, ("rank", "pairs", "own_work_ms_median")) for p in full["per_rank"][:8] if isinstance(p, dict)]
    line["full_result"] = "gpurun_out/bench_result_full.json (and stderr)"
    line = _r(line)
    # belt and braces: if a future field pushes the line over the budget, optional parts go first — the contract's keys, roofline and cpu_baseline stay
    for drop in ("per_rank", "legs", "blocks"):
        if len(json.dumps(line)) < LINE_BUDGET:
            break
        line.pop(drop, None)
    return line


def emit(full):
    """Full result -> gpurun_out/bench_result_full.json and stderr; the compact line -> stdout, the ONLY thing this run prints there."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_result_full.json"), "w") as f:
            json.dump(full, f)
    except OSError:
        pass
    print("[bench full result] " + json.dumps(full), file=sys.stderr, flush=True)
    if os.environ.get("M3D_BENCH_FULL_LINE"):   # child runs of this script (extra_legs, scripts/): the parent wants every field
        print(json.dumps(full), flush=True)
        return
    s = json.dumps(compact_line(full))
    assert len(s) < LINE_BUDGET, len(s)
    print(s, flush=True)


def measure_traffic(args):
    """HBM-side bytes of the correspondence step and of a whole step, measured IN THIS RUN when rocprofv3 is on the box: two child runs of this script
    (serial steps, no event brackets) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` — separate passes, nothing else traced, the program
    itself behind `--` — read per kernel. FETCH_SIZE is doubled (the guide's gfx950 correction for wide coalesced reads; these kernels mix 16-B streams with
    16-B gathers, for which it is uncalibrated: the uncorrected sum is reported beside it). Returns None when the tool is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3")
    if not tool:
        return None
    steps_timed = 3
    per = {}
    try:
        for ci, ctr in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                cmd = [tool, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", td, "--", sys.executable, os.path.abspath(__file__),
                       "--steps", str(steps_timed), "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--inflight", "1", "--queue-depth", "1", "--no-events", "--min-seconds", "0",
                       "--iters", str(args.iters), "--azimuth", str(args.azimuth), "--pairs-per-gpu", str(args.pairs_per_gpu)]
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
                files = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
                if r.returncode != 0 or not files:
                    return None
                for f in files:
                    for row in csv.DictReader(open(f)):
                        k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
                        e = per.setdefault(k, [0.0, 0.0, 0])
                        e[ci] += float(row["Counter_Value"]) * 1024.0
                        if ci == 0:
                            e[2] += 1
    except Exception:
        return None
    steps = steps_timed + 2   # + the step that sizes the workload and one warm-up step per handle
    nn = [per.get(k, [0.0, 0.0, 0]) for k in ("k_nn_iter", "k_nn_tiles")]
    launches = max(1, per.get("k_nn_tiles", [0, 0, 1])[2])   # the iterations that run the tile search: k_nn_iter + k_nn_tiles, one launch each
    return {"correspondence_step_bytes_per_launch": (2.0 * sum(e[0] for e in nn) + sum(e[1] for e in nn)) / launches,
            "correspondence_step_bytes_per_launch_fetch_uncorrected": (sum(e[0] for e in nn) + sum(e[1] for e in nn)) / launches,
            "step_bytes": sum(2.0 * e[0] + e[1] for e in per.values()) / steps, "step_bytes_fetch_uncorrected": sum(e[0] + e[1] for e in per.values()) / steps,
            "per_kernel_MB_per_step": {k: round((2.0 * e[0] + e[1]) / steps / 1e6, 1) for k, e in sorted(per.items(), key=lambda kv: -(2.0 * kv[1][0] + kv[1][1])) if 2.0 * e[0] + e[1] > 5e5},
            "note": "k_nn_iter + k_nn_tiles of the iterations that run the tile search (the fused k_icp_late launches of the later iterations are in step_bytes)"}


def extra_legs(args):
    """The other BASELINE configurations and the SURVEY 8d variants of the headline, each a CHILD run of this script on the same GPU
    right after the headline (a child process, started — never exec'ed — from this one; it prints its own JSON line)."""
    import subprocess
    base = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-extra", "--iters", str(args.iters), "--azimuth", str(args.azimuth)]
    runs = {
        "serial": ["--steps", "30", "--warmup", "3", "--inflight", "1", "--queue-depth", "1"],
        # the headline's schedule with NO event brackets in the timed region (what the instrument costs: every record is a barrier packet on its stream)
        "unbracketed": ["--steps", "20", "--warmup", "5", "--no-events"],
        "from_host": ["--steps", "40", "--warmup", "3", "--from-host"],
        "converge": ["--steps", "40", "--warmup", "3", "--converge"],
        # (the default bracket, every 13th iteration: with every iteration bracketed none of them runs fused, and these two legs are LATENCIES of the shipped schedule)
        "config3": ["--workload", "config3", "--steps", "60", "--warmup", "5", "--inflight", "1", "--queue-depth", "1"],
        "config2": ["--workload", "config2", "--steps", "60", "--warmup", "5", "--inflight", "1", "--queue-depth", "1"],
        "config5": ["--workload", "config5", "--steps", "10", "--warmup", "2"],
        # the big-batch regime: B pairs per step in ONE launch chain — 64 = ALL of config 4 on one GPU (pairs 0 ... 63: also the ones next to
        # obstacles, which rank 0's 8-pair shard does not hold) — with one chain (the kernels' own throughput: nothing else on the GPU) and with two in flight
        "batch64_one_chain": ["--pairs-per-gpu", "64", "--inflight", "1", "--queue-depth", "1", "--steps", "6", "--warmup", "2"],
        # the headline's schedule with the inputs changing under it: step k registers LPT shard k mod 8 of the 64 pairs (every step checked byte for byte against the
        # first result of its shard), then every shard as the whole workload in turn: the ceiling of the 8-GPU line (scale_ceiling)
        "rotate_pairs": ["--rotate-pairs", "--verify-steps", "--steps", "24", "--warmup", "8"],
        # f4: loop-closure candidate generation + the 64-pair batch it emits, end to end (run_loop64)
        "loop64": ["--workload", "loop64", "--steps", "4", "--warmup", "1"],
    }
    if args.all_legs:   # not in the driver's command: its run must stay well under a minute
        runs.update({
            "batch16": ["--pairs-per-gpu", "16", "--inflight", "2", "--queue-depth", "2", "--steps", "10", "--warmup", "3"],
            "batch32": ["--pairs-per-gpu", "32", "--inflight", "2", "--queue-depth", "2", "--steps", "8", "--warmup", "2"],
            "batch64": ["--pairs-per-gpu", "64", "--inflight", "2", "--queue-depth", "2", "--steps", "6", "--warmup", "2"],
            "serial_async_bracketed": ["--steps", "30", "--warmup", "3", "--inflight", "1", "--queue-depth", "1", "--async-calls"],   # rounds 1-4's `serial`: async call + wait, event brackets in the timed region
            "from_host_converge": ["--steps", "40", "--warmup", "3", "--from-host", "--converge"],   # SURVEY 8d's literal "registrations/s": H2D of both clouds + bucketing + iterations to eps 1e-5 (at most --iters) + D2H
        })
    legs = {}
    for name, extra in runs.items():
        t_leg = time.perf_counter()
        try:
            r = subprocess.run(base + extra + ["--min-seconds", "0.5"], capture_output=True, text=True, timeout=300, env=dict(os.environ, M3D_BENCH_FULL_LINE="1"))
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                legs[name] = {"error": (r.stderr or r.stdout)[-300:]}
                continue
            d = json.loads(line[-1])
            keep = {k: d[k] for k in ("value", "unit", "ms_per_step", "ms_per_icp_iter_batch", "ms_per_icp_iter_batch_alone", "ms_bucketing_batch_alone",
                                      "iterations_executed_pair0", "max_rot_err_deg", "max_trans_err_m") if k in d}
            keep["workload"] = d.get("config", {}).get("workload")
            keep["overlap"] = d.get("config", {}).get("overlap")
            if "roofline" in d:
                keep["roofline"] = {k: d["roofline"].get(k) for k in ("achieved", "frac", "avg_launch_ms", "algorithmic_bytes_per_launch", "unit", "alone", "iteration")}
            for k in ("levels", "map_points", "bucket_map_ms", "bucket_map_wall_ms", "registration_ms", "blocks", "scale_ceiling", "verify", "generation_ms", "pairs", "accepted", "keyframes",
                      "candidates_per_s", "batch_ms"):
                if k in d:
                    keep[k] = d[k]
            legs[name] = keep
        except Exception as e:
            legs[name] = {"error": repr(e)}
        print(f"[bench] leg {name}: {time.perf_counter() - t_leg:.1f} s of wall clock", file=sys.stderr, flush=True)
    return legs


def run_config5(args):
    """BASELINE config 5: a 100k-point live scan against the 2 066 481-point aggregated map (synth.config5: 22 sweeps along a 10 m trajectory,
    de-duplicated at 1 cm; rounds 1-3 built 20 sweeps at 2 cm = 1.51 M points: scripts/c5_old_map.py runs that map, like for like with their
    numbers), multi-resolution 0.4 / 0.2 / 0.1 m, 10 iterations each, point-to-plane. A single giant pair does not shard ("replicas only",
    SURVEY 8e). Reported: the whole registration (sort of the scan + 30 iterations; the map is bucketed once, outside), ms per iteration and
    the roofline of the correspondence step PER LEVEL (one run per level alone, started from the previous levels' result), on
    12 N + 12 M + 8 C_occ bytes — the one workload whose iteration working set (tens of MB) is larger than an XCD's L2."""
    from mandala_mapping_amd import abi, binding, synth
    leaves, dmaxs = (0.4, 0.2, 0.1), (1.0, 0.6, 0.5)
    p = abi.Params.make(leaf=leaves, iterations=(10, 10, 10), max_corr_dist=dmaxs, metric=abi.POINT_TO_PLANE, normal_leaf=0.4)
    live, mp, Tgt, T0 = synth.config5()
    R = binding.Registrar(p)
    R.profile_enable(True, every=1 << 30)              # (only the bucketing bracket matters here)
    t0 = time.perf_counter()
    tgt = R.cloud(mp)
    R.synchronize()
    bucket_wall_ms = 1e3 * (time.perf_counter() - t0)  # wall clock: includes the pageable H2D copy of the 24 MB payload and the first-touch allocations
    R.profile_read(2, reset=True)
    t0 = time.perf_counter()
    tgt2 = R.cloud(mp); R.synchronize()
    bucket_wall_ms = min(bucket_wall_ms, 1e3 * (time.perf_counter() - t0))
    nb_, mb_ = R.profile_read(2, reset=True)
    bucket_ms = mb_ / max(1, nb_)                      # the bucketing pipeline itself: hipEvents on the library's stream around its kernels (three levels + normal grid + tiles)
    R.profile_enable(False)
    tgt2.free()
    from mandala_mapping_amd.pointcloud2 import encode_xyz
    live_msg = encode_xyz(live)        # the PointCloud2 message as the producer hands it over (encoding it is the aggregator's work, not the registration's)
    src = R.clouds([live_msg], source_only=[True])[0]
    times = []
    for i in range(args.warmup + args.steps):
        R.synchronize(); t0 = time.perf_counter()
        s_ = R.clouds([live_msg], source_only=[True])[0]
        T, st = R.align(s_, tgt, T0)
        times.append(1e3 * (time.perf_counter() - t0))
        s_.free()
    times = sorted(times[args.warmup:])
    reg_ms = times[len(times) // 2]
    rot, tra = synth.pose_error(T, Tgt)
    # per level: a one-level registration on that level's grid alone, started where the coarser levels ended
    levels, Tl = [], T0
    for l in range(3):
        # (a coarser level is measured as what it is in the pyramid — a level above the finest one, whose grid is therefore sorted from the finest
        #  level's order: two levels, no iterations on the second)
        pl = (abi.Params.make(leaf=leaves[l], iterations=10, max_corr_dist=dmaxs[l], metric=abi.POINT_TO_PLANE, normal_leaf=0.4) if l == 2 else
              abi.Params.make(leaf=(leaves[l], leaves[2]), iterations=(10, 0), max_corr_dist=(dmaxs[l], dmaxs[2]), metric=abi.POINT_TO_PLANE, normal_leaf=0.4))
        Rl = binding.Registrar(pl)
        tl, sl = Rl.clouds([mp, live], source_only=[False, True])
        Rl.align(sl, tl, Tl)                       # warm-up (pools, workspaces)
        Rl.profile_enable(True, every=1)
        for w_ in range(4):
            Rl.profile_read(w_, reset=True)
        Tn, stl = Rl.align(sl, tl, Tl)
        n1, ms1 = Rl.profile_read(1, reset=True)
        n0, ms0 = Rl.profile_read(0, reset=True)
        g = tl.grid_info(0)
        alg = 12 * sl.n + 12 * tl.n + 8 * g.n_cells
        nn_ms = ms1 / max(1, n1)
        levels.append({"leaf": leaves[l], "occupied_voxels": int(g.n_cells), "ms_per_icp_iter": ms0 / max(1, n0), "ms_correspondence_step": nn_ms,
                       "algorithmic_bytes_per_iteration": alg, "achieved_GBps": alg / (nn_ms / 1e3) / 1e9 if nn_ms > 0 else 0.0,
                       "frac": alg / (nn_ms / 1e3) / 1e9 / HBM_PEAK_GBS if nn_ms > 0 else 0.0, "iterations": int(stl.iterations)})
        Tl = Tn
        del tl, sl, Rl
    fin = levels[-1]
    emit({
        "metric": "scan-to-map registrations/sec (100k-pt live scan vs ~2M-pt map, multi-resolution 0.4/0.2/0.1 m, point-to-plane)",
        "value": 1e3 / reg_ms, "unit": "registrations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": reg_ms,
        "higher_is_better": True, "scaling": "replicas only", "vs_baseline": None, "dtype": "f32 (int64 fixed-point sums, f64 solve)", "data": "synthetic",
        "config": {"workload": f"BASELINE config 5: {live.shape[0]}-point live scan against a {mp.shape[0]}-point map, leaves 0.4 / 0.2 / 0.1 m x 10 iterations, "
                               "point-to-plane; the map is bucketed once (outside the timed region), the scan is decoded and sorted inside it"},
        "map_points": int(mp.shape[0]), "bucket_map_ms": bucket_ms, "bucket_map_wall_ms": bucket_wall_ms, "registration_ms": reg_ms, "levels": levels,
        "ms_per_icp_iter_batch": sum(x["ms_per_icp_iter"] for x in levels) / 3.0,
        "max_rot_err_deg": rot, "max_trans_err_m": tra, "iterations_executed_pair0": int(st.iterations),
        "roofline": {"bound": "hbm", "kernel": "k_nn_iter + k_nn_tiles, finest level", "achieved": fin["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": fin["frac"], "avg_launch_ms": fin["ms_correspondence_step"], "algorithmic_bytes_per_launch": fin["algorithmic_bytes_per_iteration"], "traffic": None},
    })


def run_loop64(args):
    """SURVEY §8 row f4, second half, end to end: a closed trajectory of 80 full-size keyframes (two laps of synth.loop_trajectory's ellipse,
    100 000 rays per sweep, resident in HBM as bucketed clouds with their signatures — what a node has after driving it); a step = candidate generation for
    the second lap's keyframes (m3dloop_candidates: one scoring pass over the signature database, top-2 per keyframe) -> the first 64 candidates as
    m3dreg_pair[] (m3dloop_make_pairs) -> ONE m3dreg_align_batch (the headline's parameters: point-to-plane, leaf 0.1 m, 20 fixed iterations) -> m3dloop_gate.
    value = 64 registrations / (generation + batch). Also: the device time and the bytes of the scoring pass for the WHOLE table (all 80 rows)."""
    import multiprocessing as mp
    from mandala_mapping_amd import synth
    poses = synth.loop_poses(n_keyframes=80, per_lap=40, seed=9300)
    with mp.get_context("fork").Pool(min(host_cores(), 16)) as pool:   # (before this process touches the GPU)
        scans = pool.starmap(synth.hdl32_scan, [(T, args.azimuth, sd, 0.02, 0.4, 100.0, 1.0) for T, _, sd in poses])
    from mandala_mapping_amd import abi, binding
    params = abi.Params.make(leaf=0.1, iterations=args.iters, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    R = binding.Registrar(params)
    R.set_latency_mode(True)
    P = abi.LoopParams.make(sig_leaf=2.0, sig_log2_bits=16, radius=3.0, min_gap=20, top_k=2, min_overlap=0.5, max_keyframes=128)
    G = binding.LoopCloser(R, P)
    clouds = []
    for i in range(0, len(scans), 16):
        clouds += R.clouds(scans[i:i + 16])
    t0 = time.perf_counter()
    for c, (_, T_odo, _) in zip(clouds, poses):
        G.add_keyframe(c, T_odo)
    R.synchronize()
    sign_ms = 1e3 * (time.perf_counter() - t0) / len(clouds)
    NP = 64
    gen, bat, acc_n = [], [], 0
    worst = (0.0, 0.0)
    for it in range(args.warmup + args.steps):
        R.synchronize()
        t0 = time.perf_counter()
        cands = G.candidates(40, -1)
        if len(cands) < NP:
            raise SystemExit(f"loop64: only {len(cands)} candidates")
        sel = (abi.LoopCandidate * NP)(*[cands[i] for i in range(NP)])
        pairs = G.pairs(sel)
        t1 = time.perf_counter()
        T, st = R.align_batch_arr(pairs, NP)
        acc = G.gate(sel, st, min_corr=20000, max_rms=0.05)
        t2 = time.perf_counter()
        if it >= args.warmup:
            gen.append(1e3 * (t1 - t0)); bat.append(1e3 * (t2 - t1))
        acc_n = sum(acc)
        for i in range(NP):
            if acc[i]:
                e = synth.pose_error(T[i], synth.inv_T(poses[sel[i].target][0]) @ poses[sel[i].source][0])
                worst = (max(worst[0], e[0]), max(worst[1], e[1]))
    G.candidates(0, -1)                                  # the whole table: every row against every older keyframe
    all_ms, all_bytes = G.last_profile()
    G.candidates(len(clouds) - 1, 1)                     # the node's call: the newest keyframe against the database
    row_ms, row_bytes = G.last_profile()
    gen_ms, bat_ms = sorted(gen)[len(gen) // 2], sorted(bat)[len(bat) // 2]
    ach = all_bytes / (all_ms / 1e3) / 1e9 if all_ms > 0 else 0.0
    emit({
        "metric": "loop-closure registrations/sec, candidate generation included (64 pairs of 100k-pt keyframes, point-to-plane, 0.1 m voxel NN)",
        "value": NP / ((gen_ms + bat_ms) / 1e3), "unit": "registrations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": gen_ms + bat_ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (int64 fixed-point sums, f64 solve); u32 bitmaps", "data": "synthetic",
        "config": {"workload": f"f4 loop closure: {len(clouds)} keyframes x {args.azimuth * 32} rays on a closed two-lap trajectory, resident in HBM; per step: m3dloop_candidates for rows 40.. "
                               f"(top-2, radius 3 m, gap 20) -> first {NP} candidates -> m3dloop_make_pairs -> one synchronous m3dreg_align_batch ({args.iters} fixed iterations) -> m3dloop_gate",
                   "payload": "hbm-resident", "pipelining": "none: synchronous calls"},
        "keyframes": len(clouds), "pairs": NP, "accepted": int(acc_n), "generation_ms": gen_ms, "batch_ms": bat_ms, "signature_ms_per_keyframe": sign_ms,
        "max_rot_err_deg": worst[0], "max_trans_err_m": worst[1],
        "roofline": {"bound": "hbm", "kernel": "k_loop_score + k_loop_topk, all 80 rows", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "avg_launch_ms": all_ms, "algorithmic_bytes_per_launch": int(all_bytes), "traffic": None,
                     "newest_row": {"ms": row_ms, "bytes": int(row_bytes)}},
    })


def host_cores():
    """The cores this process may really use: the smaller of its affinity mask and its cgroup CPU quota (a GPU box reports every core of
    the host through os.cpu_count() but grants the job a share of them), at most 64 (beyond that the per-iteration fork/join of the
    OpenMP loops outweighs the work at these cloud sizes)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            if f.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, min(n, 64))


def _median_time(fn, budget_s=12.0):
    """warm-up once, then the median of 5 runs (3 when one run takes more than a fifth of the budget); returns (seconds, runs, spread)"""
    t0 = time.perf_counter(); fn(); w = time.perf_counter() - t0
    k = 5 if 6 * w <= budget_s else 3
    ts = []
    for _ in range(k):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], k, (ts[-1] - ts[0]) / ts[len(ts) // 2]


def cpu_baseline(params, host_pairs, iters, threads=0):
    """this repo's own CPU implementations (the reference has none), on THIS host's cores, -O3 -march=native (compiled here,
    oracle/orc.py build_native), warm-up + median: (i) `port`: the voxel algorithm of the HIP path (oracle/m3d_oracle.c: the source cloud
    source-only like the GPU leg: sorted, no normals), (ii) `kdtree`: a from-scratch k-d tree ICP (oracle/m3d_kdtree_icp.c: what
    pcl::IterativeClosestPoint with a point-to-plane estimator does), each at 1 thread and at all cores, one registration of the first
    pair per run (the 1-thread legs) / of up to 4 pairs (the all-core legs). The headline `value` is the port at all cores."""
    from oracle import orc
    import numpy as np
    cores = threads or host_cores()
    orc.native_libs()
    L = orc.native_oracle()
    src0, tgt0 = host_pairs[0]
    sample = host_pairs[:4]
    out = {}

    def port(pairs, th):
        L.orc_set_threads(th)
        for s, t in pairs:
            cs = orc.Cloud(params, s, omp=True, source_only=True)
            ct = orc.Cloud(params, t, omp=True)
            orc.align(params, cs, ct)

    def kd(pairs, th):
        for s, t in pairs:
            orc.kdtree_icp(s, t, 1, float(params.max_corr_dist[0]), iters, threads=th)

    saved = orc._LIBS.get("libm3d_oracle_omp.so")
    orc._LIBS["libm3d_oracle_omp.so"] = L          # orc.Cloud(omp=True) / orc.align then run on the -O3 -march=native object
    try:
        for name, fn in (("port", port), ("kdtree", kd)):
            t1, k1, sp1 = _median_time(lambda: fn([(src0, tgt0)], 1))
            tn, kn, spn = _median_time(lambda: fn(sample, cores))
            out[name] = {"threads_1": {"registrations_per_s": 1.0 / t1, "s_per_registration": t1, "runs": k1, "spread": sp1},
                         "threads_all": {"registrations_per_s": len(sample) / tn, "s_per_registration": tn / len(sample), "threads": cores, "runs": kn, "spread": spn}}
    finally:
        if saved is not None:
            orc._LIBS["libm3d_oracle_omp.so"] = saved
        else:
            orc._LIBS.pop("libm3d_oracle_omp.so", None)
    return {"value": out["port"]["threads_all"]["registrations_per_s"], "unit": "registrations/s", "cores": cores, "kind": "port",
            "port": out["port"], "kdtree": out["kdtree"],
            "sample": f"warm-up + median of 5 (3 when a run exceeds 2 s): one registration of the first scan pair (1 thread) / of {len(sample)} of the same pairs "
                      f"({cores} threads): bucketing (+ normals of the target; the source sorted only) + {iters} iterations; gcc -O3 -march=native -fopenmp on this host; "
                      "port = oracle/m3d_oracle.c (the voxel algorithm, bit-identical to the HIP path), kdtree = oracle/m3d_kdtree_icp.c (k-d tree + kNN-PCA normals + "
                      "point-to-plane Gauss-Newton: what pcl::IterativeClosestPoint would do)"}


if __name__ == "__main__":
    main()
