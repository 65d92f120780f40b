"""The schedule the headline is measured on, checked bit for bit.

bench.py's headline keeps eight handles on four HIP streams busy — two steps queued per stream, sampled hipEvent brackets on,
clouds recycled through the handles' block pools while other batches are in flight, payloads resident in HBM. No parity test
looked at exactly that configuration at full size; this one does: every step's eight poses and statistics must equal what ONE
handle returns for the same shard with a synchronous call (which tests/test_gpu_parity.py compares with the oracle:
test_config4_all_shards, test_full_size_pairs_equal_the_oracle)."""
import ctypes as C

import numpy as np
import pytest

from mandala_mapping_amd import abi, synth
from mandala_mapping_amd.pointcloud2 import encode_xyz

pytestmark = pytest.mark.gpu


def _shards():
    import json, os
    from mandala_mapping_amd import sharding
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    costs, _ = sharding.table_costs(json.load(open(os.path.join(root, "mandala_mapping_amd", "config4_costs.json"))), 64)
    return sharding.lpt_assign(costs, 8, capacity=8)


def _sig(T, st):
    return np.asarray(T, np.float64).tobytes() + b"".join(bytes(x) for x in st)


def _explain(sig, ref, n_pairs):
    """which pairs of a step differ from the reference, and by how much (the signature = n_pairs x 16 float64 pose words, then n_pairs x 40 bytes of statistics)"""
    T, Tr = np.frombuffer(sig[: 128 * n_pairs], np.float64).reshape(n_pairs, 16), np.frombuffer(ref[: 128 * n_pairs], np.float64).reshape(n_pairs, 16)
    out = []
    for j in range(n_pairs):
        a, b = sig[128 * n_pairs + 40 * j: 128 * n_pairs + 40 * (j + 1)], ref[128 * n_pairs + 40 * j: 128 * n_pairs + 40 * (j + 1)]
        if not np.array_equal(T[j], Tr[j]) or a != b:
            sa, sb = abi.Stats.from_buffer_copy(a), abi.Stats.from_buffer_copy(b)
            out.append(f"pair {j}: max |dT| {np.abs(T[j] - Tr[j]).max():.3e}, stats {sa.as_dict()} vs {sb.as_dict()}")
    return "; ".join(out)


def _pipelined(reg, torch, params, payloads, steps, inflight=4, queue=2, every=7, rotate=None, inspect=None):
    """bench.py run_steps(): handle j on stream j % inflight; step i on handle i % (inflight * queue); a finished step's clouds go back to
    their handle's pool before the next step is enqueued on it. payloads: list of shards, each a list of (src tensor, n, tgt tensor, n);
    step i registers shard rotate(i) (default: shard 0 for every step). Returns the list of (shard, signature) per step.
    inspect(i, clouds): called with a finished step's clouds before they are freed."""
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(inflight)]
    regs = [reg.Registrar(params, device=0, stream=C.c_void_p(streams[j % inflight].cuda_stream)) for j in range(inflight * queue)]
    for r in regs:
        r.profile_enable(every > 0, every=max(1, every))
    B = len(payloads[0])

    def enqueue(i):
        r = regs[i % len(regs)]
        sh = rotate(i) if rotate else 0
        items = []
        for ds, ns, dt, nt in payloads[sh]:
            items += [(ds.data_ptr(), ns), (dt.data_ptr(), nt)]
        cl = r.clouds_from_device(items, wait=False, source_only=[True, False] * B)
        r.align_batch_async(r._pairs([(cl[2 * j], cl[2 * j + 1], None) for j in range(B)]), B)
        return sh, cl

    out, pending, nxt = [], [], 0
    while nxt < min(len(regs), steps):
        pending.append((nxt, enqueue(nxt))); nxt += 1
    for i in range(steps):
        idx, (sh, cl) = pending.pop(0)
        T, st = regs[idx % len(regs)].batch_wait(B)
        out.append((sh, _sig(T, st)))
        if inspect:
            inspect(idx, cl)
        for c in cl:
            c.free()
        if nxt < steps:
            pending.append((nxt, enqueue(nxt))); nxt += 1
    torch.cuda.synchronize()
    for r in regs:
        r.profile_read(0, reset=True); r.profile_read(1, reset=True); r.profile_read(4, reset=True)
        r.profile_enable(False)
        r.close()
    return out


def _resident(torch, pairs):
    dev = torch.device("cuda", 0)
    out = []
    for src, tgt, _ in pairs:
        ms, mt = encode_xyz(src), encode_xyz(tgt)
        out.append((torch.frombuffer(bytearray(ms.data), dtype=torch.uint8).to(dev), ms.n, torch.frombuffer(bytearray(mt.data), dtype=torch.uint8).to(dev), mt.n))
    torch.cuda.synchronize()
    return out


R5_SHARDS = {"r5_shard6": [8, 10, 15, 22, 29, 41, 51, 52], "r5_shard0": [2, 19, 20, 24, 31, 35, 50, 59]}   # profiles/r05_shards.txt


@pytest.mark.parametrize("which", ["r5_shard6", "r5_shard0", "heaviest_now"])
def test_headline_schedule_every_step_equals_the_single_handle(reg, which):
    """BASELINE config 4 at full size under the headline's schedule, 240 steps per shard: the shard round 5's one unexplained GPU memory fault happened on
    (its shard 6), that round's heaviest (shard 0: pair 31) and the shard that holds pair 31 in the table `bench.py --gpus 8` uses now. Eight handles / four
    streams / two queued per stream, brackets every 7th iteration. Every step == the single-handle synchronous result, poses and statistics, byte for byte;
    and the poses are on the generator's ground truth."""
    import torch
    p = abi.Params.make(leaf=0.1, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    shard = R5_SHARDS[which] if which in R5_SHARDS else [x for x in _shards() if 31 in x][0]
    data = [synth.config4_pair(k) for k in shard]
    pay = _resident(torch, data)
    R = reg.Registrar(p)
    items = []
    for ds, ns, dt, nt in pay:
        items += [(ds.data_ptr(), ns), (dt.data_ptr(), nt)]
    cl = R.clouds_from_device(items, source_only=[True, False] * 8)
    Tref, stref = R.align_batch([(cl[2 * j], cl[2 * j + 1], None) for j in range(8)])
    for j in range(8):
        rot, tra = synth.pose_error(Tref[j], data[j][2])
        assert stref[j].status == abi.MAX_ITERATIONS and rot < 0.1 and tra < 0.006, (shard[j], rot, tra)
    ref = _sig(Tref, stref)
    for c in cl:
        c.free()
    got = _pipelined(reg, torch, p, [pay], steps=240)
    bad = [i for i, (_, s) in enumerate(got) if s != ref]
    assert not bad, f"shard {which}: {len(bad)} of {len(got)} pipelined steps differ from the single-handle result (steps {bad[:8]}; first: {_explain(got[bad[0]][1], ref, 8)})"


def test_headline_schedule_rotating_shards_and_bracket_densities(reg):
    """The same schedule with the WORK changing under the handles: step k registers shard (k mod 3) of three different 4-pair shards (pools hand a block
    that held one cloud to another, per-query state of one pair is overwritten by another's), with every iteration bracketed, every 7th and none.
    Every step == its shard's single-handle result."""
    import torch
    p = abi.Params.make(leaf=0.1, iterations=12, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    shards = [[31, 2, 40, 7], [4, 25, 11, 58], [17, 29, 39, 63]]   # the crowded pairs of config 4 mixed with ordinary ones
    data = {k: synth.config4_pair(k, 1600) for s in shards for k in s}   # 51 200 rays per sweep: three shards stay resident and the test stays short
    pays = [_resident(torch, [data[k] for k in s]) for s in shards]
    R = reg.Registrar(p)
    refs = []
    for pay in pays:
        items = []
        for ds, ns, dt, nt in pay:
            items += [(ds.data_ptr(), ns), (dt.data_ptr(), nt)]
        cl = R.clouds_from_device(items, source_only=[True, False] * 4)
        T, st = R.align_batch([(cl[2 * j], cl[2 * j + 1], None) for j in range(4)])
        refs.append(_sig(T, st))
        for c in cl:
            c.free()
    assert len(set(refs)) == 3
    for every in (1, 7, 0):
        got = _pipelined(reg, torch, p, pays, steps=90, every=every, rotate=lambda i: i % 3)
        bad = [i for i, (sh, s) in enumerate(got) if s != refs[sh]]
        assert not bad, f"brackets every {every}: {len(bad)} of {len(got)} steps differ (steps {bad[:8]}, shard {got[bad[0]][0]}; first: {_explain(got[bad[0]][1], refs[got[bad[0]][0]], 4)})"


def test_many_handle_lifetimes_and_the_tile_images_themselves(reg):
    """Regression for the race rounds 5 and 6 shipped in the tile builder (bucket.hip tile_build_role): on the one-image path a wave that was late to
    the test of an LDS flag could read the value a faster wave had already re-used the word for, take its tile for oversize and leave - its share of the tile
    image's stores never happened, the image kept what the pool block held before (an older image of the same tile: a slightly different registration; another
    cloud's: a grossly wrong one, or sorted positions beyond the cloud: the GPU memory fault of round 5). One pipelined step in ~10 000 on crowded tiles, and
    never in one long run on one set of handles - so this test lives the way the suite does: 160 lifetimes of eight fresh handles on four NEW streams, 120 steps
    each over three rotating shards that hold config 4's crowded pairs. Every step == its shard's single-handle result; and of every 3rd lifetime's every 16th
    step the target clouds' tile IMAGES are read back and checked against the clouds' own sorted points (mandala_mapping_amd/diag.py), whatever the pose came to."""
    import torch
    from mandala_mapping_amd import diag
    p = abi.Params.make(leaf=0.1, iterations=12, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    shards = [[31, 2, 40, 7], [17, 57, 22, 10], [32, 29, 39, 63]]
    data = {k: synth.config4_pair(k, 1600) for s in shards for k in s}
    pays = [_resident(torch, [data[k] for k in s]) for s in shards]
    R = reg.Registrar(p)
    refs = []
    for pay in pays:
        items = []
        for ds, ns, dt, nt in pay:
            items += [(ds.data_ptr(), ns), (dt.data_ptr(), nt)]
        cl = R.clouds_from_device(items, source_only=[True, False] * 4)
        for j in range(4):
            assert diag.tile_image_problems(cl[2 * j + 1]) == []   # (the checker itself, on clouds nobody raced for)
        T, st = R.align_batch([(cl[2 * j], cl[2 * j + 1], None) for j in range(4)])
        refs.append(_sig(T, st))
        for c in cl:
            c.free()
    for life in range(160):
        problems = []

        def inspect(i, cl, life=life, problems=problems):
            if life % 3 == 0 and i % 16 == 5:
                for j in range(4):
                    problems += [f"step {i} pair {j}: {m}" for m in diag.tile_image_problems(cl[2 * j + 1])]
        got = _pipelined(reg, torch, p, pays, steps=120, every=(7, 1, 0)[life % 3], rotate=lambda i: i % 3, inspect=inspect)
        assert not problems, f"lifetime {life}: {problems[:4]}"
        bad = [i for i, (sh, s) in enumerate(got) if s != refs[sh]]
        assert not bad, f"lifetime {life}: {len(bad)} of {len(got)} steps differ (steps {bad[:8]}, shard {got[bad[0]][0]}; first: {_explain(got[bad[0]][1], refs[got[bad[0]][0]], 4)})"
