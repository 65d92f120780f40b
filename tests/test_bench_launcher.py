"""bench.py's own rank launcher (`python bench.py --gpus N` without torchrun): N fresh rank processes with torchrun's environment,
the same arguments, one rendezvous port — checked here without a GPU (M3D_BENCH_DRYRUN makes a rank print what it was started with and leave).
The launching process must never import torch or the library (a process that has touched the GPU must not start the ranks)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "M3D_BENCH_RANK_PROCESS")}
    env.update(M3D_BENCH_DRYRUN="1", **(env_extra or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]


def test_gpus_n_without_torchrun_starts_n_rank_processes():
    out = _run(["--gpus", "3", "--steps", "7", "--warmup", "2", "--dist-backend", "gloo"])
    assert sorted(o["RANK"] for o in out) == ["0", "1", "2"] and all(o["LOCAL_RANK"] == o["RANK"] for o in out)
    assert all(o["WORLD_SIZE"] == "3" and o["MASTER_ADDR"] == "127.0.0.1" for o in out) and len({o["MASTER_PORT"] for o in out}) == 1
    assert all(o["argv"] == ["--gpus", "3", "--steps", "7", "--warmup", "2", "--dist-backend", "gloo"] for o in out)


def test_under_torchrun_the_script_is_one_rank_and_starts_nothing():
    out = _run(["--gpus", "4"], {"RANK": "2", "LOCAL_RANK": "2", "WORLD_SIZE": "4", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert len(out) == 1 and out[0]["RANK"] == "2" and out[0]["WORLD_SIZE"] == "4"


def test_spawn_puts_the_launcher_in_front_of_one_rank_and_n1_runs_in_process():
    assert [o["WORLD_SIZE"] for o in _run(["--spawn", "--gpus", "1"])] == ["1"]
    assert _run(["--gpus", "1"])[0]["RANK"] is None      # no launcher, no rank environment: the plain N = 1 run


def test_the_launching_process_imports_neither_torch_nor_the_library():
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2']; import os; os.environ['M3D_BENCH_DRYRUN'] = '1';"
            "runpy.run_path(%r, run_name='__main__'); bad = [m for m in sys.modules if m == 'torch' or m.startswith('mandala_mapping_amd')]; assert not bad, bad" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "M3D_BENCH_RANK_PROCESS")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr


def test_a_failing_rank_ends_the_others_and_its_code_is_the_run_s():
    """the launcher polls all ranks; the first non-zero exit terminates the siblings (which would otherwise wait in the
    rendezvous until the collective's timeout) and becomes the launcher's exit code — promptly."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "M3D_BENCH_RANK_PROCESS")}
    env.update(M3D_BENCH_DRYRUN="1", M3D_BENCH_DRYRUN_FAIL_RANK="1")
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 3 and time.monotonic() - t0 < 60, (r.returncode, r.stderr)
    assert "rank process 1 exited with code 3" in r.stderr


def test_the_launcher_s_overall_timeout():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "M3D_BENCH_RANK_PROCESS")}
    env.update(M3D_BENCH_DRYRUN="1", M3D_BENCH_DRYRUN_FAIL_RANK="7")   # nobody fails: every rank "hangs"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-timeout", "3"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 124, (r.returncode, r.stderr)


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_the_one_stdout_line_stays_under_6000_bytes():
    """round 4's line was 27 KB and the driver could not parse it. The compact line is built here from that very result
    (profiles/r04_final_bench.json, the full dict of the round-4 run) and from a copy of it with every string eight times as long and
    twice the legs: < 6000 bytes, round-trips through json, keeps the contract's keys, a numeric roofline and a cpu_baseline."""
    b = _bench_module()
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_final_bench.json")))
    assert len(json.dumps(full)) > 20000   # the canned result really is the oversized one

    def wordy(x):
        if isinstance(x, str):
            return x * 8
        if isinstance(x, dict):
            return {k: wordy(v) for k, v in x.items()}
        if isinstance(x, list):
            return [wordy(v) for v in x]
        return x
    big = wordy(full)
    big["legs"].update({k + "_again": v for k, v in big["legs"].items()})
    big["per_rank"] = [{"rank": r, "pairs": list(range(8 * r, 8 * r + 8)), "own_work_ms_median": 20.123456789, "own_work_ms_min": 1.0, "own_work_ms_max": 2.0} for r in range(8)]
    for res in (full, big):
        line = b.compact_line(res)
        s = json.dumps(line)
        assert len(s) < 6000, len(s)
        back = json.loads(s)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in back, k
        assert back["value"] == float(f"{full['value']:.6g}") and back["config"]["workload"]
        rf = back["roofline"]
        assert rf["bound"].startswith("hbm") and rf["unit"].startswith("GB/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
        assert all(isinstance(rf[k], (int, float)) for k in ("achieved", "peak", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms"))
        assert all(isinstance(rf["alone"][k], float) for k in ("avg_launch_ms", "achieved", "frac")) and isinstance(rf["iteration"]["alone"]["frac"], float)
        cb = back["cpu_baseline"]
        assert cb["kind"].startswith("port") and cb["cores"] == 16 and cb["value"] > 0 and cb["port"]["threads_1"] > 0 and cb["kdtree"]["threads_all"] > 0 and cb["sample"]
    assert set(b.compact_line(full)["legs"]) == set(full["legs"]) and b.compact_line(full)["legs"]["config5"]["registration_ms"] > 0


def test_emit_prints_exactly_one_stdout_line(tmp_path):
    code = ("import json, sys, importlib.util; spec = importlib.util.spec_from_file_location('b', %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b);"
            "b.ROOT = %r; b.emit(json.load(open(%r)))" % (os.path.join(ROOT, "bench.py"), str(tmp_path), os.path.join(ROOT, "profiles", "r04_final_bench.json")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env={k: v for k, v in os.environ.items() if k != "M3D_BENCH_FULL_LINE"})
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and len(lines[0]) < 6000 and json.loads(lines[0])["value"] > 0
    assert json.load(open(tmp_path / "gpurun_out" / "bench_result_full.json"))["legs"]["config5"]["levels"]      # the full result is on disk (and on stderr)
    assert "[bench full result] {" in r.stderr
