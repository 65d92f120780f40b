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
