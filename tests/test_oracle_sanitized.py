"""CPU-side checks of the checker itself.

* `make -C oracle asan` + `oracle/asan_driver`: every entry point of the C oracle and of the k-d tree baseline under
  AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers do not exist on the MI355X pool: the oracle gets them).
* the oracle's source-only cloud (what the HIP path's `M3DREG_CLOUD_SOURCE_ONLY` builds) registers to the same bits as a full cloud.
* the from-scratch k-d tree ICP of bench.py's cpu_baseline converges to the oracle's pose (it is a different algorithm: tolerance).
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from mandala_mapping_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_kdtree_baseline_under_asan_ubsan():
    if shutil.which("gcc") is None and shutil.which("cc") is None:
        pytest.skip("no C compiler")
    b = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], capture_output=True, text=True)
    if b.returncode != 0 and ("asan" in b.stderr.lower() and "cannot find" in b.stderr.lower()):
        pytest.skip("this gcc has no libasan")
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([os.path.join(ROOT, "oracle", "asan_driver")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "asan_driver: ok" in r.stdout


@pytest.mark.parametrize("metric", [abi.POINT_TO_POINT, abi.POINT_TO_PLANE])
def test_source_only_cloud_registers_to_the_same_bits(orc, metric):
    src, tgt, _ = synth.hdl32_pair(300, 51, 151, dx=0.2, dy=-0.05, dyaw_deg=1.5)
    p = abi.Params.make(leaf=(0.4, 0.2), iterations=(5, 6), max_corr_dist=(1.0, 0.5), metric=metric, normal_leaf=0.5)
    ct = orc.Cloud(p, tgt)
    full, lean = orc.Cloud(p, src), orc.Cloud(p, src, source_only=True)
    Ta, sa, ta = orc.align(p, full, ct, trace_cap=16)
    Tb, sb, tb = orc.align(p, lean, ct, trace_cap=16)
    assert np.array_equal(Ta, Tb) and np.array_equal(ta, tb)
    assert (sa.status, sa.iterations, sa.n_corr, sa.rms) == (sb.status, sb.iterations, sb.n_corr, sb.rms)


def test_kdtree_baseline_agrees_with_the_oracle_pose(orc):
    src, tgt, Tgt = synth.hdl32_pair(300, 52, 152, dx=0.15, dy=0.05, dyaw_deg=1.0)
    p = abi.Params.make(leaf=0.2, iterations=20, max_corr_dist=0.5, metric=abi.POINT_TO_PLANE, normal_leaf=0.4, eps_rot=0.0, eps_trans=0.0)
    To, so, _ = orc.align(p, orc.Cloud(p, src, source_only=True), orc.Cloud(p, tgt))
    Tk, n, ms = orc.kdtree_icp(src, tgt, abi.POINT_TO_PLANE, 0.5, 20, threads=2)
    assert n > len(src) // 3 and all(v >= 0.0 for v in ms.values())
    ro, to = synth.pose_error(To, Tgt)
    rk, tk = synth.pose_error(Tk, Tgt)
    # two algorithms (27-voxel exact NN + voxel normals vs global NN + kNN normals) on the same data: both at the truth
    assert ro < 0.2 and to < 0.03 and rk < 0.2 and tk < 0.03
    rd, td = synth.pose_error(Tk, To)
    assert rd < 0.25 and td < 0.04
