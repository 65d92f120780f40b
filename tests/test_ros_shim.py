"""ros/gpu_6dslam_node.cpp — the one file a maintainer must compile — meets a compiler, a linker and (gpu) two sweeps, against a
stand-in for the five ROS headers it includes (tests/ros_stub/: test infrastructure, not ROS; no ROS exists in any environment of this
repository). Replaces the node of /root/reference/m3d/m3d_husky_launch/launch/m3d_husky_bringup.launch:13; consumer idiom as in
m3d_aggregator.cpp:149-186,231-254."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mandala_mapping_amd", "csrc")
SHIM = os.path.join(ROOT, "ros", "gpu_6dslam_node.cpp")
FLAGS = ["-std=c++14", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "tests", "ros_stub"), "-I", os.path.join(ROOT, "include")]


def _build(tmp_path):
    exe = str(tmp_path / "gpu_6dslam_node")
    subprocess.run(["g++", *FLAGS, SHIM, "-o", exe, "-L", CSRC, "-lm3dreg", "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"], check=True)
    return exe


def test_shim_compiles_without_warnings():
    r = subprocess.run(["g++", *FLAGS, "-fsyntax-only", SHIM], capture_output=True, text=True)
    assert r.returncode == 0 and r.stderr.strip() == "", r.stderr


def test_shim_links_against_the_library_and_fails_loudly_without_a_gpu(tmp_path):
    """Every m3dreg_* / m3dmap_* symbol the shim calls exists in libm3dreg.so; without a device the node logs FATAL and shuts down
    (same policy as the reference's drivers, encoder_node_li.cpp:60-80) — there is no CPU fallback to fall into."""
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the failure path is the CPU suite's")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "FATAL" in r.stderr and "no usable MI355X" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["scan_to_scan", "scan_to_map"])
def test_shim_registers_two_sweeps_like_the_binding(reg, tmp_path, mode):
    """PointCloud2 in, pose out through the shim's own code: two HDL-32-shaped sweeps played into the subscriber; the published pose is
    the pose the ctypes binding gets for the same pair with the same parameters (same library: bit-identical floats)."""
    from mandala_mapping_amd import abi, synth
    exe = _build(tmp_path)
    src, tgt, Tgt = synth.hdl32_pair(600, 810, 811, dx=0.2, dy=0.05, dyaw_deg=1.5)
    files = []
    for name, a in (("first", tgt), ("second", src)):
        p = str(tmp_path / (name + ".f32")); np.ascontiguousarray(a, np.float32).tofile(p); files.append(p)
    env = dict(os.environ, M3D_STUB_CLOUDS=":".join(files), M3D_STUB_PARAMS=f"mode={mode};iterations=20;map_leaf=0.05")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    poses = [[float(x) for x in l.split()[1:]] for l in r.stdout.splitlines() if l.startswith("pose ")]
    assert len(poses) == 1, (r.stdout, r.stderr)     # the first sweep only becomes the target / the map
    if mode == "scan_to_scan":
        pr = reg.default_params()             # the node is started without parameters beyond the test's ~iterations (finest level)
        assert pr.n_levels == 2
        pr.iterations[1] = 20
        R = reg.Registrar(pr)
        T, st = R.align(R.cloud(src), R.cloud(tgt))
        assert np.array_equal(np.asarray(poses[0][:3], np.float32), np.asarray(T, np.float64)[:3, 3].astype(np.float32))
    rot, tra = synth.pose_error(_pose_matrix(poses[0]), Tgt)
    assert rot < 0.1 and tra < 0.03, (rot, tra)


def _pose_matrix(p):
    x, y, z, qx, qy, qz, qw = p
    R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                  [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                  [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = [x, y, z]
    return T
