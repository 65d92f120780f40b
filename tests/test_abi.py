"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/m3dreg.h
declares, mirrors the header's structure layouts, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from mandala_mapping_amd import abi, binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "m3dreg.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(m3d(?:reg|agg|cal|map|loop)_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = binding.lib()
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/m3dreg.h but not exported"
    assert set(names) == set(binding.EXPORTS)
    assert L.m3dreg_backend_name() == b"hip-gfx950"
    assert L.m3dreg_abi_version() == abi.ABI_VERSION


def test_struct_layouts_match_header(tmp_path):
    """sizeof/offsetof as gcc sees include/m3dreg.h vs the ctypes mirror in abi.py."""
    import subprocess
    probes = [("m3dreg_params", abi.Params, ["n_levels", "leaf", "iterations", "max_corr_dist", "metric", "min_correspondences",
                                            "eps_rot", "eps_trans", "pivot_rel_tol", "plane_ratio", "normal_min_pts", "normal_leaf",
                                            "normal_min_spread"]),
              ("m3dreg_stats", abi.Stats, ["status", "iterations", "n_corr", "rms", "last_rot", "last_trans"]),
              ("m3dreg_grid_info", abi.GridInfo, ["n", "n_valid", "n_cells", "dims", "bits", "mn", "mx", "center", "leaf", "inv_leaf",
                                                 "lbound", "has_normals"]),
              ("m3dreg_pair", abi.Pair, ["source", "target", "init_T"]),
              ("m3dloop_params", abi.LoopParams, ["sig_leaf", "sig_log2_bits", "radius", "min_gap", "top_k", "min_overlap", "max_keyframes", "reserved"]),
              ("m3dloop_candidate", abi.LoopCandidate, ["source", "target", "overlap", "pop_source", "pop_target", "dist2", "init_T"])]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "m3dreg.h"', 'int main(void){']
    for cname, _, fields in probes:
        src.append(f'printf("%zu", sizeof({cname}));')
        for f in fields:
            src.append(f'printf(" %zu", offsetof({cname}, {f}));')
        src.append('printf("\\n");')
    src.append('return 0;}')
    c = tmp_path / "probe.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for (cname, ct, fields), line in zip(probes, lines):
        nums = [int(x) for x in line.split()]
        assert nums[0] == C.sizeof(ct), cname
        assert nums[1:] == [getattr(ct, f).offset for f in fields], cname
    p = binding.default_params()
    assert p.n_levels == 2 and abs(p.leaf[0] - 0.4) < 1e-7 and abs(p.leaf[1] - 0.1) < 1e-7 and p.metric == abi.POINT_TO_PLANE   # ABI 4: coarse to fine


def test_default_params_agree_with_oracle(orc):
    a, b = binding.default_params(), orc.default_params()
    assert bytes(a) == bytes(b)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(abi.M3dregError) as ei:
        binding.Registrar()
    assert ei.value.code == abi.ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mandala_mapping_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert "libm3d_oracle" not in txt and "import orc" not in txt and "from oracle" not in txt, f


def test_no_exception_crosses_the_boundary():
    """Every exported function runs inside the guard of m3dreg_api.cpp: a std::bad_alloc raised underneath comes back as
    M3DREG_ERR_OUT_OF_MEMORY, anything else as M3DREG_ERR_HIP — never as a C++ exception through ctypes (no device needed)."""
    L = binding.lib()
    assert L.m3dreg_debug_throw(0) == abi.ERR_OUT_OF_MEMORY
    assert L.m3dreg_debug_throw(1) == abi.ERR_HIP
    assert L.m3dreg_debug_fail_alloc(0) == 0


def test_a_stale_library_is_refused_also_through_M3DREG_LIB(tmp_path):
    """M3DREG_LIB is how instrumented and older builds are selected, so the ABI version is checked there too; only
    M3DREG_ALLOW_ABI_MISMATCH=1 lets a mismatch through (and says so on stderr)."""
    import subprocess
    import sys
    src = tmp_path / "stale.c"
    src.write_text("int m3dreg_abi_version(void) { return 3; }\n")
    so = tmp_path / "libstale.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    code = "from mandala_mapping_amd import binding\ntry:\n    binding.lib()\nexcept RuntimeError as e:\n    print('REFUSED', e)\nexcept AttributeError as e:\n    print('PASSED THE CHECK')\n"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict({k: v for k, v in os.environ.items() if k != "M3DREG_ALLOW_ABI_MISMATCH"}, M3DREG_LIB=str(so), PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert "REFUSED" in r.stdout and "library 3" in r.stdout, (r.stdout, r.stderr)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, M3DREG_ALLOW_ABI_MISMATCH="1"), timeout=300)
    assert "PASSED THE CHECK" in r.stdout and "M3DREG_ALLOW_ABI_MISMATCH" in r.stderr, (r.stdout, r.stderr)
