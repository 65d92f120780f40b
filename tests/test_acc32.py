"""The reduction kernels keep a thread's 29 running sums in 32-bit registers plus carry counts (mandala_mapping_amd/csrc/m3d_acc.h);
the block reduction widens them back to the int64 sums of the spec. The header is plain C++ apart from one qualifier macro: compiled here with g++ and
checked, term by term, against 64-bit sums — random terms, terms that wrap at every addition in either direction, the carry fields' full range."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_acc32_equals_int64_sums(tmp_path):
    exe = str(tmp_path / "acc32_driver")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-Wno-unknown-pragmas", "-I", os.path.join(ROOT, "mandala_mapping_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "acc32_driver.cpp"), "-o", exe], check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "0 mismatches" in r.stdout
