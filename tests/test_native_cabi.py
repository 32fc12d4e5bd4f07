"""The C-ABI from native code: tests/native/cabi_smoke.cpp is compiled with g++ against include/*.h and linked
with libtfhe_hip.so (+ the oracle as the checker) -- the way a Rust or C++ host would use the library."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    from oracle import pyoracle
    from tfhe_rs_string_amd import build

    build.build()
    pyoracle.build()
    exe = os.path.join(ROOT, "tests", "native", "cabi_smoke")
    src = exe + ".cpp"
    lib_dir, orc_dir = os.path.join(ROOT, "tfhe_rs_string_amd"), os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", orc_dir, src, "-o", exe,
                           "-L", lib_dir, "-ltfhe_hip", "-L", orc_dir, "-loracle",
                           f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{orc_dir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return exe


def test_native_program_links_against_the_cabi():
    """CPU: the program compiles and links (every symbol it uses is exported with C linkage)"""
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_native_program_runs():
    exe = _build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cabi_smoke ok" in out.stdout
