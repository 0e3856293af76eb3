"""The drop-in boundary called from C++ ("none tested from C++"): tests/cpp/abi_caller.cc is compiled
with g++ against include/mtd_abi.h and linked with libmtd_hip.so — no Python, no torch in the process — and compares the
reference-shaped entry points (gpu_calculate_fourier_modes, gpu_compute_sq_forces, the grid engine) with the CPU oracle."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "abi_caller.cc")
LIB = os.path.join(ROOT, "metadynamics-plugin_amd", "lib")
ORACLE = os.path.join(ROOT, "oracle", "_build")


def build(out):
    cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"),
           "-isystem", "/opt/rocm/include", SRC, "-o", out, "-L" + LIB, "-lmtd_hip", "-L" + ORACLE, "-lmtd_ref", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + LIB, "-Wl,-rpath," + ORACLE, "-Wl,-rpath,/opt/rocm/lib"]
    return subprocess.run(cmd, capture_output=True, text=True)


def test_cpp_caller_compiles(tmp_path):
    """CPU: the C++ translation unit sees every entry point it uses with the declared signature and links"""
    if shutil.which("g++") is None or not os.path.exists(os.path.join(LIB, "libmtd_hip.so")) or not os.path.exists(os.path.join(ORACLE, "libmtd_ref.so")):
        pytest.skip("needs g++ and the built libraries (python -c 'import __graft_entry__ as g; g.build()')")
    r = build(str(tmp_path / "abi_caller"))
    assert r.returncode == 0, r.stderr
    # (a sanitizer build of the oracle — tools/asan.sh — pulls libasan into the link, and ld remarks on its tmpnam: not ours)
    ours = [l for l in r.stderr.splitlines() if "libasan" not in l and "libubsan" not in l]
    assert not any("warning" in l for l in ours), r.stderr


@pytest.mark.gpu
def test_cpp_caller_runs(tmp_path):
    r = build(str(tmp_path / "abi_caller"))
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(tmp_path / "abi_caller")], capture_output=True, text=True, timeout=120)
    print(run.stdout)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.strip().endswith("PASS")
