"""AddressSanitizer + UBSan over the host optimiser of the product library (CPU build of csrc/optim.hip as plain C++).
GPU AddressSanitizer is not available on the pool, so the device code is covered by parity tests only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not found")
def test_bobyqa_and_fd_hessian_under_asan_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "glmmrmcml_amd", "csrc")
    exe = str(tmp_path / "host_asan_driver")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + csrc,
           "-I" + os.path.join(ROOT, "include"),
           "-x", "c++", os.path.join(csrc, "optim.hip"), "-x", "c++", os.path.join(csrc, "common.hip"),
           "-x", "c++", os.path.join(ROOT, "tests", "host_asan_driver.cpp"),
           "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "fails=0" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not found")
def test_band_plan_decomposition_under_asan_ubsan(tmp_path):
    """the work decomposition of the banded HMC products (csrc/band_plan.h: paired, or the streamed split-K cut with
    its fixed-order second stage) is host logic: coverage / ordering / slot / balance invariants over many shapes"""
    csrc = os.path.join(ROOT, "glmmrmcml_amd", "csrc")
    exe = str(tmp_path / "host_bandplan_driver")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + csrc,
           "-I" + os.path.join(ROOT, "include"),
           "-x", "c++", os.path.join(csrc, "common.hip"), "-x", "c++", os.path.join(ROOT, "tests", "host_bandplan_driver.cpp"),
           "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "fails=0" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
