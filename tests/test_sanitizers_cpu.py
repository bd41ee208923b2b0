"""Sanitizer job (CPU build only — GPU AddressSanitizer is not available on the pool): the oracle's C sources under
AddressSanitizer + UndefinedBehaviorSanitizer through the oracle's own CPU tests, and the host-only parts of
dsdtm_amd/host/dsdtm_host.hpp through a small self-test program built with the same sanitizers."""
import os
import shutil
import subprocess
import sys

import pytest

from dsdtm_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_oracle_sources_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("sanitizer runtimes not installed")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"], check=True)
    env = dict(os.environ, DSDTM_ORACLE_LIB=os.path.join(ROOT, "oracle", "_build", "liboracle_san.so"),
               LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_oracle_cpu.py",
                        "tests/test_pose_opt_cpu.py", "tests/test_detector_cpu.py"], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_only_parts_of_the_cpp_layer_under_asan_and_ubsan(tmp_path):
    if not _runtime("libasan.so"):
        pytest.skip("sanitizer runtimes not installed")
    lib = capi.lib_path()
    exe = str(tmp_path / "host_only_selftest")
    subprocess.run(["g++", "-O1", "-g", "-std=c++14", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-fno-omit-frame-pointer", "-o", exe, os.path.join(ROOT, "dsdtm_amd", "host", "host_only_selftest.cpp"), lib,
                    "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "host-only selftest ok" in r.stdout, r.stdout + r.stderr[-3000:]
