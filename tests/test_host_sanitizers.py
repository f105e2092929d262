"""Host-side C-ABI entry points (pack, CPython-exact rng, sampler, shuffle) under AddressSanitizer + UBSan.

The GPU pool offers no device sanitizer, so this covers the part of the library that runs on the CPU: the driver
tests/native/host_sanitizer_driver.cpp is compiled together with opal_amd/csrc/opal_host.cpp by plain g++ (no GPU,
no HIP runtime call on these paths) with exact-size heap buffers, including the one-byte-short refusals."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_entry_points_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_san"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT}/include", f"-I{ROOT}/opal_amd/csrc",
           f"{ROOT}/tests/native/host_sanitizer_driver.cpp", f"{ROOT}/opal_amd/csrc/opal_host.cpp", "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and "hip/hip_runtime.h" in build.stderr:
        pytest.skip("HIP headers not found under /opt/rocm/include")
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, env=env)
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
    assert "host sanitizer driver: ok" in run.stdout
