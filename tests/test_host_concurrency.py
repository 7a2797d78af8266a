"""The library's host-side concurrency (the per-rank worker pool + sense-reversing barrier of the row-sharded loops, the pinned
free list, the resident-chain bookkeeping: csrc/host_pool.hpp, device-free) under ThreadSanitizer and AddressSanitizer on the
CPU -- GPU sanitizers are not available on this pool, and this code only ever ran on the GPU box before (SURVEY 5 "race detection")."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_concurrency.cpp")


@pytest.mark.parametrize("san", ["thread", "address"])
def test_host_concurrency_under_sanitizer(san, tmp_path):
    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    exe = str(tmp_path / f"host_concurrency_{san}")
    extra = ["-fsanitize=undefined", "-fno-sanitize-recover=all"] if san == "address" else []
    subprocess.run([gxx, "-std=c++17", "-O1", "-g", f"-fsanitize={san}", *extra, SRC, "-o", exe, "-lpthread"], check=True, timeout=300)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([exe, "10000"], capture_output=True, text=True, timeout=600, env=env)
    log = r.stdout + r.stderr
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"host_concurrency_{san}.log"), "w") as f:
        f.write(log)
    assert r.returncode == 0, log[-4000:]
    assert "host concurrency OK" in log and "WARNING: ThreadSanitizer" not in log and "ERROR: AddressSanitizer" not in log, log[-4000:]
