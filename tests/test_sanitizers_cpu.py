"""CPU-tier sanitizer run: the oracle (oracle/gsf_oracle.c) and the product's per-lane math headers
(tests/host_harness.cpp) rebuilt with -fsanitize=address,undefined, and the golden / host-math suites re-run against those
builds in a child interpreter with libasan preloaded.  GPU code cannot be sanitized on this pool; this covers the host-side C/C++."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g"]


def test_oracle_and_host_math_under_asan_ubsan(tmp_path):
    if os.environ.get("GSF_ORACLE_LIBRARY"):
        pytest.skip("already inside the sanitizer run")
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not available")
    so = tmp_path / "libgsf_oracle_san.so"
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-std=gnu11", "-ffp-contract=off", "-shared"] + SAN +
                          ["-o", str(so), os.path.join(ROOT, "oracle", "gsf_oracle.c"), "-lm"])
    env = dict(os.environ, GSF_ORACLE_LIBRARY=str(so), GSF_HARNESS_CXXFLAGS=" ".join(SAN), LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_host_math.py")], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-4000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-4000:]
    assert " passed" in out
