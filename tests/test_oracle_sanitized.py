"""The oracle's C code under AddressSanitizer + UBSan (CPU build only; GPU sanitizers are not available on
this pool).  Runs the KAT and golden-vector tests in a child process with the instrumented library."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_kats_clean_under_asan_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not available")
    lib = os.path.join(ROOT, "oracle", "libtsdf_oracle_asan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", lib])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1", TSDF_ORACLE_LIB=lib)
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "not gpu",
                        os.path.join(ROOT, "tests", "test_oracle_kat.py"), os.path.join(ROOT, "tests", "test_golden.py"),
                        os.path.join(ROOT, "tests", "test_mesh_tables.py")],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "passed" in p.stdout and "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
