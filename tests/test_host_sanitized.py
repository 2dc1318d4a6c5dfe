"""The library's HOST-ONLY code under the sanitizers: tracking_sdf_amd/csrc/host_util.{hpp,cpp} -- the
staging thread pool, the PCL-cloud repack, the tracker's sample gather, the slab arithmetic, host_math.hpp through the
tsdf_host_* entry points, and the POSIX shared-memory rendezvous + fan-in of the ranks of one node with 2 and 3 rank
PROCESSES -- built WITHOUT HIP by plain g++ (tests/host/host_util_test.cpp) under -fsanitize=address,undefined and
-fsanitize=thread.  (GPU sanitizers are not available on this pool; this is the code that runs on the host's threads.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [["pool"], ["repack"], ["gather"], ["slabs"], ["math"], ["shm", "2"], ["shm", "3"]]


def _build(sanitize):
    tag = sanitize.replace(",", "_")
    out = os.path.join(ROOT, "build", "host", "host_util_test_" + tag)
    srcs = [os.path.join(ROOT, "tests", "host", "host_util_test.cpp"), os.path.join(ROOT, "tracking_sdf_amd", "csrc", "host_util.cpp")]
    deps = srcs + [os.path.join(ROOT, "tracking_sdf_amd", "csrc", h) for h in ("host_util.hpp", "host_math.hpp")] + [os.path.join(ROOT, "include", "tsdf.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    p = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fsanitize=" + sanitize,
                        "-fno-sanitize-recover=all", "-Wall", "-Wextra", "-Werror", "-o", out] + srcs + ["-pthread", "-lrt"],
                       capture_output=True, text=True, timeout=600)
    if p.returncode != 0 and ("cannot find" in p.stderr or "unrecognized" in p.stderr):
        pytest.skip("this g++ has no -fsanitize=" + sanitize)
    assert p.returncode == 0, p.stderr[-3000:]
    return out


@pytest.mark.parametrize("sanitize", ["address,undefined", "thread"])
def test_host_only_code_is_clean_under_the_sanitizers(sanitize):
    exe = _build(sanitize)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    for case in CASES:
        p = subprocess.run([exe] + case, env=env, capture_output=True, text=True, timeout=300)
        tail = (p.stdout + p.stderr)[-3000:]
        assert p.returncode == 0, f"{sanitize} {case}: rc {p.returncode}\n{tail}"
        for bad in ("ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error", "WARNING: ThreadSanitizer"):
            assert bad not in tail, f"{sanitize} {case}\n{tail}"
