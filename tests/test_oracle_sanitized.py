"""The oracle's C restatement under AddressSanitizer + UndefinedBehaviorSanitizer, CPU only.

The golden-vector suite of the oracle is re-run in a child process whose oracle library is the -fsanitize=address,undefined
build (oracle/Makefile) with libasan preloaded: an out-of-bounds read or write, a signed overflow or a misaligned access in
any oracle routine on any fixture aborts the child.  The GPU box never runs this (sanitizers are CPU-only on this pool)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    try:
        path = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    except (OSError, subprocess.CalledProcessError):
        return None
    return path if os.path.isabs(path) and os.path.exists(path) else None


@pytest.mark.skipif(_libasan() is None, reason="gcc's libasan is not installed")
def test_oracle_golden_suite_under_asan_ubsan():
    env = dict(os.environ, SAME_ORACLE_SANITIZE="1", LD_PRELOAD=_libasan(),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "passed" in p.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail
    # the child really ran the instrumented library
    q = subprocess.run([sys.executable, "-c", "from oracle import same_oracle as o; print(o.lib()._name)"], env=env,
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert q.returncode == 0 and q.stdout.strip().endswith("libsame_oracle_asan.so"), q.stdout + q.stderr
