"""Sanitizer build of the plain-C oracle: oracle/rvq_ref.c + its self-test under AddressSanitizer and
UndefinedBehaviorSanitizer (`make -C oracle sanitize`), so that the checker the parity tests lean on is itself memory-clean."""
import os
import shutil
import subprocess

import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


@pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("make") is None, reason="needs gcc + make")
def test_rvq_oracle_is_clean_under_asan_ubsan():
    subprocess.run(["make", "-C", ORACLE, "sanitize"], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([os.path.join(ORACLE, "_build", "rvq_selftest_san")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rvq_selftest: ok" in r.stdout
