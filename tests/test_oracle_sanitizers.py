"""The oracle's translation unit under AddressSanitizer + UndefinedBehaviorSanitizer
(`make -C oracle asan`): every task, the 22-robot scrum with every command mode, coincident bodies
with and without physics.  Out-of-bounds indexing, signed overflow or an invalid float -> int
conversion in the code the parity tests trust aborts the run; non-finite states fail it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_selftest_is_clean_under_asan_and_ubsan():
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    res = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "oracle selftest: 0 non-finite states" in res.stdout
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr
