"""Host-only logic of liborbx (geometry builder, frame grid, archive records, generators; the matchers' host passes with the two
device drivers replaced inside the test program) under AddressSanitizer + UndefinedBehaviorSanitizer."""
import os
import subprocess

import pytest

HELPERS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers")


@pytest.mark.parametrize("name", ["host_sanitize", "host_sanitize_match"])
def test_host_logic_under_sanitizers(name):
    subprocess.check_call(["make", "-C", HELPERS, "-s", name])
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(HELPERS, name)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "0 failed expectations" in r.stdout
