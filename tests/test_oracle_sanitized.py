"""(CPU) SURVEY section 5: the oracle under AddressSanitizer + UndefinedBehaviorSanitizer.  The golden-vector and mapping tests are
re-run in a child process whose oracle is oracle/libygz_oracle_asan.so (gcc -fsanitize=address,undefined, no recovery from UB):
an out-of-bounds read of a patch window, a signed overflow in the fixed-point paths or a misaligned access would abort the child."""
import os
import subprocess
import sys
import pytest
from conftest import ROOT


def test_golden_tests_pass_under_asan_ubsan():
    if os.environ.get("YGZ_ORACLE_VARIANT"):
        pytest.skip("already inside the sanitizer run")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan here")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libygz_oracle_asan.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               YGZ_ORACLE_VARIANT="asan")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_oracle_mapping.py"), os.path.join(ROOT, "tests", "test_oracle_bow.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
