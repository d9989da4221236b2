"""The CPU restatement (oracle/tlc_oracle.c) under AddressSanitizer + UBSan: the golden-vector tests again, against
`make -C oracle asan`, in a child interpreter with libasan preloaded (sanitizers on the CPU build only)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    try:
        p = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    except Exception:
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.timeout(900)
def test_golden_vectors_under_asan_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip("gcc's libasan.so not found")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ, TLC_ORACLE_ASAN="1", LD_PRELOAD=asan, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q",
                          "-p", "no:cacheprovider"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    tail = res.stdout[-3000:]
    assert res.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in res.stdout and "runtime error:" not in res.stdout, tail
    assert " passed" in res.stdout, tail
