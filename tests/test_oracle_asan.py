"""The CPU oracle under AddressSanitizer + UBSan (sanitizers on the CPU build only).

`make -C oracle asan` builds oracle/liboracle_asan.so from the same source; a child pytest with
libasan preloaded and SHEPSEG_ORACLE_LIB pointing at it re-runs the oracle-vs-golden tests.  Any
report aborts the child (halt_on_error, -fno-sanitize-recover)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_golden_under_asan_ubsan():
    asan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip('no libasan in this toolchain')
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s', 'asan'])
    env = dict(os.environ, LD_PRELOAD=asan, SHEPSEG_ORACLE_LIB=os.path.join(ROOT, 'oracle', 'liboracle_asan.so'),
               ASAN_OPTIONS='detect_leaks=0:halt_on_error=1:abort_on_error=1', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-p', 'no:cacheprovider',
                        os.path.join(ROOT, 'tests', 'test_oracle_golden.py'),
                        os.path.join(ROOT, 'tests', 'test_oracle_stats.py')],
                       env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert 'AddressSanitizer' not in tail and 'runtime error' not in r.stdout + r.stderr, tail
