"""Part of the emulator suite again under AddressSanitizer.

The library sources are compiled a second time with -fsanitize=address (tests/hostemu/build_emu.py,
HOSTEMU_SANITIZE=1) and the re-entrancy tests run on that build in a subprocess that has the sanitizer runtime
preloaded: a dangling pointer inside the library (round 3's Mat* held across a callback) then stops the run with a
report instead of depending on what the heap happens to hold.  tools/emu_sanitized.sh runs the WHOLE CPU suite this
way with UBSan on top (HOSTEMU_SANITIZE=2; about half an hour; log under profiles/)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sanitized_env():
    sys.path.insert(0, os.path.join(REPO, 'tests', 'hostemu'))
    import build_emu
    rt = build_emu.asan_runtime()
    if not os.path.exists(rt):
        pytest.skip('no shared AddressSanitizer runtime next to the compiler')
    env = dict(os.environ)
    env.update(HOSTEMU_SANITIZE='1', LD_PRELOAD=rt,
               ASAN_OPTIONS='detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0:exitcode=86',
               UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    env.pop('SELLA_EMU_FULL', None)
    return env


def test_reentrancy_under_address_sanitizer():
    env = sanitized_env()
    cmd = [sys.executable, '-m', 'pytest', os.path.join(REPO, 'tests', 'test_reentrancy.py'), '-x', '-q', '-m', 'not gpu',
           '-p', 'no:cacheprovider', '-k', 'structured or error']
    r = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = r.stdout.decode(errors='replace')
    assert r.returncode == 0, out[-6000:]
    assert 'AddressSanitizer' not in out and 'runtime error' not in out, out[-6000:]
    assert ' passed' in out
