#!/bin/bash
# The whole CPU suite on the AddressSanitizer + UBSan build of the emulator (tests/hostemu/build_emu.py, HOSTEMU_SANITIZE=2),
# optionally with a perturbed heap as well:   tools/emu_sanitized.sh [pytest args...]   (log: profiles/${SAN_LOG:-r06_emu_sanitized.log})
set -u
cd "$(dirname "$0")/.."
RT=$(python -c "import sys; sys.path.insert(0,'tests/hostemu'); import build_emu; print(build_emu.asan_runtime())")
export HOSTEMU_SANITIZE=2 LD_PRELOAD="$RT" MALLOC_PERTURB_=165
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:exitcode=86 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_emu_sanitized.py "$@" 2>&1 | tee profiles/${SAN_LOG:-r06_emu_sanitized.log}
