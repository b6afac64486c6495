"""TEST INFRASTRUCTURE: compile sella_amd/csrc/*.hip against the fake HIP runtime header
(tests/hostemu/include) into tests/hostemu/_build/libsella_hostemu.so with host clang++.
See tests/hostemu/README.md.  Never loaded by the product."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, 'sella_amd', 'csrc')
# HOSTEMU_SANITIZE=1: AddressSanitizer + UBSan build in a directory of its own (tests/test_emu_sanitized.py runs part of
# the suite on it in a subprocess with the sanitizer runtime preloaded).  The plain build could not see a use after
# free inside the library (round 3: a Mat* held across a re-entering callback).
# HOSTEMU_SANITIZE=1: AddressSanitizer only, -O1 (what the test in the CPU suite builds: a quarter of the compile time);
# HOSTEMU_SANITIZE=2: AddressSanitizer + UBSan at -O2 (tools/emu_sanitized.sh, the whole suite).
SANITIZE = {'1': 1, '2': 2}.get(os.environ.get('HOSTEMU_SANITIZE', ''), 0)
OUT = os.path.join(HERE, {0: '_build', 1: '_build_asan', 2: '_build_asan_ubsan'}[SANITIZE])
LIB = os.path.join(OUT, 'libsella_hostemu.so')
CXX = os.environ.get('HOSTEMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')
FLAGS = ['-x', 'c++', '-std=c++17', '-O2', '-g', '-fPIC', '-I', os.path.join(HERE, 'include'),
         '-Wall', '-Wno-unused-function', '-Wno-unused-result', '-Wno-unknown-pragmas',
         '-Wno-pass-failed']
SAN_FLAGS = (['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer', '-shared-libasan',
              '-fno-sanitize=vptr,function'] if SANITIZE == 2 else
             ['-fsanitize=address', '-fno-omit-frame-pointer', '-shared-libasan'])
if SANITIZE:
    FLAGS = [f for f in FLAGS if f not in ('-O2', '-g')] + (['-O2', '-g'] if SANITIZE == 2 else ['-O1', '-gline-tables-only'])
    FLAGS = FLAGS + SAN_FLAGS


def asan_runtime():
    """Path of the shared AddressSanitizer runtime of the compiler (to LD_PRELOAD into the python that loads the library)."""
    out = subprocess.check_output([CXX, '-print-file-name=libclang_rt.asan-x86_64.so']).decode().strip()
    return os.path.realpath(out)


def build(force=False, verbose=False):
    """Build (if stale) and return the library path; an exclusive file lock serialises concurrent callers (pytest-xdist
    workers, ensemble worker processes of the CPU tests)."""
    import fcntl
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.hip')]
    srcs.append(os.path.join(HERE, 'hostemu.cpp'))
    hooks = os.path.join(HERE, 'hooks.cpp')
    if os.path.exists(hooks):
        srcs.append(hooks)
    deps = list(srcs) + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    deps += [os.path.join(HERE, 'include', 'hip', 'hip_runtime.h'),
             os.path.join(REPO, 'include', 'sella_hip.h')]
    newest = max(os.path.getmtime(d) for d in deps)
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + '.o')
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) > newest:
            continue
        cmd = [CXX, *FLAGS, '-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    bad = False
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            bad = True
            sys.stderr.write(f'--- {s} ---\n{out}\n')
        elif verbose and out.strip():
            print(out)
    if bad:
        raise RuntimeError('hostemu build failed')
    if procs or not os.path.exists(LIB):
        subprocess.check_call([CXX, '-shared', '-fPIC', *(SAN_FLAGS if SANITIZE else []), *objs, '-o', LIB])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
