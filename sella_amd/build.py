"""Build libsella_hip.so (HIP, gfx950) in-tree with hipcc.

    python -m sella_amd.build            # incremental
    python -m sella_amd.build --force

The shared object is written next to this file (sella_amd/libsella_hip.so); it is
git-ignored but travels to the GPU box with the repository snapshot.
"""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, '_obj')
LIB = os.path.join(HERE, 'libsella_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
# Host code (the k x k algebra of the Davidson loop, the secular solves of the step families, deflation planning) is
# compiled for AVX2 + FMA hosts: every MI355X platform ships with a CPU that has them.  SELLA_HOST_MARCH='' builds for
# the baseline x86-64 instead.
HOST_MARCH = os.environ.get('SELLA_HOST_MARCH', 'x86-64-v3')
FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-result']
if HOST_MARCH:
    # no fused multiply-add contraction on the host: the host arithmetic stays what the baseline build computes
    FLAGS += ['-Xarch_host', f'-march={HOST_MARCH}', '-Xarch_host', '-ffp-contract=off']


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _newest_header():
    t = os.path.getmtime(os.path.join(HERE, '..', 'include', 'sella_hip.h'))
    for f in os.listdir(CSRC):
        if f.endswith('.h'):
            t = max(t, os.path.getmtime(os.path.join(CSRC, f)))
    return t


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr = _newest_header()
    objs = []
    procs = []
    for src in sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        objs.append(o)
        if (not force and os.path.exists(o)
                and os.path.getmtime(o) > max(os.path.getmtime(s), hdr)):
            continue
        cmd = [HIPCC, *FLAGS, '-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f'--- {src} failed ---\n{out}\n')
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError('hipcc failed')
    if (force or procs or not os.path.exists(LIB)
            or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)):
        cmd = [HIPCC, '-shared', '-fPIC', f'--offload-arch={ARCH}', *objs, '-o', LIB]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    t0 = time.time()
    build(force='--force' in sys.argv)
    print(f'built {LIB} in {time.time() - t0:.1f}s')
