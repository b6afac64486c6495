"""The C-ABI library builds for gfx950, loads, and exports every symbol include/sella_hip.h
declares; the product refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


@pytest.fixture(scope='module')
def hip_library():
    from sella_amd import build
    return build.build(verbose=False)


def declared_functions():
    text = open(os.path.join(REPO, 'include', 'sella_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sella_[a-z0-9_]+)\s*\(', text)) - {'sella_matvec_fn'})


def test_header_matches_python_signatures():
    from sella_amd import _lib
    assert declared_functions() == sorted(_lib.SIGNATURES)


def test_hip_library_exports_every_declared_symbol(hip_library):
    lib = ctypes.CDLL(hip_library)
    missing = [f for f in declared_functions() if not hasattr(lib, f)]
    assert not missing, missing
    lib.sella_version.restype = ctypes.c_char_p
    assert b'gfx950' in lib.sella_version()


def test_hostemu_build_exports_every_declared_symbol(emu_library):
    missing = [f for f in declared_functions() if not hasattr(emu_library, f)]
    assert not missing, missing


def test_no_cpu_fallback_without_device(hip_library):
    """On a box without a GPU the product must fail loudly instead of computing on the CPU."""
    from sella_amd import _lib, device
    _lib._set_library_for_tests(None)
    n = ctypes.c_int(0)
    _lib.lib().sella_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip('a HIP device is visible here')
    with pytest.raises(_lib.SellaHipError):
        device.Context(0)


def test_product_does_not_import_the_oracle():
    """sella_amd/ must never import or reference oracle/ or the host emulator."""
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, 'sella_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(root, f)).read()
                if re.search(r'\b(import|from)\s+oracle\b', src) or 'sella_oracle' in src \
                        or 'hostemu' in src.replace('tests/hostemu', ''):
                    bad.append(f)
    assert not bad, bad
