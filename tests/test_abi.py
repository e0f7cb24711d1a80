"""The C-ABI library loads and exports every symbol include/bgm_hip.h declares
(no compute calls: runs without a GPU)."""
import os
import re

import pytest

from bayesgm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built_library():
    # The .so is git-ignored: on a fresh checkout build it the same way __graft_entry__.build()
    # does (hipcc cross-compiles gfx950 without a GPU; a no-op when the library is current).
    from bayesgm_amd.csrc.build import build
    build(force=False, verbose=False)


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "bgm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bgm_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "libbgm_hip.so does not export " + n
        assert n in _lib.SYMBOLS, "ctypes binding misses " + n
    assert sorted(_lib.SYMBOLS) == names


def test_version_and_error_strings():
    lib = _lib.load()
    assert b"gfx950" in lib.bgm_version()
    assert lib.bgm_last_error() is not None


def test_struct_layouts_match_header():
    import ctypes as C
    # sizes implied by the header (int32/float fields, 8-byte pointers)
    assert C.sizeof(_lib.CausalConfig) == 4 * (1 + 4 + 1 + 4 * (1 + 8) + 3)
    a = _lib.MhArgs()
    assert C.sizeof(a) % 8 == 0
    assert _lib.MhArgs.n.offset == 24 and _lib.MhArgs.state_dev.offset == 40


def test_library_exports_nothing_of_its_own_besides_the_abi():
    """-fvisibility=hidden + BGM_API: no host helper or state struct of the library leaks into the dynamic symbol table.  (hipcc gives
    the host stubs of __global__ functions default visibility whatever the flag says; they are the only other names of the library
    left there, next to a few std:: template instantiations and the toolchain's __hip_cuid markers.)"""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    so = os.path.join(ROOT, "bayesgm_amd", "libbgm_hip.so")
    out = subprocess.run([nm, "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout
    syms = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    own = [s for s in syms if re.search(r"bgm|gx_|Gx|bnf|Bnf|bnn|Bnn|causal|egm|Egm|fit_", s) and not re.match(r"_Z(N\d+[a-z]+)?\d+[A-Za-z0-9_]*_kernel", s)]      # (N..: stubs inside a namespace, the per-format copies of the split-precision unit)
    assert sorted(own) == _declared_symbols(), sorted(set(own) - set(_declared_symbols()))


def test_abi_map_is_current():
    """ABI_MAP.md (entry point -> translation unit -> kernel headers -> Python wrappers -> tests) is generated from the header, the sources
    and the test files; it must list every declared entry point and match what scripts/gen_abi_map.py produces now."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "gen_abi_map.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
