"""The C-ABI library builds for gfx950, loads, and exports every symbol the header
declares.  No compute call is made (no GPU needed)."""
import ctypes
import os
import re

import pytest

from bore_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(_lib.LIB_PATH):
        if _lib.hipcc_path() is None:
            pytest.fail("libbore_hip.so missing and no hipcc to build it")
        _lib.build_native()
    return _lib.lib()


def header_functions():
    src = open(os.path.join(ROOT, "include", "bore_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bore_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol(built):
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(raw, name), name
    assert built.bore_abi_version() == 12


def test_library_carries_the_digest_of_its_sources(built):
    """ABI 11: the sha256 of the kernel sources is compiled into the library; `build_native` compares it with the
    tree's instead of file times (which say nothing once a tree has travelled) -- a library built from THESE sources
    is what the suite runs, and an up-to-date one is not rebuilt."""
    if _lib.LIB_PATH != _lib.DEFAULT_LIB_PATH:
        pytest.skip("an experiment build named by BORE_LIB_PATH")
    digest = _lib.source_digest()
    assert len(digest) == 64 and _lib.built_digest() == digest
    built.bore_source_digest.restype = ctypes.c_char_p
    assert built.bore_source_digest().decode() == digest
    assert _lib.build_native() == _lib.DEFAULT_LIB_PATH      # (nothing to do: returns at once)


def test_param_count_and_descriptor_validation(built):
    d = _lib.make_desc(2, [16, 16, 1], ["relu", "relu", "sigmoid"])
    assert built.bore_param_count(ctypes.byref(d)) == 337
    d = _lib.make_desc(6, [32, 32, 1], ["relu", "relu", None])
    assert built.bore_param_count(ctypes.byref(d)) == 1313
    d = _lib.make_desc(16, [64, 64, 64, 1], ["relu"] * 3 + [None])
    assert built.bore_param_count(ctypes.byref(d)) == 9473
    d = _lib.make_desc(32, [128, 128, 1], ["relu"] * 2 + [None])
    assert built.bore_param_count(ctypes.byref(d)) == 20865
    bad = _lib.MlpDesc()
    assert built.bore_param_count(ctypes.byref(bad)) < 0
    assert b"bore_mlp_desc" in built.bore_last_error()
    with pytest.raises(ValueError):
        _lib.make_desc(2, [4] * 9, ["relu"] * 9)
    with pytest.raises(ValueError):
        _lib.make_desc(2, [4, 1], ["swish", None])


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_lib.MlpDesc) == 4 * (2 + 4 * _lib.MAX_LAYERS + 1)      # + compute
    assert ctypes.sizeof(_lib.AdamCfg) == 16


def test_reading_the_built_digest_does_not_load_the_library():
    """`build()` asks for the digest before anything else of the package has run.  Opened ahead of torch, the library
    binds to the system's HIP runtime and every later call of it in that process fails with "no ROCm-capable device"
    (found on the GPU box: build() followed by smoke() in one interpreter) -- so the digest is read from the file."""
    import subprocess
    import sys
    code = ("from bore_amd import _lib; d = _lib.built_digest(); import sys; "
            "assert d is None or len(d) == 64; assert 'torch' not in sys.modules; "
            "assert 'libbore_hip' not in open('/proc/self/maps').read()")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]


def test_ctypes_structures_have_the_headers_layout(tmp_path):
    """The ctypes mirrors of the C structures (bore_amd._lib) against the header as a C compiler lays it out: sizes and
    the offsets of the fields added last (ABI 12: bore_engine_cfg's resident_wait_us / worker_streams / work_queue)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "bore_hip.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(bore_engine_cfg), '
                   'offsetof(bore_engine_cfg, low), offsetof(bore_engine_cfg, resident_wait_us), '
                   'offsetof(bore_engine_cfg, work_queue), sizeof(bore_engine_stats), sizeof(bore_lbfgsb_opts), '
                   'sizeof(bore_adam_cfg)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [ctypes.sizeof(_lib.EngineCfg), _lib.EngineCfg.low.offset, _lib.EngineCfg.resident_wait_us.offset,
            _lib.EngineCfg.work_queue.offset, ctypes.sizeof(_lib.EngineStats), ctypes.sizeof(_lib.LbfgsbOpts),
            ctypes.sizeof(_lib.AdamCfg)]
    assert got == want, (got, want)
