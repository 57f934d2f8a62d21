"""CPU-side checks of the product library: it builds for gfx950, loads, and exports every symbol of include/mrp_ll.h
(no compute calls without a GPU), and fails loudly — no CPU fallback — when no device is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from libmultirobotplanning_amd import _build
    return _build.build()


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mrp_(?:ll|hl)_[a-z_0-9]+)\s*\(", txt)))


def test_ll_header_symbols_exported(built):
    from libmultirobotplanning_amd import ll
    lib = ll.load_library()
    names = _declared("mrp_ll.h")
    assert set(names) == set(ll.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n
    assert b"gfx950" in lib.mrp_ll_version()


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from libmultirobotplanning_amd import ll
    with pytest.raises(RuntimeError):
        ll.LowLevelEngine()


def test_struct_layouts_match_header(built):
    from libmultirobotplanning_amd import ll
    # sizes the C compiler produces for the structs of mrp_ll.h on LP64
    assert ctypes.sizeof(ll.mrp_ll_options) == 32
    assert ctypes.sizeof(ll.mrp_ll_job) == 176
    assert ctypes.sizeof(ll.mrp_ll_result) == 64
    assert ctypes.sizeof(ll.mrp_ll_stats) == 208
