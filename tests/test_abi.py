"""The C-ABI library loads without a GPU and exports every symbol include/pgsd.h declares.
No compute calls here (CPU box)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from practical_path_guiding_lab_amd import _native

    if not os.path.exists(_native.LIB_PATH):
        _native.build()
    return _native


def test_header_symbols_are_exported(native):
    hdr = open(os.path.join(ROOT, "include", "pgsd.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*)\s*(pg_[a-z_0-9]+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed from include/pgsd.h"
    assert declared == set(native.EXPORTS)
    lib = ctypes.CDLL(native.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_abi_version_and_loud_failure_without_gpu(native):
    L = native.lib()
    hdr = open(os.path.join(ROOT, "include", "pgsd.h")).read()
    declared = int(re.search(r"^#define PGSD_ABI_VERSION (\d+)", hdr, flags=re.M).group(1))
    assert L.pg_abi_version() == declared == native.ABI_VERSION == 6   # header, library and bindings agree
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: the no-device path is not reachable")
    h = ctypes.c_void_p()
    rc = L.pg_create(ctypes.byref(h), 0)
    assert rc == -5 and not h.value  # PG_ERR_NO_DEVICE
    assert b"no HIP device" in L.pg_last_error(None)
    from practical_path_guiding_lab_amd.sdtree import SDTree

    with pytest.raises(RuntimeError):
        SDTree()


def test_struct_layouts_match_header(native):
    # sizes the C side computes for the same structs (pointers 8 B, natural alignment)
    assert ctypes.sizeof(native.pg_records) == 6 * 8
    assert ctypes.sizeof(native.pg_dense_records) == 9 * 8
    assert ctypes.sizeof(native.pg_tree_sizes) == 24
    assert ctypes.sizeof(native.pg_tree_columns) == 8 + 3 * 4 + 4 + 19 * 8
    assert ctypes.sizeof(native.pg_stats) == 5 * 8 + 2 * 8 + 2 * 4 + 3 * 8 + 8 + 2 * 4
    assert ctypes.sizeof(native.pg_depth_counters) == 40
