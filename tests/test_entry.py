"""The driver's entry point: __graft_entry__.build() compiles libpgsd.so (hipcc cross-compiles gfx950 without a
GPU) and the oracle, imports the package and checks the ABI version of what it built.  Round 3 shipped a build()
whose last line still asserted the previous ABI version: nothing called it.  This test does."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_entry_point_runs_clean():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import __graft_entry__ as g

    g.build()  # make is incremental: seconds when nothing changed
    from practical_path_guiding_lab_amd import _native

    assert os.path.exists(_native.LIB_PATH)
    assert _native.lib().pg_abi_version() == _native.ABI_VERSION
    assert os.path.exists(os.path.join(ROOT, "oracle", "libpg_oracle.so"))


def test_smoke_refuses_to_run_without_a_gpu():
    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import __graft_entry__ as g

    with pytest.raises(RuntimeError):
        g.smoke()
