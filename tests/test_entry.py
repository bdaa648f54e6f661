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
    # the probe build -- the one compile-time switch left in csrc/ (-DPG_SHADE_PHASES=1, Makefile target `probe`) -- is built by the
    # same entry point, exports the same ABI, and is a different library from the product's
    import ctypes
    probe = os.path.join(os.path.dirname(_native.LIB_PATH), "libpgsd_phases.so")
    assert os.path.exists(probe) and os.path.getmtime(probe) >= os.path.getmtime(os.path.join(_native.CSRC, "pg_render_wave.hip"))
    P = ctypes.CDLL(probe)
    assert P.pg_abi_version() == _native.ABI_VERSION and all(hasattr(P, n) for n in _native.EXPORTS)
    assert open(probe, "rb").read() != open(_native.LIB_PATH, "rb").read()


def test_the_product_sources_carry_one_compile_time_switch():
    """round 5 left ~45 A/B switches in the product kernels, only the default combination of which was ever
    tested.  They are resolved to their defaults now (git history and profiles/ keep the variants); what is left is
    PG_SHADE_PHASES, the probe build that test_build_entry_point_runs_clean builds.  A new switch needs a test that builds it
    -- and an entry in this list."""
    import glob
    import re
    allowed = {"PG_SHADE_PHASES"}
    found = set()
    for f in glob.glob(os.path.join(ROOT, "practical_path_guiding_lab_amd", "csrc", "*.h*")):
        for line in open(f, errors="replace"):
            m = re.match(r"\s*#\s*(?:if|ifdef|ifndef|elif)\b(.*)", line)
            if m:
                found |= set(re.findall(r"\bPG_[A-Z0-9_]+\b", m.group(1)))
    assert found == allowed, sorted(found)


def test_smoke_refuses_to_run_without_a_gpu():
    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import __graft_entry__ as g

    with pytest.raises(RuntimeError):
        g.smoke()
