"""What the compiler made of the kernels (no GPU needed: the build writes hipcc's -Rpass-analysis=kernel-resource-usage report of
every translation unit beside its object, csrc/*.remarks).  Two budgets are load-bearing and one register wide:

  * no kernel of the library uses scratch memory (a spilled register or a call frame is a memory round trip in the inner loops);
  * the occupancies the design rests on: k_wave_shade at five waves per SIMD (96 registers: at 97 the kernel loses a wave and
    3 ms of a 50 ms step, profiles/r05/ab_guide_grid_head_rejected.txt), the closest-hit and shadow-ray kernels at seven, the
    fused bounce of quad scenes at six (diffuse) and five (rough conductors).
"""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "practical_path_guiding_lab_amd", "csrc")


@pytest.fixture(scope="module")
def kernels():
    files = sorted(glob.glob(os.path.join(CSRC, "*.remarks")))
    if len(files) < 8:   # a library built by an older Makefile: rebuild (hipcc cross-compiles without a GPU)
        subprocess.run(["make", "-C", CSRC, "-B", "-j4"], check=True, capture_output=True)
        files = sorted(glob.glob(os.path.join(CSRC, "*.remarks")))
    out = {}
    for f in files:
        cur = None
        for line in open(f, errors="replace"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = out.setdefault(m.group(1), {"file": os.path.basename(f)})
                continue
            for key, pat in (("vgprs", r"remark:\s+VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, line)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
    names = subprocess.run(["c++filt"] + list(out), capture_output=True, text=True).stdout.split("\n")
    return {re.sub(r"\(.*", "", n).replace("void ", ""): v for n, v in zip(names, out.values())}


def test_no_kernel_of_the_library_uses_scratch(kernels):
    assert len(kernels) > 40
    bad = {k: v["scratch"] for k, v in kernels.items() if v.get("scratch", 0) != 0}
    assert not bad, bad


def test_the_occupancies_the_design_rests_on(kernels):
    def occ(prefix):
        hits = {k: v["occupancy"] for k, v in kernels.items() if k.startswith(prefix)}
        assert hits, prefix
        return hits

    for prefix, least in (("pg::k_wave_shade<", 5), ("pg::k_wave_shade_l3<", 5), ("pg::k_wave_trace<", 7), ("pg::k_wave_cast<", 7),
                          ("pg::k_bounce<true, 0>", 6), ("pg::k_bounce<false, 0>", 6), ("pg::k_bounce<true, 1>", 5),
                          ("pg::k_bounce<false, 1>", 5), ("pg::k_splat_list", 7), ("pg::k_wave_guide", 8)):
        for k, o in occ(prefix).items():
            assert o >= least, (k, o, least, kernels[k])
