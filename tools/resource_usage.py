"""Writes registers, scratch bytes, occupancy and LDS of every kernel as hipcc reports them for the build
flags of csrc/Makefile (cross-compiles: no GPU needed).
    python tools/resource_usage.py profiles/r02/resource_usage.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "practical_path_guiding_lab_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt".split()
out = ["# hipcc " + " ".join(FLAGS) + " -Rpass-analysis=kernel-resource-usage, per source file of practical_path_guiding_lab_amd/csrc",
       "# kernel | VGPRs | scratch bytes per lane | occupancy (waves per SIMD) | LDS bytes per workgroup"]
for f in sorted(os.listdir(SRC)):
    if not f.endswith(".hip"):
        continue
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/tmp/ru.o"],
                       cwd=SRC, capture_output=True, text=True)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", name).replace("void ", "")}
            rows.append(cur)
            continue
        for key, pat in (("v", r"remark:\s+VGPRs: (\d+)"), ("s", r"ScratchSize \[bytes/lane\]: (\d+)"), ("o", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("l", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = m.group(1)
    if rows:
        out.append("## " + f)
        out += ["%s | %s | %s | %s | %s" % (r_["name"], r_.get("v"), r_.get("s"), r_.get("o"), r_.get("l")) for r_ in rows]
open(os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "profiles/resource_usage.txt"), "w").write("\n".join(out) + "\n")
