#!/bin/bash
# Dev tool (GPU box, repo root): what the hardware ALLOCATES per kernel -- rocprofv3's VGPR_Count / LDS_Block_Size columns of a
# kernel trace -- beside what the compiler reported (csrc/*.remarks).  bash tools/vgpr_alloc.sh OUT [bench args]
export TMPDIR=/tmp; R=$PWD; D=$R/gpurun_out/$1; shift; mkdir -p $D; cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $D/t -- python3 $R/bench.py --steps 2 --warmup 1 --train-iters 3 --cpu 0 --full-schedule 0 --spp1 0 --synthetic-kernels 0 --other-configs 0 --pmc-in-run 0 --phase-probe 0 "$@" > /dev/null 2> $D/err.txt
echo rc=$?
python3 - <<PY
import csv, glob
seen = {}
for p in glob.glob("$D/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if k.startswith("void pg::") and k not in seen:
            seen[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
for k, v in sorted(seen.items()):
    print("%-70s VGPR_Count %s accum %s SGPR %s LDS_Block %s scratch %s" % (k.replace("void pg::", "")[:70], *v))
PY
