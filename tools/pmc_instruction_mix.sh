#!/bin/bash
export TMPDIR=/tmp; R=$PWD; D=$R/gpurun_out/pmc_f64; mkdir -p $D; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*" | sort -u > $D/avail.txt
ARGS="--steps 3 --warmup 1 --cpu 0 --full-schedule 0 --spp1 0 --synthetic-kernels 0 --other-configs 0"
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 --kernel-include-regex "k_wave_" --output-format csv -d $D/a -- python3 $R/bench.py $ARGS > /dev/null 2> $D/a.err
echo rc=$?
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$D/a/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void pg::", "").split("<")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in acc.items():
        tot = max(c["SQ_INSTS_VALU"], 1)
        print(k, " ".join("%s %.3g (%.1f%%)" % (a.replace("SQ_INSTS_VALU_", ""), b, 100 * b / tot) for a, b in sorted(c.items())))
PY
