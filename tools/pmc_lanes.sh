#!/bin/bash
export TMPDIR=/tmp; R=$PWD; D=$R/gpurun_out/pmc_lanes; mkdir -p $D; cd /tmp
ARGS="--steps 3 --warmup 1 --cpu 0 --full-schedule 0 --spp1 0 --synthetic-kernels 0 --other-configs 0"
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-include-regex "k_wave_" --output-format csv -d $D/a -- python3 $R/bench.py $ARGS > /dev/null 2> $D/a.err
echo rc=$?
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$D/a/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void pg::", "").split("<")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in acc.items():
        print(k, " ".join("%s %.4g" % (a, b) for a, b in sorted(c.items())), "| lanes busy per VALU cycle %.1f of 64" % (c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"], 1) ))
PY
