#!/bin/bash
# Dev tool (GPU box, repo root): the vector-instruction mix of the mesh pipeline's kernels -- how much of their
# work is fp64 (the deterministic series), fp32 arithmetic, integer / address arithmetic.  One bounded pass.
export TMPDIR=/tmp; R=$PWD; D=$R/gpurun_out/pmc_valu; mkdir -p $D; cd /tmp
ARGS="--steps 2 --warmup 1 --cpu 0 --train-iters 4"
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 --kernel-include-regex "k_wave_" --output-format csv -d $D/a -- python3 $R/bench.py $ARGS > /dev/null 2> $D/a.err
echo rc=$?
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$D/a/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void pg::", "").split("<")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in acc.items():
        tot = c.get("SQ_INSTS_VALU", 1.0)
        print(k, "VALU %.3g" % tot, " ".join("%s %.1f%%" % (n.replace("SQ_INSTS_VALU_", ""), 100 * v / tot) for n, v in sorted(c.items()) if n != "SQ_INSTS_VALU"))
PY
