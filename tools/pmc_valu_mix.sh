#!/bin/bash
# Dev tool (GPU box, repo root): the vector-instruction mix of the mesh pipeline's kernels -- how much of their
# work is fp64 (the deterministic series), fp32 arithmetic, integer / address arithmetic.  Two bounded passes
# (eight counters each, nothing traced beside them).
export TMPDIR=/tmp; R=$PWD; D=$R/gpurun_out/pmc_valu; mkdir -p $D; cd /tmp
ARGS="--steps 3 --warmup 1 --cpu 0 --full-schedule 0 --spp1 0 --synthetic-kernels 0 --other-configs 0 --pmc-in-run 0"
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 --kernel-include-regex "k_wave_" --output-format csv -d $D/a -- python3 $R/bench.py $ARGS > /dev/null 2> $D/a.err &&
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT --kernel-include-regex "k_wave_" --output-format csv -d $D/b -- python3 $R/bench.py $ARGS > /dev/null 2> $D/b.err
echo rc=$?
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
tot = collections.defaultdict(dict)
for sub in ("a", "b"):
    for p in sorted(glob.glob("$D/%s/**/*counter_collection.csv" % sub, recursive=True))[-1:]:
        part = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].replace("void pg::", "").replace("pg::", "").split("<")[0]
            part[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, c in part.items():
            t = c.get("SQ_INSTS_VALU", 1.0)
            tot[k][sub] = t
            for n, v in c.items():
                if n != "SQ_INSTS_VALU":
                    acc[k][n] = 100 * v / t   # share of the pass's own SQ_INSTS_VALU
for k, c in acc.items():
    print(k, "VALU %.3g" % tot[k].get("a", tot[k].get("b", 0)), " ".join("%s %.1f%%" % (n.replace("SQ_INSTS_VALU_", ""), v) for n, v in sorted(c.items())))
PY
