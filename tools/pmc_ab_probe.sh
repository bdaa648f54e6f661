#!/bin/bash
export TMPDIR=/tmp; R=$PWD; D=$R/gpurun_out/pmc_ab; mkdir -p $D; cd /tmp
ARGS="--steps 3 --warmup 1 --cpu 0 --full-schedule 0 --spp1 0 --synthetic-kernels 0 --other-configs 0 --pmc-in-run 0 --phase-probe 0"
for v in default k1; do
  if [ $v = default ]; then unset PGSD_LIBRARY; else export PGSD_LIBRARY=$R/practical_path_guiding_lab_amd/libpgsd_$v.so; fi
  timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --kernel-include-regex "k_wave_shade" --output-format csv -d $D/$v -- python3 $R/bench.py $ARGS > /dev/null 2> $D/$v.err
  echo $v rc=$?
  timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SENDMSG --kernel-include-regex "k_wave_shade" --output-format csv -d $D/${v}_b -- python3 $R/bench.py $ARGS > /dev/null 2> $D/${v}_b.err
  echo $v rc=$?
done
python3 - <<PY
import csv, glob, collections
for v in ("default","default_b","k1","k1_b"):
  for p in sorted(glob.glob("$D/%s/**/*counter_collection.csv" % v, recursive=True)):
    acc = collections.defaultdict(float); n=0
    rows=[r for r in csv.DictReader(open(p))]
    for r in rows:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
    nd=len(set(r["Dispatch_Id"] for r in rows))
    print(v, "dispatches", nd, " ".join("%s %.4g" % (a, b/nd) for a, b in sorted(acc.items())))
PY
