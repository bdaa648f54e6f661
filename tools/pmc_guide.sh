#!/bin/bash
# Dev tool (run on the GPU box from the repo root): texture-path counters of k_wave_guide, to tell what
# the SD-tree kernel of a bounce waits on.  Separate --pmc passes (at most two counters of a block each),
# never combined with tracing; every pass under its own timeout.
export TMPDIR=/tmp; R=$PWD; D=$R/gpurun_out/pmc_guide; mkdir -p $D; cd /tmp
ARGS="--steps 2 --warmup 1 --cpu 0 --train-iters 4"
pass() { # name, counters...
	local n=$1; shift
	echo "pass $n: $@" >> $D/progress.log
	timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-include-regex "k_wave_guide" --output-format csv -d $D/$n -- python3 $R/bench.py $ARGS > /dev/null 2> $D/$n.err
	local rc=$?
	echo "pass $n rc=$rc" >> $D/progress.log
	return $rc
}
pass a GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum &&
pass b TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum &&
pass c TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum &&
pass d TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
echo rc=$?
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$D/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[-26:]  # the launches of the timed region (2 passes x 13 bounces)
        print(p.split("/")[-3], k, "mean per launch %.4g" % (sum(v) / len(v)), "n", len(v))
PY
