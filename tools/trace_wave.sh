#!/bin/bash
# Dev tool (GPU box, repo root): per-launch durations of the wave kernels of one veach-ajar bench pass.
#   bash tools/trace_wave.sh <outdir-under-gpurun_out> [bench.py arguments]
OUT=${1:-trace_wave}; shift
export TMPDIR=/tmp; R=$PWD; cd /tmp
mkdir -p $R/gpurun_out/$OUT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$OUT/trace -- python3 $R/bench.py --steps 2 --warmup 1 --cpu 0 --train-iters 4 "$@" --detail $R/gpurun_out/$OUT/bench.json > $R/gpurun_out/$OUT/bench.line 2> $R/gpurun_out/$OUT/err.txt
echo rc=$?
