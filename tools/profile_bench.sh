#!/bin/bash
# Profiles a bench.py run (run on the GPU box from the repo root):
#   bash tools/profile_bench.sh <outdir-under-gpurun_out> [extra bench.py arguments, e.g. --scene veach-mis]
# 1. rocprofv3 --kernel-trace --stats  (per-kernel time; the default bench line incl. the S1/S2/S3 kernels)
# 2. separate --pmc passes (FETCH_SIZE / WRITE_SIZE / L2 + atomics), never combined with tracing
# 3. FETCH_SIZE / WRITE_SIZE of the stand-alone entry points on S1 / S2 / S3
OUT=${1:-prof_final}
shift
EXTRA="$@"
export TMPDIR=/tmp
R=$PWD
D=$R/gpurun_out/$OUT
mkdir -p $D
cd /tmp
FILTER="k_bounce|k_wave_|k_process_and_splat|k_splat_list|k_finish|k_sort_"
SFILTER="k_sample|k_pdf|k_guide_bounce|k_leaf_index|pg::k_splat\("
B="--cpu 0 --full-schedule 0 --spp1 0 --other-configs 0 --pmc-in-run 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/bench.py --steps 10 --warmup 2 $B $EXTRA --detail $D/bench_under_trace.json > $D/bench_under_trace.line 2> $D/trace.err &&
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$FILTER" --output-format csv -d $D/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 $B --synthetic-kernels 0 $EXTRA > /dev/null 2> $D/pmc_fetch.err &&
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$FILTER" --output-format csv -d $D/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 $B --synthetic-kernels 0 $EXTRA > /dev/null 2> $D/pmc_write.err &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum --kernel-include-regex "$FILTER" --output-format csv -d $D/pmc_l2 -- python3 $R/bench.py --steps 3 --warmup 1 $B --synthetic-kernels 0 $EXTRA > /dev/null 2> $D/pmc_l2.err &&
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-include-regex "$FILTER" --output-format csv -d $D/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 $B --synthetic-kernels 0 $EXTRA > /dev/null 2> $D/pmc_sq.err &&
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$SFILTER" --output-format csv -d $D/pmc_syn_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --train-iters 3 $B $EXTRA > /dev/null 2> $D/pmc_syn_fetch.err &&
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$SFILTER" --output-format csv -d $D/pmc_syn_write -- python3 $R/bench.py --steps 1 --warmup 1 --train-iters 3 $B $EXTRA > /dev/null 2> $D/pmc_syn_write.err
echo rc=$?
