#!/bin/bash
# dev tool: SQ/TCP counters for k_guide_bounce (run on the GPU box from the repo root)
export TMPDIR=/tmp; R=$PWD; cd /tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-include-regex "k_guide_bounce" --output-format csv -d $R/gpurun_out/pmcb/$name -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-passes 0 --no-compaction > /dev/null 2> $R/gpurun_out/pmcb/$name.err; }
mkdir -p $R/gpurun_out/pmcb
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD &&
run b SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE &&
run c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum &&
run d TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TA_BUSY_avr TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
echo rc=$?
