export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
mkdir -p $R/gpurun_out/pmc_trace
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --kernel-include-regex "k_wave_" --output-format csv -d $R/gpurun_out/pmc_trace/a -- python3 $R/bench.py --steps 2 --warmup 1 --cpu 0 --train-iters 4 > /dev/null 2> $R/gpurun_out/pmc_trace/a.err
echo rc=$?
