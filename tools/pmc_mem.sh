#!/bin/bash
# usage: tools/pmc_mem.sh <outdir> <python script + args...>   -- memory-path counters (own passes, no tracing besides kernel-trace)
out=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/$out
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out/trace -- python3 "$@" > gpurun_out/$out/trace.log 2>&1 || true
timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/$out/pmc1 -- python3 "$@" > gpurun_out/$out/pmc1.log 2>&1 || true
timeout 240 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/$out/pmc2 -- python3 "$@" > gpurun_out/$out/pmc2.log 2>&1 || true
timeout 240 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d gpurun_out/$out/pmc3 -- python3 "$@" > gpurun_out/$out/pmc3.log 2>&1 || true
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/$out/pmc4 -- python3 "$@" > gpurun_out/$out/pmc4.log 2>&1 || true
tail -3 gpurun_out/$out/pmc1.log gpurun_out/$out/pmc2.log gpurun_out/$out/pmc3.log | cut -c1-300
