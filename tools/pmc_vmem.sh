#!/bin/bash
# usage: tools/pmc_vmem.sh <outdir> <python script + args...>  -- memory-pipeline issue counters (one PMC pass)
out=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/$out
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM --output-format csv -d gpurun_out/$out/pmc3 -- python3 "$@" > gpurun_out/$out/pmc3.log 2>&1 || true
timeout -k 10 240 rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d gpurun_out/$out/pmc4 -- python3 "$@" > gpurun_out/$out/pmc4.log 2>&1 || true
