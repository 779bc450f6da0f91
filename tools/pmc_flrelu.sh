#!/bin/bash
# usage: tools/pmc_flrelu.sh <outdir> <python script + args...>  -- kernel trace + two PMC passes incl. MFMA busy
out=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/$out
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out/trace -- python3 "$@" > gpurun_out/$out/trace.log 2>&1 || true
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/$out/pmc1 -- python3 "$@" > gpurun_out/$out/pmc1.log 2>&1 || true
timeout -k 10 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/$out/pmc2 -- python3 "$@" > gpurun_out/$out/pmc2.log 2>&1 || true
