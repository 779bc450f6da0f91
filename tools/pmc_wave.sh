#!/bin/bash
# usage: tools/pmc_wave.sh <outdir> <python script + args...>  -- issue / wait counters of the wave filtered_lrelu kernels, three --pmc passes
# (each in its own run, no trace domains next to --pmc)
out=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/$out
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/$out/pmc1 -- python3 "$@" > gpurun_out/$out/pmc1.log 2>&1 || true
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/$out/pmc2 -- python3 "$@" > gpurun_out/$out/pmc2.log 2>&1 || true
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/$out/pmc3 -- python3 "$@" > gpurun_out/$out/pmc3.log 2>&1 || true
tail -2 gpurun_out/$out/pmc1.log gpurun_out/$out/pmc2.log gpurun_out/$out/pmc3.log | cut -c1-200
