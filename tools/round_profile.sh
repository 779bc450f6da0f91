#!/bin/bash
# usage: tools/round_profile.sh <tag>   (run on the GPU box, from the repo root)
# The round's evidence set: GPU tests, the bench line, smoke, a kernel trace of the bench command with HBM traffic (FETCH_SIZE /
# WRITE_SIZE, each --pmc pass in its own run, no trace domains next to --pmc), the per-launch in-step table of the filtered_lrelu
# kernels, issue / wait counters of the wave kernels, per-layer tables.  Everything lands in gpurun_out/.
set -e
tag=$1
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1 < /dev/null
echo tests done
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err < /dev/null
echo bench done
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1 < /dev/null
bash tools/r03_step_profile.sh ${tag} < /dev/null > /dev/null 2>&1 || true
cp gpurun_out/${tag}_traf/trace/*/*kernel_stats.csv gpurun_out/${tag}_bench_kernel_stats.csv 2>/dev/null || true
echo traffic done
bash tools/pmc_wave.sh ${tag}_pmcw bench.py --steps 2 --warmup 1 --lean --no-kernel-timing < /dev/null > /dev/null 2>&1 || true
python tools/pmc_wave_table.py gpurun_out/${tag}_pmcw > gpurun_out/${tag}_flrelu_pmc_wave.txt 2>&1 || true
bash tools/pmc_flrelu.sh ${tag}_pmc bench.py --steps 2 --warmup 1 --lean --no-kernel-timing < /dev/null > gpurun_out/${tag}_pmc.log 2>&1 || true
python tools/pmc_table.py gpurun_out/${tag}_pmc > gpurun_out/${tag}_bench_pmc_raw.txt 2>&1 || true
echo pmc done
python tools/bench_conv.py --dtype bf16 > gpurun_out/${tag}_conv_layers_bf16.txt 2>&1 < /dev/null || true
python tools/bench_flrelu.py --dtype bf16 --no-bias --raw pitched > gpurun_out/${tag}_flrelu_layers_bf16.txt 2>&1 < /dev/null || true
python tools/bench_flrelu.py --dtype fp32 > gpurun_out/${tag}_flrelu_layers_fp32.txt 2>&1 < /dev/null || true
# load-only replica of the wave kernels' access pattern: built here from its source (binaries are not tracked)
if [ -x /opt/rocm/bin/hipcc ]; then
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/ubench/strip_read.hip -o /tmp/strip_read.bin > /dev/null 2>&1 && timeout -k 5 120 /tmp/strip_read.bin > gpurun_out/${tag}_strip_read.txt 2>&1 || echo "strip_read: not built / failed (skipped)"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-result tools/ubench/buffer_range_probe.hip -o /tmp/buffer_range_probe.bin > /dev/null 2>&1 && timeout -k 5 60 /tmp/buffer_range_probe.bin > gpurun_out/${tag}_buffer_range_probe.txt 2>&1 || echo "buffer_range_probe: not built / failed (skipped)"
else
  echo "strip_read / buffer_range_probe: no hipcc on this box (skipped)"
fi
# fp32 step: kernel table, the split-operand modes against the native fp32 MFMA kernels, the fp32 -> 16-bit passes
bash tools/fp32_prof.sh ${tag} > gpurun_out/${tag}_fp32_step_kernels.txt 2>&1 < /dev/null || true
bash tools/fp32_split_probe.sh > gpurun_out/${tag}_fp32_split_modes.txt 2>&1 < /dev/null || true
PYTHONPATH=. python tools/bench_split16.py > gpurun_out/${tag}_split16_passes.txt 2>&1 < /dev/null || true
echo ALLDONE
