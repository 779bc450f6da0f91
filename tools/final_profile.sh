set -e
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r01e_tests.log 2>&1
python bench.py > gpurun_out/r01e_bench.json 2> gpurun_out/r01e_bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r01e_smoke.log 2>&1
bash tools/pmc_traffic.sh r01e bench.py --steps 4 --warmup 2 --lean > gpurun_out/r01e_traffic.log 2>&1
python tools/traffic_summary.py gpurun_out/r01e gpurun_out/r01e_pmc_traffic.json > gpurun_out/r01e_bench_hbm_traffic.txt 2>&1
python tools/bench_conv.py --dtype bf16 > gpurun_out/r01e_conv_layers_bf16.txt 2>&1
python tools/bench_flrelu.py --dtype bf16 --no-bias > gpurun_out/r01e_flrelu_layers_bf16.txt 2>&1
echo ALLDONE
