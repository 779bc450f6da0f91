set -e
tag=r05
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1 < /dev/null
echo tests done; tail -2 gpurun_out/${tag}_tests.log
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err < /dev/null
echo bench done
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1 < /dev/null
echo smoke done
bash tools/r03_step_profile.sh ${tag} < /dev/null > /dev/null 2>&1 || true
cp gpurun_out/${tag}_traf/trace/*/*kernel_stats.csv gpurun_out/${tag}_bench_kernel_stats.csv 2>/dev/null || true
echo traffic done
