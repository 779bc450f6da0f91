set -e
python -m pytest tests/test_gpu_conv.py -x -q -k "split" 2>&1 | tail -15
for m in split6 split633 split3 native; do
python bench.py --dtype fp32 --steps 4 --warmup 2 --cpu-baseline off --fp32-conv $m 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$m', d['value'], d['ms_per_step'], {k: round(v.get('ms_per_step', 0), 1) for k, v in d.get('kernels', {}).items()} if isinstance(d.get('kernels'), dict) else '')"
done
