# fp32 step with the 3x3 convs on the native fp32 MFMA kernels and on the bf16 pipe from split operands (bench.py --fp32-conv)
for m in native f16x3 bf16x6 bf16x663 bf16x633 bf16x3; do
python bench.py --dtype fp32 --steps 6 --warmup 2 --lean --fp32-conv $m 2> gpurun_out/fp32_$m.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$m', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 1), 'ms/step', {k: round(v.get('ms_per_step', 0), 1) for k, v in d.get('kernels', {}).items()} if isinstance(d.get('kernels'), dict) else '')" || { echo "$m failed"; tail -5 gpurun_out/fp32_$m.err; }
done
