#!/bin/bash
# usage: tools/round_extra.sh <tag>   -- the other BASELINE configurations and dtypes, one JSON line each (GPU box, repo root)
tag=$1
out=gpurun_out/${tag}_bench_other_configs.jsonl
: > $out
python bench.py --cpu-baseline off --dtype fp16 2>/dev/null < /dev/null | tail -n 1 >> $out                       # fp16: the scored dtype (evaluation copy)
python bench.py --cpu-baseline off --batch 32 --steps 4 2>/dev/null < /dev/null | tail -n 1 >> $out               # configs[3]'s per-GPU workload
python bench.py --cpu-baseline off --res 512 --batch 8 --dtype fp16 --steps 4 2>/dev/null < /dev/null | tail -n 1 >> $out   # configs[4]'s per-GPU workload
python bench.py --cpu-baseline off --with-discriminator --steps 4 2>/dev/null < /dev/null | tail -n 1 >> $out     # row f1: full D + G iteration
python bench.py --cpu-baseline off --dtype fp32 --steps 3 --warmup 1 2>/dev/null < /dev/null | tail -n 1 >> $out  # the reference's own dtype
python bench.py --cpu-baseline off --force-dist 2>/dev/null < /dev/null | tail -n 1 >> $out                       # one-rank RCCL: bucket hooks + reduced-gradient Adam
AFCM_FLRELU_READ_ALIGNED=0 python bench.py --cpu-baseline off 2>/dev/null < /dev/null | tail -n 1 >> $out          # ablation: general sign-reading kernels
for f in $out; do python - <<PY
import json
for l in open("$f"):
    l = l.strip()
    if not l.startswith("{"): continue
    d = json.loads(l)
    k = d.get("kernels", {})
    print(f"{d['value']:8.1f} {d['unit']:10s} {d['ms_per_step']:7.1f} ms  {d['dtype']:5s} batch {d['config']['per_gpu_batch']:3d} res {d['config']['resolution']}  " + "  ".join(f"{n} {v['ms_per_step']:.1f} ms ({v['frac']:.3f})" for n, v in k.items()) + ("  [D+G]" if "FULL" in d['config']['workload'] else "") + ("  [dist]" if d['config'].get('backend') else ""))
PY
done
