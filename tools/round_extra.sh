#!/bin/bash
# usage: tools/round_extra.sh <tag>   -- the other BASELINE configurations and dtypes, one JSON line each (GPU box, repo root).
# Every configuration keeps its stderr (<tag>_bench_other_configs.err) and reports its exit code; a failed one leaves no line behind.
tag=$1
out=gpurun_out/${tag}_bench_other_configs.jsonl
err=gpurun_out/${tag}_bench_other_configs.err
: > $out; : > $err
run() {   # run <label> <bench.py arguments...>
  local label=$1; shift
  echo "== $label: bench.py $*" >> $err
  timeout -k 10 400 python bench.py --lean "$@" 2>> $err < /dev/null | tail -n 1 > /tmp/extra_line.json
  local rc=${PIPESTATUS[0]}
  if [ $rc -eq 0 ] && head -c 1 /tmp/extra_line.json | grep -q '{'; then cat /tmp/extra_line.json >> $out; fi
  echo "$label: rc $rc"
}
run fp16 --dtype fp16                                              # fp16: the scored dtype (evaluation copy)
run batch32 --batch 32 --steps 4                                   # configs[3]'s per-GPU workload
run res512 --res 512 --batch 8 --dtype fp16 --steps 4              # configs[4]'s per-GPU workload
run d_plus_g --with-discriminator --steps 4                        # row f1: full D + G iteration
run fp32 --dtype fp32 --steps 3 --warmup 1                         # the reference's own dtype (3x3 convs: scaled float16 split operands, 3 terms)
run fp32_bf16x6 --dtype fp32 --steps 3 --warmup 1 --fp32-conv bf16x6      # ... bfloat16 parts, six terms
run fp32_native --dtype fp32 --steps 3 --warmup 1 --fp32-conv native      # ... the native fp32 MFMA kernels
run one_rank_rccl --force-dist                                     # one-rank RCCL: bucket hooks + reduced-gradient Adam
python - <<PY
import json
for l in open("$out"):
    l = l.strip()
    if not l.startswith("{"): continue
    d = json.loads(l)
    k = d.get("kernels", {})
    print(f"{d['value']:8.1f} {d['unit']:10s} {d['ms_per_step']:7.1f} ms  {d['dtype']:5s} batch {d['config']['per_gpu_batch']:3d} res {d['config']['resolution']}  " + "  ".join(f"{n} {v['ms_per_step']:.1f} ms ({v['frac']:.3f})" for n, v in k.items()) + ("  [D+G]" if "FULL" in d['config']['workload'] else "") + ("  [dist]" if d['config'].get('backend') else "") + (f"  [fp32 conv: {d['config']['fp32_conv']}]" if d['config'].get('fp32_conv') else ""))
PY
