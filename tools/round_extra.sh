#!/bin/bash
# usage: tools/round_extra.sh <tag>  (GPU box): smoke, the other BASELINE configurations, the D + G iteration, the 2-rank rehearsal
tag=$1
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"
python bench.py --steps 6 --warmup 2 --cpu-baseline off --batch 32 > gpurun_out/${tag}_bench_b32.json 2> gpurun_out/${tag}_extra.err; echo "b32 rc=$?"
python bench.py --steps 6 --warmup 2 --cpu-baseline off --res 512 --dtype fp16 --batch 8 > gpurun_out/${tag}_bench_512.json 2>> gpurun_out/${tag}_extra.err; echo "512 rc=$?"
python bench.py --steps 4 --warmup 2 --cpu-baseline off --with-discriminator > gpurun_out/${tag}_bench_d.json 2>> gpurun_out/${tag}_extra.err; echo "D rc=$?"
python bench.py --steps 6 --warmup 2 --cpu-baseline off --force-dist > gpurun_out/${tag}_bench_dist1.json 2>> gpurun_out/${tag}_extra.err; echo "dist1 rc=$?"
AFCM_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 3 --warmup 2 --cpu-baseline off > gpurun_out/${tag}_rehearse2.log 2>&1; echo "rehearse rc=$?"
echo EXTRADONE
