#!/bin/bash
# usage: tools/power_probe.sh  (GPU box, repo root) -- clock and socket power while a kernel family runs back to back, against idle:
# is the sustained MFMA rate a power ceiling?  Reads only (rocm-smi --showclocks --showpower), changes nothing.
sample() {  # sample <label> <seconds>
  local label=$1 n=$2
  for i in $(seq 1 $n); do
    echo "$label $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power|Average Graphics Package Power" | sed -E 's/.*(sclk|Power).*:\s*//' | tr '\n' ' ')"
    sleep 0.5
  done
}
sample idle 3
python tools/bench_conv.py --dtype bf16 --iters 120 > /tmp/pp_conv.txt 2>&1 &
pid=$!
sleep 8; sample conv_family 10; wait $pid; tail -1 /tmp/pp_conv.txt | cut -c1-160
python tools/bench_conv.py --dtype bf16 --iters 120 --zeros > /tmp/pp_convz.txt 2>&1 &
pid=$!
sleep 8; sample conv_family_zero_operands 10; wait $pid; tail -1 /tmp/pp_convz.txt | cut -c1-160
python tools/bench_flrelu.py --dtype bf16 --no-bias --raw pitched --iters 150 --repeats 3 > /tmp/pp_fl.txt 2>&1 &
pid=$!
sleep 8; sample filtered_lrelu 10; wait $pid; tail -1 /tmp/pp_fl.txt | cut -c1-160
sample idle_after 2
