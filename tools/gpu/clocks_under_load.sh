#!/bin/bash
# GPU box helper: engine / memory clocks and power while the demod loop runs (rocm-smi sampled once a second)
mkdir -p gpurun_out
python bench.py --steps 30000 --warmup 10 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/clk_bench.json 2>/dev/null &
pid=$!
sleep 4
: > gpurun_out/clocks.txt
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" >> gpurun_out/clocks.txt
  echo "--" >> gpurun_out/clocks.txt
  sleep 1
done
wait $pid
echo "idle:" >> gpurun_out/clocks.txt
sleep 2
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" >> gpurun_out/clocks.txt
cat gpurun_out/clocks.txt
python3 -c "
import json; d=json.load(open('gpurun_out/clk_bench.json')); print('loop', d['steps'], 'steps', d['ms_per_step'], 'ms/step, kernel', d['roofline']['kernel_ms'])"
