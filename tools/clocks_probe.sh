#!/bin/bash
# samples rocm-smi (clock, power) every 2 s while bench.py runs; prints the samples with the highest power
python bench.py --steps 60 --warmup 3 --cpu-events 0 > gpurun_out/clocks_bench.json 2>/dev/null &
pid=$!
: > gpurun_out/clocks_samples.txt
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*: //' | tr '\n' ' ' >> gpurun_out/clocks_samples.txt; echo >> gpurun_out/clocks_samples.txt
  sleep 2
done
sort -t' ' -k2 -n -r gpurun_out/clocks_samples.txt | head -8
tail -1 gpurun_out/clocks_bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1))"
