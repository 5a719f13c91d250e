#!/bin/bash
# Whole-step comparison of run-time plan overrides (environment variables the library reads per call), on ONE box.
#   tools/bench_env.sh "" "SEI_WAVE_PLAN=64:4,64,2" "SEI_WAVE_PLAN=64:4,64,4" ""
out=gpurun_out/bench_env.log; : > $out
for v in "$@"; do
  echo "== [$v]" >> $out
  env $v timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-events 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1), round(d['roofline']['frac'],3), {k:round(v['ms'],1) for k,v in d['roofline']['per_kind_per_step'].items()})" >> $out
done
cat $out
