#!/bin/bash
# Whole-step A/B of run-time plan overrides on ONE box (experiment build: the library reads SEI_*_PLAN), product build restored.
#   tools/bench_plan_ab.sh "" "SEI_FWD_PLAN=1,1,2,96,1,8" ""
out=gpurun_out/bench_plan_ab.log; : > $out
make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS" > /dev/null 2>&1
for v in "$@"; do
  echo "== [$v]" >> $out
  env $v timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-events 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1), round(d['roofline']['frac'],3), {k:round(v['ms'],1) for k,v in d['roofline']['per_kind_per_step'].items()})" >> $out
done
make -C sparseeventid_amd/csrc -j16 -B > /dev/null 2>&1
cat $out
