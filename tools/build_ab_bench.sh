#!/bin/bash
# whole-step A/B of two builds of the library on one box: FLAGS="-DA|-DB" tools/build_ab_bench.sh  (serial and overlapped step, no opt-in leg)
out=gpurun_out/r05/build_ab_bench.log; mkdir -p gpurun_out/r05; : > $out
IFS='|' read -ra SETS <<< "$FLAGS"
for rep in 1 2; do
for v in "${SETS[@]}"; do
  make -C sparseeventid_amd/csrc -j16 -B EXTRA="$v" > gpurun_out/r05/build_ab.log 2>&1 || { tail -20 gpurun_out/r05/build_ab.log; exit 1; }
  s=$(python3 bench.py --steps 10 --warmup 3 --cpu-events 0 --no-optin-leg --overlap-backward 0 --no-prefetch 2>/dev/null | tail -1 | python3 -c "import json,sys;d=json.load(sys.stdin);print(round(d['ms_per_step'],1), round(d['roofline']['hbm_view_32plane_layers']['frac'],3))")
  o=$(python3 bench.py --steps 10 --warmup 3 --cpu-events 0 --no-optin-leg 2>/dev/null | tail -1 | python3 -c "import json,sys;d=json.load(sys.stdin);print(round(d['ms_per_step'],1), round(d['roofline']['hbm_view_32plane_layers']['frac'],3))")
  echo "[$v] serial ms, 32-plane hbm frac: $s | overlapped: $o" >> $out
done
done
make -C sparseeventid_amd/csrc -j16 -B > gpurun_out/r05/build_ab.log 2>&1
cat $out
