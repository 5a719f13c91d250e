#!/bin/bash
# round 5: A/B of compile-time schedules of the sixteen-wave round kernels at the given levels ("|"-separated flag sets).
#   FLAGS="|-DSEI_DEFER_G0|-DSEI_DEFER_G0 -DSEI_INTERLEAVE" tools/defer_ab.sh "1:64 2:96"
FLAGS=${FLAGS:-"|-DSEI_DEFER_G0|-DSEI_DEFER_G0 -DSEI_INTERLEAVE"}
out=gpurun_out/r05/defer_ab.log; mkdir -p gpurun_out/r05; : > $out
IFS='|' read -ra SETS <<< "$FLAGS"
for rep in $(seq 1 ${REPS:-2}); do
for v in "${SETS[@]}"; do
  make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS $v" > gpurun_out/r05/defer_build.log 2>&1 || { tail -20 gpurun_out/r05/defer_build.log; exit 1; }
  echo "== [$v]" >> $out
  for lc in ${1:-1:64}; do
    timeout -k 10 300 python tools/sweep_tile.py --events ${SEI_AB_EVENTS:-64} --level ${lc%%:*} --c ${lc##*:} --rounds 3 --plans ${PLANS:-rounds tile} 2>&1 | grep -E "^L|ERROR|Error" >> $out
  done
done
done
cat $out
