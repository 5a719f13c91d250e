#!/bin/bash
# round 5: multi-offset rounds (conv_mround_mfma_kernel) against the two-offset rounds, the tile kernel and (64 planes) the wave
# kernel, per level at 64 events, ONE experiment build on ONE box:   tools/mround_ab.sh "1:64 2:96 3:128 4:160 5:192" [EXTRA -D flags]
lv=$1; shift
out=gpurun_out/r05/mround_ab.log; mkdir -p gpurun_out/r05; : > $out
make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS $*" > gpurun_out/r05/mround_ab_build.log 2>&1 || { tail -20 gpurun_out/r05/mround_ab_build.log; exit 1; }
for lc in $lv; do
  l=${lc%%:*}; c=${lc##*:}
  plans="default rounds:SEI_ROUND_MO=2 rounds:SEI_ROUND_MO=3 rounds"
  [ "$c" = 64 ] && plans="wave $plans"
  timeout -k 10 400 python tools/sweep_tile.py --events ${SEI_AB_EVENTS:-64} --level $l --c $c --plans $plans 2>&1 | grep -E "^L|ERROR|Error|plan" >> $out
  echo "done $lc"
done
make -C sparseeventid_amd/csrc -j16 -B > gpurun_out/r05/mround_ab_build.log 2>&1
cat $out
