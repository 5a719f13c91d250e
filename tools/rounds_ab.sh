#!/bin/bash
# tile kernel vs its two-offset-round form (conv_round_mfma_kernel) and the column-owner kernel on ONE box, experiment build:
#   tools/rounds_ab.sh "2:96 3:128 4:160" [EXTRA -D flags]
lv=$1; shift
out=gpurun_out/r04/rounds_ab.log; mkdir -p gpurun_out/r04; : > $out
make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS $*" > gpurun_out/r04/rounds_ab_build.log 2>&1 || { tail -20 gpurun_out/r04/rounds_ab_build.log; exit 1; }
for lc in $lv; do
  l=${lc%%:*}; c=${lc##*:}
  timeout -k 10 300 python tools/sweep_tile.py --events 64 --level $l --c $c --plans default rounds 2>&1 | grep -E "^L|ERROR|Error|plan" >> $out
done
make -C sparseeventid_amd/csrc -j16 -B > gpurun_out/r04/rounds_ab_build.log 2>&1
cat $out
