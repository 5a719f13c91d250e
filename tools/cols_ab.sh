#!/bin/bash
# A/B of column-owner kernel (conv_cols.hip) source variants against the workgroup-tile kernel on ONE box:
#   tools/cols_ab.sh "2:96 3:128" VARIANT...     (VARIANT = extra -D flags, "" = the tree as it is)
# Every variant is built once (full library, -DSEI_EXPERIMENTS); the product build is restored at the end.
lv=$1; shift
out=gpurun_out/r04/cols_ab.log; mkdir -p gpurun_out/r04; : > $out
for v in "$@"; do
  make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS $v" > gpurun_out/r04/cols_ab_build.log 2>&1 || { tail -20 gpurun_out/r04/cols_ab_build.log; exit 1; }
  echo "== variant [$v]" >> $out
  for lc in $lv; do
    l=${lc%%:*}; c=${lc##*:}
    timeout -k 10 300 python tools/sweep_tile.py --events 64 --level $l --c $c --plans default cols 2>&1 | grep -E "^L|ERROR|Error" >> $out
  done
done
make -C sparseeventid_amd/csrc -j16 -B > gpurun_out/r04/cols_ab_build.log 2>&1
cat $out
