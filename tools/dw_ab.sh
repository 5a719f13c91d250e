#!/bin/bash
# A/B of dW-kernel source variants on ONE box (devices differ by several %): rebuild conv.o per variant (EXTRA flags).
#   tools/dw_ab.sh "" "-DDW_DEPTH=2" ""
out=gpurun_out/dw_ab.log; : > $out
for v in "$@"; do
  rm -f sparseeventid_amd/csrc/conv.o
  make -C sparseeventid_amd/csrc -j16 EXTRA="-DSEI_EXPERIMENTS $v" > /dev/null 2>&1
  echo "== variant [$v]" >> $out
  timeout -k 10 400 python tools/dw_micro.py --events ${EVENTS:-64} --levels ${LEVELS:-0 1 2 3 4 5} 2>&1 | grep -E "^L" >> $out
done
rm -f sparseeventid_amd/csrc/conv.o
cat $out
