#!/bin/bash
# Ablation builds of the LDS-operand dW kernel on one box (timing only):  tools/dwr_abl.sh 0 1 2 4 5
out=gpurun_out/dwr_abl.log; : > $out
for v in "$@"; do
  rm -f sparseeventid_amd/csrc/conv.o
  make -C sparseeventid_amd/csrc -j16 EXTRA="-DSEI_DWR_ABL=$v" > /dev/null 2>&1
  echo "== SEI_DWR_ABL=$v" >> $out
  timeout -k 10 200 python tools/dw_micro.py --events 64 --levels ${LEVELS:-2 3} --variants reg ring 2>&1 | grep "^L" >> $out
done
rm -f sparseeventid_amd/csrc/conv.o; make -C sparseeventid_amd/csrc -j16 > /dev/null 2>&1
cat $out
