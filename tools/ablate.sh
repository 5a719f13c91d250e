#!/bin/bash
out=gpurun_out/ablate.log; : > $out
for spec in "0 32" "1 64" "2 96" "3 128"; do
  set -- $spec; l=$1; c=$2
  for ab in 0 1 2 4 8 16 3 31; do
    export SEI_ABLATE=$ab
    echo "== level $l C=$c ablate=$ab" >> $out
    timeout -k 5 120 python tools/conv_micro.py --level $l --cin $c --cout $c --events 16 --iters 5 2>&1 | grep -E "fwd:|Error|error" >> $out
  done
done
cat $out
