#!/bin/bash
# tile-size / slice sweep with the shared-remainder dealing
out=gpurun_out/sweep8.log; : > $out
for spec in "0 32" "1 64" "2 96"; do
  set -- $spec; l=$1; c=$2
  for plan in "" "1,1,2,$c" "1,1,1,$c" "2,1,1,$c" "3,1,2,$c" "4,1,2,$c"; do
    echo "== level $l C=$c plan=[$plan]" >> $out
    SEI_FWD_PLAN="$plan" timeout -k 5 120 python tools/conv_micro.py --level $l --cin $c --cout $c --events 16 --iters 10 2>&1 | grep -E "fwd:|rror" >> $out
  done
done
for spec in "3 128" "4 160"; do
  set -- $spec; l=$1; c=$2
  for plan in "" "1,1,1,$c" "2,1,2,$c" "2,1,1,$c"; do
    echo "== level $l C=$c plan=[$plan]" >> $out
    SEI_FWD_PLAN="$plan" timeout -k 5 120 python tools/conv_micro.py --level $l --cin $c --cout $c --events 16 --iters 10 2>&1 | grep -E "fwd:|rror" >> $out
  done
done
cat $out
