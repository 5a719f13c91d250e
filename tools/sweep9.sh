#!/bin/bash
# column-block sweep for the wide layers
out=gpurun_out/sweep9.log; : > $out
for spec in "3 128" "4 160" "5 192"; do
  set -- $spec; l=$1; c=$2
  for b in 1 2 3 4; do
    for maxs in 1 2; do
      if [ $(( (c / 16) % b )) -ne 0 ]; then continue; fi
      echo "== level $l C=$c blocks=$b maxs=$maxs" >> $out
      SEI_FWD_BLOCKS=$b SEI_FWD_PLAN="$maxs,1,2,$c" timeout -k 5 120 python tools/conv_micro.py --level $l --cin $c --cout $c --events 16 --iters 10 2>&1 | grep -E "fwd:|rror" | tail -1 >> $out
    done
  done
done
cat $out
