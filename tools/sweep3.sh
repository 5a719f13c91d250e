#!/bin/bash
out=gpurun_out/sweep3.log; : > $out
run() { l=$1; c=$2; plan=$3
  if [ "$plan" = default ]; then unset SEI_FWD_PLAN; else export SEI_FWD_PLAN=$plan,$c; fi
  echo "== level $l C=$c plan=$plan" >> $out
  timeout -k 5 120 python tools/conv_micro.py --level $l --cin $c --cout $c --events 16 --iters 5 2>&1 | grep -E "fwd:|Error|error" >> $out
}
for p in default "1,1,2" "1,1,1" "2,1,1"; do run 1 64 $p; run 2 96 $p; done
for p in default "2,1,2" "1,1,2" "2,1,1"; do run 3 128 $p; done
for p in default "1,1,2"; do run 4 160 $p; run 5 192 $p; done
grep -A1 "==" $out | paste - - | awk '{print $3,$4,$5,$7,$8,$9,$10}'
