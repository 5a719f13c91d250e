#!/bin/bash
out=gpurun_out/sweep7.log; : > $out
run() { l=$1; c=$2; plan=$3
  if [ "$plan" = default ]; then unset SEI_FWD_PLAN; else export SEI_FWD_PLAN=$plan,$c; fi
  echo "== level $l C=$c plan=$plan" >> $out
  timeout -k 5 120 python tools/conv_micro.py --level $l --cin $c --cout $c --events 16 --iters 5 2>&1 | grep -E "fwd:|Error:" >> $out
}
for p in default "3,1,2"; do run 0 32 $p; run 1 64 $p; done
paste - - < $out | cut -c1-110
