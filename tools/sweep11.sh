#!/bin/bash
out=gpurun_out/sweep11.log; : > $out
run() { echo "== level $1 C=$2 plan=[$3]" >> $out; SEI_FWD_PLAN="$3" timeout -k 5 200 python tools/conv_micro.py --level $1 --cin $2 --cout $2 --events 64 --iters 5 2>&1 | grep -E "fwd:|rror" | tail -1 >> $out; }
run 5 192 "2,1,2,192,6"; run 5 192 "1,1,2,192,6"; run 5 192 "3,1,2,192,6"
run 4 160 "3,1,2,160,5"; run 4 160 "4,1,2,160,5"
run 3 128 "3,1,2,128,2"; run 3 128 "4,1,2,128,4"; run 3 128 "3,1,2,128,4"
run 2 96 "4,1,2,96,3"; run 2 96 "3,1,2,96,3"; run 2 96 "3,1,2,96,2"; run 2 96 "4,1,2,96,2"
run 1 64 ""; run 1 64 "4,1,2,64,2"; run 1 64 "3,1,2,64,2"; run 1 64 "3,1,2,64,1"
run 0 32 ""; run 0 32 "3,1,2,32,1"; run 0 32 "4,1,2,32,1"
cat $out
