#!/bin/bash
# two waves per workgroup (pairs of SIMDs coupled by the barrier instead of all four), 64 events
out=gpurun_out/sweep13.log; : > $out
run() { echo "== level $1 C=$2 plan=[$3]" >> $out; SEI_FWD_PLAN="$3" timeout -k 5 200 python tools/conv_micro.py --level $1 --cin $2 --cout $2 --events 64 --iters 5 2>&1 | grep -E "fwd:|rror" | tail -1 >> $out; }
run 2 96 ""; run 2 96 "4,1,2,96,2,2"; run 2 96 "4,1,2,96,3,2"; run 2 96 "2,1,2,96,2,2"; run 2 96 "2,1,2,96,1,2"; run 2 96 "3,1,2,96,2,2"
run 1 64 ""; run 1 64 "4,1,2,64,1,2"; run 1 64 "4,1,2,64,2,2"; run 1 64 "2,1,2,64,1,2"; run 1 64 "3,1,2,64,1,2"
run 3 128 ""; run 3 128 "4,1,2,128,2,2"; run 3 128 "4,1,2,128,4,2"
run 0 32 ""; run 0 32 "4,1,2,32,1,2"; run 0 32 "2,1,2,32,1,2"
cat $out
