#!/bin/bash
# fewer barriers per offset: longer weight slices (SLJ) on column blocks, 64 events
out=gpurun_out/sweep12.log; : > $out
run() { echo "== level $1 C=$2 plan=[$3]" >> $out; SEI_FWD_PLAN="$3" timeout -k 5 200 python tools/conv_micro.py --level $1 --cin $2 --cout $2 --events 64 --iters 5 2>&1 | grep -E "fwd:|rror" | tail -1 >> $out; }
run 2 96 ""; run 2 96 "2,1,6,96,2"; run 2 96 "2,1,3,96,2"; run 2 96 "2,1,6,96,3"; run 2 96 "2,1,3,96,3"
run 1 64 ""; run 1 64 "2,1,4,64,2"; run 1 64 "2,1,4,64,1"
run 3 128 ""; run 3 128 "2,1,4,128,2"; run 3 128 "2,1,4,128,4"; run 3 128 "2,1,8,128,2"
run 4 160 ""; run 4 160 "2,1,5,160,5"
cat $out
