#!/bin/bash
# wide layers at the bench's batch size (64 events)
out=gpurun_out/sweep10.log; : > $out
run() { echo "== level $1 C=$2 plan=[$3]" >> $out; SEI_FWD_PLAN="$3" timeout -k 5 200 python tools/conv_micro.py --level $1 --cin $2 --cout $2 --events 64 --iters 5 2>&1 | grep -E "fwd:|rror" | tail -1 >> $out; }
run 3 128 "1,1,2,128,1"; run 3 128 ""; run 3 128 "1,1,2,128,2"; run 3 128 "2,1,2,128,4"
run 4 160 "1,1,2,160,1"; run 4 160 ""; run 4 160 "2,1,2,160,2"; run 4 160 "1,1,2,160,5"; run 4 160 "2,1,2,160,5"
run 5 192 "1,1,2,192,1"; run 5 192 ""; run 5 192 "2,1,2,192,3"; run 5 192 "1,1,2,192,2"; run 5 192 "1,1,2,192,4"; run 5 192 "2,1,2,192,4"
run 2 96 ""; run 2 96 "2,1,2,96,2"; run 2 96 "2,1,2,96,3"
cat $out
