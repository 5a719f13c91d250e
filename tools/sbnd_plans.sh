#!/bin/bash
# experiment build: tile-kernel plans for the 224-plane layers of the SBND network (default: 7 column blocks of 32 columns),
# whole-step SBND bench per plan (SEI_FWD_PLAN=maxs,cs,slj,nOut,blocks,waves)
make -C sparseeventid_amd/csrc -j16 -B EXTRA=-DSEI_EXPERIMENTS > /dev/null 2>&1
for plan in "" "1,1,1,224,2,4" "1,1,2,224,2,4" "1,1,1,224,1,4" "1,1,2,224,1,4" "1,1,1,224,7,8" "2,1,2,224,7,4" ""; do
  SEI_FWD_PLAN="$plan" python bench.py --model sbnd --batch 128 --steps 8 --warmup 3 --cpu-events 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('plan [$plan]', round(d['value'],1), 'events/s', round(d['ms_per_step'],1), 'ms')
except Exception as e: print('plan [$plan] failed')"
done
make -C sparseeventid_amd/csrc -j16 -B > /dev/null 2>&1
