#!/bin/bash
# A/B on one device: lazy/batched rulebook build inside the forward pass vs built one batch ahead on a side stream
# (bench.py --prefetch), interleaved.  Usage (GPU box): bash tools/prefetch_ab.sh [rounds]
set -e
R=${1:-3}
mkdir -p gpurun_out
for r in $(seq 1 $R); do
  for mode in "" "--prefetch"; do
    python bench.py --steps 10 --warmup 3 --cpu-events 0 $mode 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('round $r', '${mode:-default}', round(d['value'],1), 'events/s', round(d['ms_per_step'],1), 'ms')"
  done
done
