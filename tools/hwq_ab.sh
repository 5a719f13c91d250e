#!/bin/bash
# HIP's GPU_MAX_HW_QUEUES (default 4): the overlapped step keeps 3-4 streams busy (step, dW backlog, rulebook build, RCCL)
for r in 1 2; do
for q in default 8 2; do
if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
python bench.py --steps 10 --warmup 3 --cpu-events 0 --force-pg 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=$q', round(d['value'],1), 'events/s', round(d['ms_per_step'],1), 'ms')"
done; done
