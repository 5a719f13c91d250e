#!/bin/bash
# peak allocator figures of the bench step at 5 and 30 steps (growth check of the overlapped default) and of the serial step
for args in "--steps 5" "--steps 30" "--steps 5 --overlap-backward 0 --no-prefetch" "--steps 5 --overlap-backward 0"; do
python bench.py --warmup 3 --cpu-events 0 $args 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$args', round(d['value'],1), 'events/s', round(d['ms_per_step'],1), 'ms', d['hbm_peak_gb'])"
done
