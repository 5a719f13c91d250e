#!/bin/bash
# 2 ranks on ONE device (gloo): which of the overlapped defaults two processes sharing a GPU do not tolerate
port=29520
for flags in "--overlap-backward 0 --no-prefetch" "--overlap-backward 2 --no-prefetch" "--overlap-backward 0" ""; do
port=$((port+1))
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --backend gloo --one-device --steps 2 --warmup 1 --cpu-events 0 --no-prof $flags 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$flags]', round(d['value'],1), 'events/s', round(d['ms_per_step'],1), 'ms', d['data_parallel_checks']['parameters_bit_identical_across_ranks'])"
done
