#!/bin/bash
# The bench lines of the configurations beside the headline (BASELINE configs[1], [4]; SURVEY 8d's padded grid and uniform
# cloud; the RCCL path at world size 1), one box:   SEI_ROUND=r04 tools/round_configs.sh   -> gpurun_out/r04/*.json
R=${SEI_ROUND:-r05}
mkdir -p gpurun_out/$R
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" 2> gpurun_out/$R/$name.err | tail -1 > gpurun_out/$R/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('gpurun_out/$R/$name.json'));print(round(d['value'],1),'events/s',round(d['ms_per_step'],1),'ms', (d.get('roofline') or {}).get('frac'))" 2>&1)"; }
run ${R}_bench_fwd32   --mode fwd --batch 32 --steps 10 --warmup 3 --cpu-events 8
run ${R}_bench_fwd32_batchstats --mode fwd --batch 32 --steps 10 --warmup 3 --cpu-events 0 --fwd-bn train
run ${R}_bench_sbnd    --model sbnd --batch 128 --steps 5 --warmup 2 --cpu-events 0
run ${R}_bench_padded  --grid 1024 512 1280 --steps 5 --warmup 2 --cpu-events 0
run ${R}_bench_uniform --kind uniform --steps 5 --warmup 2 --cpu-events 0
run ${R}_bench_force_pg_nccl_world1 --force-pg --steps 5 --warmup 2 --cpu-events 0
