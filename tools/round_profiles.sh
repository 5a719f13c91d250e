#!/bin/bash
# The round's headline evidence on ONE box: bench line (with the CPU baseline), rocprofv3 kernel stats of the same command,
# HBM traffic (separate --pmc passes), MFMA-busy pass.   SEI_COMMIT=<sha> SEI_ROUND=r04 bash tools/round_profiles.sh
R=${SEI_ROUND:-r05}; export SEI_ROUND=$R
export TMPDIR=/tmp SEI_GEN_WORKERS=${SEI_GEN_WORKERS:-16}
mkdir -p gpurun_out/$R
python3 bench.py --steps 10 --warmup 3 2> gpurun_out/$R/${R}_bench.err | tail -1 > gpurun_out/$R/${R}_bench.json
echo "bench: $(python3 -c "import json;d=json.load(open('gpurun_out/$R/${R}_bench.json'));print(round(d['value'],1),round(d['ms_per_step'],1),round(d['roofline']['frac'],3),d['cpu_baseline']['value'])")"
rm -rf gpurun_out/$R/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/prof -- python3 bench.py --steps 5 --warmup 2 --cpu-events 0 --no-optin-leg > gpurun_out/$R/prof.log 2>&1
f=$(find gpurun_out/$R/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/$R/${R}_bench_kernel_stats.csv 2>/dev/null; rm -rf gpurun_out/$R/prof
echo "stats: $(head -4 gpurun_out/$R/${R}_bench_kernel_stats.csv | cut -c1-160)"
# the same step with every kernel ALONE on the GPU (no side streams): the per-kernel durations that describe the kernels
python3 bench.py --steps 10 --warmup 3 --cpu-events 0 --overlap-backward 0 --no-prefetch 2> gpurun_out/$R/${R}_bench_serial.err | tail -1 > gpurun_out/$R/${R}_bench_serial.json
echo "serial: $(python3 -c "import json;d=json.load(open('gpurun_out/$R/${R}_bench_serial.json'));print(round(d['value'],1),round(d['ms_per_step'],1),round(d['roofline']['frac'],3))")"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/prof -- python3 bench.py --steps 5 --warmup 2 --cpu-events 0 --no-optin-leg --overlap-backward 0 --no-prefetch > gpurun_out/$R/prof_serial.log 2>&1
f=$(find gpurun_out/$R/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/$R/${R}_bench_serial_kernel_stats.csv 2>/dev/null; rm -rf gpurun_out/$R/prof
export SEI_GEN_WORKERS=1
bash tools/traffic.sh > gpurun_out/$R/traffic.log 2>&1; cp gpurun_out/${R}_conv_fwd_hbm_traffic.json gpurun_out/$R/ 2>/dev/null
echo "traffic: $(python3 -c "import json;d=json.load(open('gpurun_out/$R/${R}_conv_fwd_hbm_traffic.json'));print(d['traffic_bytes_per_launch'], d['launches'])")"
rm -rf gpurun_out/$R/pmc
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/$R/pmc -- python3 bench.py --steps 1 --warmup 1 --cpu-events 0 --no-prof --batches 1 --no-optin-leg > gpurun_out/$R/pmc.log 2>&1
python3 tools/pmc_sum.py gpurun_out/$R/pmc conv_ > gpurun_out/$R/${R}_bench_mfma_busy.txt 2>&1; rm -rf gpurun_out/$R/pmc
echo "pmc: $(wc -l < gpurun_out/$R/${R}_bench_mfma_busy.txt) lines"
