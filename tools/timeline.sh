#!/bin/bash
# kernel timeline of the default (overlapped) step: per-queue busy time, overlap, and how the backward pass ends
# (SEI_TIMELINE_ARGS="--model sbnd --batch 128" for another configuration)
export TMPDIR=/tmp
rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 3 --warmup 2 --cpu-events 0 --no-prof $SEI_TIMELINE_ARGS > gpurun_out/tl/run.log 2>&1
python3 tools/timeline.py $(find gpurun_out/tl -name "*kernel_trace.csv" | head -1) | tee gpurun_out/timeline.txt
rm -rf gpurun_out/tl
