#!/bin/bash
# One box, one pass: everything profiles/r03_conv_band_*.log holds (plan sweep at 64 events, ablation builds, stamps).
mkdir -p gpurun_out/band
bash tools/band_sweep.sh --events 64 --levels 0 1 2 3 4 5 --rounds 2 > gpurun_out/band/r03_conv_band_sweep.log 2>&1
echo "sweep done"
bash tools/band_abl.sh "2 3" "3,8,8 3,4,8 4,8,6 3,8,8,1 3,8,16,1 4,8,8,1" 0 1 2 64 66 4 8 16 32 34 98 > gpurun_out/band/r03_conv_band_ablation.log 2>&1
echo "ablation done"
bash tools/band_diag.sh 2 96 "3,8,8 3,4,8 6,4,8 3,8,8,1 3,8,16,1" > gpurun_out/band/r03_conv_band_stamps.log 2>&1
echo "stamps done"
