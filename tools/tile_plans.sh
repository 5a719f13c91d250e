#!/bin/bash
# Workgroup-tile plan sweep (experiment build) on ONE box:   tools/tile_plans.sh LEVEL C "plan plan ..."
lvl=$1; c=$2; plans=$3
out=gpurun_out/tile_plans.log
make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS" > gpurun_out/tile_build.log 2>&1 || { tail -20 gpurun_out/tile_build.log; exit 1; }
timeout -k 10 400 python tools/sweep_tile.py --events 64 --level $lvl --c $c --rounds 3 --plans $plans 2>&1 | grep -v amdgpu.ids >> $out
make -C sparseeventid_amd/csrc -j16 -B > /dev/null 2>&1
tail -12 $out
