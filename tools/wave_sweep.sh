#!/bin/bash
# Wave-kernel plan sweep on ONE box (experiment build), at the benchmark's 64 events:  tools/wave_sweep.sh [args of sweep_wave.py]
out=gpurun_out/wave_sweep.log; : > $out
make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS" > gpurun_out/wave_build.log 2>&1 || { tail -20 gpurun_out/wave_build.log; exit 1; }
timeout -k 10 600 python tools/sweep_wave.py "$@" >> $out 2>&1
make -C sparseeventid_amd/csrc -j16 -B > /dev/null 2>&1
grep "^L" $out
