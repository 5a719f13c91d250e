#!/bin/bash
# Row-band plan sweep on ONE box: experiment build (SEI_FWD/WAVE/BAND_PLAN enabled), sweep, product build restored.
#   tools/band_sweep.sh [args of tools/sweep_band.py]
out=gpurun_out/band_sweep.log; : > $out
make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS" > gpurun_out/band_build.log 2>&1 || { tail -20 gpurun_out/band_build.log; exit 1; }
timeout -k 10 900 python tools/sweep_band.py "$@" >> $out 2>&1
make -C sparseeventid_amd/csrc -j16 -B > /dev/null 2>&1
cat $out
