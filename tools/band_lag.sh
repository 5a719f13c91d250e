#!/bin/bash
# Loader look-ahead (weight slices in flight before the older one is published) of the ring kernels, one box.
#   tools/band_lag.sh "2 3" "3,8,16,1 3,8,12,1" 4 8 12
levels=$1; plans=$2; shift 2
out=gpurun_out/band_lag.log; : > $out
make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS" > /dev/null 2>&1
for v in "$@"; do
  rm -f sparseeventid_amd/csrc/conv_band.o
  make -C sparseeventid_amd/csrc -j16 EXTRA="-DSEI_EXPERIMENTS -DSEI_BAND_LAG=$v" > /dev/null 2>&1
  echo "== SEI_BAND_LAG=$v" >> $out
  timeout -k 10 300 python tools/sweep_band.py --events 64 --levels $levels --plans tile $plans --rounds 2 2>&1 | grep -v amdgpu.ids >> $out
done
make -C sparseeventid_amd/csrc -j16 -B > /dev/null 2>&1
cat $out
