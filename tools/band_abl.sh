#!/bin/bash
# Ablation builds of the row-band kernel on ONE box (timing only: results are wrong): each SEI_BAND_ABL bit removes a cost.
#   tools/band_abl.sh "2 3" "3,8,8 4,8,6" 0 1 2 4 8 16
levels=$1; plans=$2; shift 2
out=gpurun_out/band_abl.log; : > $out
make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS" > /dev/null 2>&1
for v in "$@"; do
  rm -f sparseeventid_amd/csrc/conv_band.o
  make -C sparseeventid_amd/csrc -j16 EXTRA="-DSEI_EXPERIMENTS -DSEI_BAND_ABL=$v" > /dev/null 2>&1
  echo "== SEI_BAND_ABL=$v" >> $out
  timeout -k 10 300 python tools/sweep_band.py --events 64 --levels $levels --plans tile $plans --rounds 1 2>&1 | grep -v amdgpu.ids >> $out
done
make -C sparseeventid_amd/csrc -j16 -B > /dev/null 2>&1
cat $out
