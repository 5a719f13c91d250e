#!/bin/bash
# Stamps of the row-band kernel on one box:   tools/band_diag.sh LEVEL C "PLAN PLAN ..." [EVENTS]
lvl=$1; c=$2; plans=$3; ev=${4:-64}
out=gpurun_out/band_diag.log; : > $out
make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS -DSEI_STAMPS" > gpurun_out/band_build.log 2>&1 || { tail -20 gpurun_out/band_build.log; exit 1; }
for p in $plans; do
  timeout -k 10 200 python tools/band_stamps.py $lvl $c $p $ev 2>&1 | grep -v amdgpu.ids >> $out
done
make -C sparseeventid_amd/csrc -j16 -B > /dev/null 2>&1
cat $out
