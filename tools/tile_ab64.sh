#!/bin/bash
# A/B of tile-kernel source variants at the benchmark's size (64 events) on ONE box: tools/tile_ab64.sh LEVEL C "plans..." VARIANT...
lvl=$1; c=$2; plans=$3; shift 3
out=gpurun_out/tile_ab64.log; : > $out
for v in "$@"; do
  rm -f sparseeventid_amd/csrc/conv_fwd.o
  make -C sparseeventid_amd/csrc -j16 EXTRA="-DSEI_EXPERIMENTS $v" > /dev/null 2>&1
  echo "== variant [$v]" >> $out
  timeout -k 10 300 python tools/sweep_tile.py --events 64 --level $lvl --c $c --plans $plans 2>&1 | grep -E "^L" >> $out
done
rm -f sparseeventid_amd/csrc/conv_fwd.o
cat $out
