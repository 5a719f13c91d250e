#!/bin/bash
# A/B of tile-kernel source variants on ONE box (devices differ by several %): rebuild per variant (EXTRA flags).
#   tools/tile_ab.sh "" "-DSEI_TILE_NATURAL" ""
out=gpurun_out/tile_ab.log; : > $out
for v in "$@"; do
  rm -f sparseeventid_amd/csrc/conv_fwd.o sparseeventid_amd/csrc/capi.o
  make -C sparseeventid_amd/csrc -j16 EXTRA="-DSEI_EXPERIMENTS $v" > /dev/null 2>&1
  echo "== variant [$v]" >> $out
  timeout -k 10 200 python tools/sweep_wave.py --events 16 --levels 1 2 3 4 5 2>&1 | grep -E "plan (tile|4,64,3|6,64,2|8,64,3|5,64,2|4,64,2) " >> $out
done
cat $out
