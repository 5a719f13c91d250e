#!/bin/bash
# A/B of tile-kernel source variants on ONE box (devices differ by several %): rebuild conv_fwd.o per variant.
out=gpurun_out/tile_ab.log; : > $out
for v in "" "-DSEI_TILE_SELECT" "-DSEI_TILE_SKIP" "-DSEI_TILE_SELECT -DSEI_TILE_SKIP" ""; do
  rm -f sparseeventid_amd/csrc/conv_fwd.o
  make -C sparseeventid_amd/csrc -j16 EXTRA="$v" > /dev/null 2>&1
  echo "== variant [$v]" >> $out
  SEI_CONV_KERNEL=tile timeout -k 10 200 python tools/sweep_wave.py --events 16 --levels 1 2 3 2>&1 | grep -E "plan tile " >> $out
done
cat $out
