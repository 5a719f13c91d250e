#!/bin/bash
# Ablation builds of the workgroup-tile kernel on ONE box: each variant removes a part of the phase (results are wrong,
# timing only) -- what remains when a part is gone bounds what that part costs.
#   tools/tile_abl.sh "" "-DSEI_ABL_NORMW" "-DSEI_ABL_NOB" ...
out=gpurun_out/tile_abl.log; : > $out
for v in "$@"; do
  rm -f sparseeventid_amd/csrc/conv_fwd.o
  make -C sparseeventid_amd/csrc -j16 EXTRA="-DSEI_EXPERIMENTS $v" > /dev/null 2>&1
  echo "== variant [$v]" >> $out
  timeout -k 10 200 python tools/sweep_wave.py --events 16 --levels 2 3 --only tile 2>&1 | grep -E "plan tile" >> $out
done
rm -f sparseeventid_amd/csrc/conv_fwd.o
cat $out
