#!/bin/bash
# A/B of wave-kernel source variants on ONE box: rebuild conv_wave.o per variant (EXTRA flags), levels 0-2.
out=gpurun_out/wave_ab.log; : > $out
for v in "$@"; do
  rm -f sparseeventid_amd/csrc/conv_wave.o
  make -C sparseeventid_amd/csrc -j16 EXTRA="-DSEI_EXPERIMENTS $v" > /dev/null 2>&1
  echo "== variant [$v]" >> $out
  timeout -k 10 200 python tools/sweep_wave.py --events 16 --levels 0 1 2 2>&1 | grep -E "plan (2,128,2|4,64,3|4,64,2|6,64,2|3,128,4) " >> $out
done
cat $out
