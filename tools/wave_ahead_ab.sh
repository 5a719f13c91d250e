#!/bin/bash
# round 5: wave-private kernel at 32 input planes, both slices of the NEXT group requested before this group computes
# (-DSEI_WAVE_GROUP_AHEAD=1, the shipped form) against round 2's one-slice-ahead form (=0); level 0 at 64 events, one box.
out=gpurun_out/r05/wave_group_ahead_ab.log; mkdir -p gpurun_out/r05; : > $out
for rep in 1 2; do
  for v in 0 1; do
    make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS -DSEI_WAVE_GROUP_AHEAD=$v" > gpurun_out/r05/wave_ab_build.log 2>&1 || { tail -20 gpurun_out/r05/wave_ab_build.log; exit 1; }
    echo "== SEI_WAVE_GROUP_AHEAD=$v" >> $out
    timeout -k 10 300 python tools/sweep_tile.py --events 64 --level 0 --c 32 --plans wave 2>&1 | grep -E "^L|ERROR|Error" >> $out
  done
done
make -C sparseeventid_amd/csrc -j16 -B > gpurun_out/r05/wave_ab_build.log 2>&1
cat $out
