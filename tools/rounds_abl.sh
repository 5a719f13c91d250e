#!/bin/bash
# Timing-only ablations of the two-offset-round kernel at one level (results wrong): which part of a phase costs what.
#   tools/rounds_abl.sh 2 96
l=${1:-2}; c=${2:-96}
out=gpurun_out/r04/rounds_abl_L${l}_C${c}.log; mkdir -p gpurun_out/r04; : > $out
for v in "" "-DSEI_ABL_NOBAR" "-DSEI_ABLATE_GATHER" "-DSEI_ABLATE_NOGATHER" "-DSEI_ABL_NODMA" "-DSEI_ABL_NORMW" "-DSEI_ABL_NOB" "-DSEI_ABL_NOMFMA" "-DSEI_ABL_NODMA -DSEI_ABLATE_NOGATHER" "-DSEI_ABL_NODMA -DSEI_ABLATE_NOGATHER -DSEI_ABL_NOBAR"; do
  rm -f sparseeventid_amd/csrc/conv_fwd.o
  make -C sparseeventid_amd/csrc -j16 EXTRA="-DSEI_EXPERIMENTS $v" > gpurun_out/r04/rounds_abl_build.log 2>&1 || { tail -20 gpurun_out/r04/rounds_abl_build.log; exit 1; }
  echo "== [$v]" >> $out
  timeout -k 10 300 python tools/sweep_tile.py --events 64 --level $l --c $c --plans rounds 2>&1 | grep -E "^L|ERROR|Error" >> $out
done
rm -f sparseeventid_amd/csrc/conv_fwd.o
make -C sparseeventid_amd/csrc -j16 -B > gpurun_out/r04/rounds_abl_build.log 2>&1
cat $out
