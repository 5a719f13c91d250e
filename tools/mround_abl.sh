#!/bin/bash
# round 5: timing-only ablations of the sixteen-wave round kernels at one level (results wrong): which part of a phase costs what.
#   tools/mround_abl.sh 1 64 [plans...]      default plans: "rounds:SEI_ROUND_MO=2 rounds"
l=${1:-1}; c=${2:-64}; shift; shift
plans=${*:-"rounds:SEI_ROUND_MO=2 rounds"}
out=gpurun_out/r05/mround_abl_L${l}_C${c}.log; mkdir -p gpurun_out/r05; : > $out
make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS" > gpurun_out/r05/mround_abl_build.log 2>&1 || { tail -20 gpurun_out/r05/mround_abl_build.log; exit 1; }
for v in "" "-DSEI_ABL_NOBAR" "-DSEI_ABLATE_NOGATHER" "-DSEI_ABL_NODMA" "-DSEI_ABL_NORMW" "-DSEI_ABL_NOB" "-DSEI_ABL_NOMFMA" "-DSEI_ABL_NODMA -DSEI_ABLATE_NOGATHER -DSEI_ABL_NORMW" "-DSEI_ABL_NODMA -DSEI_ABLATE_NOGATHER -DSEI_ABL_NORMW -DSEI_ABL_NOBAR"; do
  rm -f sparseeventid_amd/csrc/conv_fwd.o
  make -C sparseeventid_amd/csrc -j16 EXTRA="-DSEI_EXPERIMENTS $v" > gpurun_out/r05/mround_abl_build.log 2>&1 || { tail -20 gpurun_out/r05/mround_abl_build.log; exit 1; }
  echo "== [$v]" >> $out
  timeout -k 10 300 python tools/sweep_tile.py --events ${SEI_AB_EVENTS:-64} --level $l --c $c --rounds 3 --plans $plans 2>&1 | grep -E "^L|ERROR|Error" >> $out
  echo "done [$v]"
done
rm -f sparseeventid_amd/csrc/conv_fwd.o
make -C sparseeventid_amd/csrc -j16 -B > gpurun_out/r05/mround_abl_build.log 2>&1
cat $out
