#!/bin/bash
# round 5: can the level-2 dX launches (conv_round / conv_mround, 96 planes) and the dW backlog share CUs?  The shipped pair cannot:
# 144 KB of LDS + 4 x 104 registers per SIMD lane leave no room for a dW workgroup (28 KB, 114 registers).  Arms: the level-2
# rounds with 16-plane slices (120 KB: SEI_ROUND_L2_SLJ1=1) and / or conv_dw_mfma_kernel<2,6> held to 96 registers
# (-DSEI_DW_OCC=2); whole overlapped step, one box, interleaved twice.      tools/coresidency_ab.sh
out=gpurun_out/r05/coresidency_ab.log; mkdir -p gpurun_out/r05; : > $out
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1), {k:round(v['ms'],1) for k,v in d['roofline']['per_kind_per_step'].items()})"; }
for rep in 1 2; do
  for occ in "" "-DSEI_DW_OCC=2"; do
    make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS $occ" > gpurun_out/r05/coresidency_build.log 2>&1 || { tail -20 gpurun_out/r05/coresidency_build.log; exit 1; }
    for slj1 in 0 1; do
      echo "== dW build [$occ] SEI_ROUND_L2_SLJ1=$slj1" >> $out
      SEI_ROUND_L2_SLJ1=$slj1 timeout -k 10 300 python bench.py --steps 8 --warmup 3 --cpu-events 0 --batches 1 2>/dev/null | tail -1 | line >> $out
    done
  done
  echo "rep $rep done"
done
make -C sparseeventid_amd/csrc -j16 -B > gpurun_out/r05/coresidency_build.log 2>&1
cat $out
