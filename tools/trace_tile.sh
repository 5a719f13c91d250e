#!/bin/bash
# Per-CU timeline of the workgroup-tile conv kernel (run on the GPU box through gpurun): builds the trace diagnostics,
# runs tools/trace_tile.py for the given "LEVEL:C" pairs at 64 events, restores the product build.
#   tools/trace_tile.sh 2:96 3:128
out=gpurun_out/r04/trace; mkdir -p $out
make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS -DSEI_TRACE" > $out/build.log 2>&1 || { tail -20 $out/build.log; exit 1; }
for lc in "$@"; do
  l=${lc%%:*}; c=${lc##*:}
  timeout -k 10 300 python3 tools/trace_tile.py $l $c 64 --dump $out/trace_L${l}_C${c}.bin > $out/trace_L${l}_C${c}.txt 2>&1
  grep -v amdgpu.ids $out/trace_L${l}_C${c}.txt
done
make -C sparseeventid_amd/csrc -j16 -B > $out/build_product.log 2>&1
