#!/bin/bash
# Diagnostics of the workgroup-tile conv kernel on one layer at its DEFAULT plan (run on the GPU box through gpurun):
# PMC passes of the product build, then in-kernel s_memtime stamps (diagnostic build, tools/stamps.py).
#   tools/tile_diag.sh LEVEL C      e.g.  tools/tile_diag.sh 2 96
lvl=${1:-2}; c=${2:-96}
out=gpurun_out/tile_diag_L${lvl}_C${c}; mkdir -p $out
export SEI_CONV_KERNEL=tile SEI_GEN_WORKERS=1 TMPDIR=/tmp
micro="python3 tools/conv_micro.py --level $lvl --cin $c --cout $c --events 16 --iters 5"
echo "== product build (16 events)" > $out/summary.txt
timeout -k 10 200 $micro 2>&1 | grep -E "level|fwd" >> $out/summary.txt
echo "== PMC (SQ): fractions of SQ_WAVE_CYCLES / SQ_BUSY_CYCLES" >> $out/summary.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $out/pmc1 -- $micro > $out/pmc1.log 2>&1
python3 tools/pmc_sum.py $out/pmc1 conv_tile >> $out/summary.txt
echo "== PMC (LDS)" >> $out/summary.txt
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU \
  --kernel-trace --output-format csv -d $out/pmc2 -- $micro > $out/pmc2.log 2>&1
python3 tools/pmc_sum.py $out/pmc2 conv_tile >> $out/summary.txt
rm -rf $out/pmc1 $out/pmc2
unset SEI_GEN_WORKERS
echo "== stamps (diagnostic build -DSEI_STAMPS)" >> $out/summary.txt
make -C sparseeventid_amd/csrc -j16 EXTRA="-DSEI_EXPERIMENTS -DSEI_STAMPS" -B > $out/build_stamps.log 2>&1
timeout -k 10 200 python3 tools/stamps.py $lvl $c 2>&1 | grep -v amdgpu.ids >> $out/summary.txt
cat $out/summary.txt
