#!/bin/bash
# PMC passes of the dW kernel on one layer (run on the GPU box):  tools/dw_diag.sh LEVEL
lvl=${1:-2}
out=gpurun_out/dw_diag_L$lvl; mkdir -p $out
export SEI_GEN_WORKERS=1 TMPDIR=/tmp
micro="python3 tools/dw_micro.py --events 16 --levels $lvl --iters 5"
echo "== $micro" > $out/summary.txt
timeout -k 10 200 $micro 2>&1 | grep dW >> $out/summary.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $out/p1 -- $micro > $out/p1.log 2>&1
python3 tools/pmc_sum.py $out/p1 conv_dw_mfma >> $out/summary.txt
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr \
  --kernel-trace --output-format csv -d $out/p2 -- $micro > $out/p2.log 2>&1
python3 tools/pmc_sum.py $out/p2 conv_dw_mfma >> $out/summary.txt
rm -rf $out/p1 $out/p2
cat $out/summary.txt
