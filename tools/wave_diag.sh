#!/bin/bash
# Diagnostics of the wave conv kernel on one layer (kernel work; run on the GPU box through gpurun):
#   PMC pass of the product build, in-kernel stamps (diagnostic build), ablation builds.
#   tools/wave_diag.sh LEVEL C PLAN   e.g.  tools/wave_diag.sh 2 96 6,64,2
lvl=${1:-2}; c=${2:-96}; plan=${3:-6,64,2}
out=gpurun_out/wave_diag_L${lvl}_C${c}; mkdir -p $out
export SEI_CONV_KERNEL=wave SEI_WAVE_PLAN=$c:$plan SEI_GEN_WORKERS=1 TMPDIR=/tmp
micro="python3 tools/conv_micro.py --level $lvl --cin $c --cout $c --events 16 --iters 5"
echo "== product build" > $out/summary.txt
timeout -k 10 200 $micro 2>&1 | grep -E "level|fwd" >> $out/summary.txt
echo "== PMC (SQ)" >> $out/summary.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $out/pmc1 -- $micro > $out/pmc1.log 2>&1
python3 tools/pmc_sum.py $out/pmc1 conv_wave >> $out/summary.txt
echo "== PMC (memory path)" >> $out/summary.txt
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA \
  --kernel-trace --output-format csv -d $out/pmc2 -- $micro > $out/pmc2.log 2>&1
python3 tools/pmc_sum.py $out/pmc2 conv_wave >> $out/summary.txt
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr \
  --kernel-trace --output-format csv -d $out/pmc3 -- $micro > $out/pmc3.log 2>&1
python3 tools/pmc_sum.py $out/pmc3 conv_wave >> $out/summary.txt
rm -rf $out/pmc1 $out/pmc2 $out/pmc3
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TA|TCP|TCC|GRBM)_[A-Za-z0-9_]+" | sort -u > gpurun_out/pmc_counters.txt
unset SEI_GEN_WORKERS
echo "== stamps (diagnostic build)" >> $out/summary.txt
make -C sparseeventid_amd/csrc -j16 EXTRA="-DSEI_EXPERIMENTS -DSEI_STAMPS" -B > $out/build_stamps.log 2>&1
timeout -k 10 200 python3 tools/wave_stamps.py $lvl $c 2>&1 | grep -v amdgpu.ids >> $out/summary.txt
for abl in 1 2 3 4 7 8; do
  echo "== ablation $abl (1 gathers cached, 2 weights L1-resident, 4 no RMW, 8 no MFMA)" >> $out/summary.txt
  rm -f sparseeventid_amd/csrc/conv_wave.o
  make -C sparseeventid_amd/csrc -j16 EXTRA="-DSEI_EXPERIMENTS -DSEI_WAVE_ABL=$abl" > $out/build_abl$abl.log 2>&1
  timeout -k 10 200 $micro 2>&1 | grep -E "fwd" >> $out/summary.txt
done
cat $out/summary.txt
