#!/bin/bash
# round 5: SQ / TCP / TCC counters of ONE conv layer (tools/sweep_tile.py), separate --pmc passes (kernel trace only).
#   tools/pmc_probe.sh "1:64 2:96" [plans]     -> gpurun_out/r05/pmc_probe_L<l>.txt
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
plans=${2:-rounds}
rocprofv3 -L > gpurun_out/r05/pmc_counters_available.txt 2>&1
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"
 "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM"
 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVES_EQ_64"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum"
 "TA_BUSY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
)
for lc in ${1:-1:64}; do
  l=${lc%%:*}; c=${lc##*:}
  out=gpurun_out/r05/pmc_probe_L${l}.txt; : > $out
  i=0
  for p in "${PASSES[@]}"; do
    i=$((i+1)); d=gpurun_out/r05/pmcp_$i; rm -rf $d
    timeout -k 10 240 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $d -- python3 tools/sweep_tile.py --events ${SEI_AB_EVENTS:-64} --level $l --c $c --rounds 1 --iters 2 --plans $plans > gpurun_out/r05/pmcp_$i.log 2>&1
    echo "== pass $i rc=$? : $p" >> $out
    python3 tools/pmc_sum.py $d conv_ >> $out 2>&1
    rm -rf $d
  done
  grep -E "^L" gpurun_out/r05/pmcp_1.log >> $out
done
wc -l gpurun_out/r05/pmc_probe_L*.txt
