#!/bin/bash
# dW kernel time via rocprof on the micro-benchmark
for spec in "0 32" "1 64" "2 96" "3 128" "4 160" "5 192"; do set -- $spec
 for mt in default 1 2 3 4; do
  if [ $mt = default ]; then unset SEI_DW_MT; else export SEI_DW_MT=$mt; fi
  rm -rf gpurun_out/dwp; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dwp -- python3 tools/conv_micro.py --level $1 --cin $2 --cout $2 --events 32 --iters 3 --bwd > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/dwp/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'conv_dw_mfma' in r['Name']:
        print("L$1 C=$2 mt=$mt", r['Name'].split('(')[0][-28:], "avg us", float(r['AverageNs'])/1e3)
PY
 done
done
