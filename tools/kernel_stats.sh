#!/bin/bash
# rocprofv3 --stats of a bench configuration, serial step (every kernel alone): top kernels per step
#   tools/kernel_stats.sh --model sbnd --batch 128
export TMPDIR=/tmp
rm -rf gpurun_out/ks; mkdir -p gpurun_out/ks
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py "$@" --steps 3 --warmup 2 --cpu-events 0 --no-prof --overlap-backward 0 --no-prefetch > gpurun_out/ks/run.log 2>&1
f=$(find gpurun_out/ks -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms per step: %.1f (5 steps traced)" % (tot/5e6))
for r in rows[:40]:
    print("%-72s %5s %8.2f ms/step %8.1f us avg" % (r['Name'].replace('void ','')[:72], r['Calls'], float(r['TotalDurationNs'])/5e6, float(r['AverageNs'])/1e3))
PY
rm -rf gpurun_out/ks
