#!/bin/bash
# round 5: level 1 (64 -> 64 planes) on the wave-private kernel (SEI_L1_ROUNDS=0) against the sixteen-wave rounds (=1), WHOLE step,
# ONE experiment build on ONE box, interleaved twice, overlapped default and serial; then a rocprofv3 kernel trace of each
# overlapped arm (3 steps) so that the per-kernel totals say WHERE a layer-level win goes in the step.
#   tools/l1_step_ab.sh
export TMPDIR=/tmp
out=gpurun_out/r05/l1_step_ab.log; mkdir -p gpurun_out/r05; : > $out
make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_EXPERIMENTS" > gpurun_out/r05/l1_step_ab_build.log 2>&1 || { tail -20 gpurun_out/r05/l1_step_ab_build.log; exit 1; }
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1), {k:round(v['ms'],1) for k,v in d['roofline']['per_kind_per_step'].items()})"; }
for rep in 1 2; do
  for arm in 0 1; do
    echo "== overlapped SEI_L1_ROUNDS=$arm" >> $out
    SEI_L1_ROUNDS=$arm timeout -k 10 300 python bench.py --steps 8 --warmup 3 --cpu-events 0 --batches 1 2>/dev/null | tail -1 | line >> $out
    echo "== serial     SEI_L1_ROUNDS=$arm" >> $out
    SEI_L1_ROUNDS=$arm timeout -k 10 300 python bench.py --steps 8 --warmup 3 --cpu-events 0 --batches 1 --overlap-backward 0 --no-prefetch 2>/dev/null | tail -1 | line >> $out
  done
  echo "rep $rep done"
done
for arm in 0 1; do
  rm -rf gpurun_out/ks; mkdir -p gpurun_out/ks
  SEI_L1_ROUNDS=$arm SEI_GEN_WORKERS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py --steps 3 --warmup 2 --cpu-events 0 --no-prof --batches 1 > gpurun_out/ks/run.log 2>&1
  f=$(find gpurun_out/ks -name "*kernel_stats.csv" | head -1)
  echo "== kernel totals, overlapped step, SEI_L1_ROUNDS=$arm (5 steps traced, ms per step)" >> $out
  python3 - "$f" >> $out <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("  all kernels %.1f" % (tot/5e6))
grp={}
for r in rows:
    n=r['Name']
    key = "conv_dw" if "conv_dw" in n else "conv fwd/dX" if re.search(r"conv_(round|mround|tile|wave)", n) else "bn/elementwise" if re.search(r"bn_|colreduce|finalize|leaky|add_", n) else "rulebook" if re.search(r"probe|tile_rules|scan|insert|assign|flag|transpose_seg|rowof", n) else "other"
    grp[key]=grp.get(key,0)+float(r['TotalDurationNs'])/5e6
print("  " + "  ".join("%s %.1f" % kv for kv in sorted(grp.items())))
for r in rows[:14]:
    print("  %-64s %5s %8.2f ms/step %8.1f us avg" % (re.sub(r"\(.*","",r['Name'].replace('void ','').replace('sei::',''))[:64], r['Calls'], float(r['TotalDurationNs'])/5e6, float(r['AverageNs'])/1e3))
PY
done
rm -rf gpurun_out/ks
make -C sparseeventid_amd/csrc -j16 -B > gpurun_out/r05/l1_step_ab_build.log 2>&1
cat $out
