#!/bin/bash
# A/B of source variants on the WHOLE step, on ONE box: rebuild the library per variant (EXTRA flags), run bench.py twice.
#   tools/bench_ab.sh "" "-DSEI_XCD_CHUNK=64" ""
out=gpurun_out/bench_ab.log; : > $out
for v in "$@"; do
  make -C sparseeventid_amd/csrc -j16 -B EXTRA="$v" > /dev/null 2>&1
  echo "== variant [$v]" >> $out
  for i in 1 2; do
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-events 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1), round(d['roofline']['frac'],3), {k:round(v['ms'],1) for k,v in d['roofline']['per_kind_per_step'].items()})" >> $out
  done
done
make -C sparseeventid_amd/csrc -j16 -B > /dev/null 2>&1
cat $out
