#!/bin/bash
# Work-item order of the conv kernels (runs of SEI_XCD_CHUNK items per XCD): whole-step time AND HBM traffic per variant, one box.
#   tools/xcd_chunk_ab.sh 16 64 256
out=gpurun_out/xcd_chunk_ab.log; : > $out
for v in "$@"; do
  make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_XCD_CHUNK=$v" > /dev/null 2>&1
  echo "== SEI_XCD_CHUNK=$v" >> $out
  for i in 1 2; do
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-events 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1), round(d['roofline']['frac'],3), {k:round(v['ms'],1) for k,v in d['roofline']['per_kind_per_step'].items()})" >> $out
  done
  SEI_COMMIT=xcd$v bash tools/traffic.sh > /dev/null 2>&1
  python3 -c "
import json;d=json.load(open('gpurun_out/r03_conv_fwd_hbm_traffic.json'))
print('traffic MB per launch %.0f = 2 x %.1f + %.1f' % (d['traffic_bytes_per_launch']/1e6, d['fetch_size_bytes_per_launch']/1e6, d['write_size_bytes_per_launch']/1e6))
for k,v in sorted(d['per_variant_raw_fetch_write_mb'].items()):
    if 'dw_' not in k: print('   ', k.replace('sei::',''), [round(x,1) for x in v[:2]])
" >> $out
done
make -C sparseeventid_amd/csrc -j16 -B > /dev/null 2>&1
cat $out
