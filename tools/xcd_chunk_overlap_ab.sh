#!/bin/bash
# Work-item run length per XCD under the OVERLAPPED default step (dX beside dW beside BatchNorm-backward: does the conv
# kernels' HBM traffic matter now?): whole-step time only, interleaved, one box.   tools/xcd_chunk_overlap_ab.sh 16 64 512 16 64 512
out=gpurun_out/xcd_chunk_overlap_ab.log; : > $out
for v in "$@"; do
  make -C sparseeventid_amd/csrc -j16 -B EXTRA="-DSEI_XCD_CHUNK=$v" > /dev/null 2>&1
  echo "== SEI_XCD_CHUNK=$v" >> $out
  for i in 1 2; do
    timeout -k 10 300 python bench.py --steps 8 --warmup 3 --cpu-events 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],1), round(d['roofline']['frac'],3))" >> $out
  done
done
make -C sparseeventid_amd/csrc -j16 -B > /dev/null 2>&1
cat $out
