#!/bin/bash
# experiment build: tile-kernel plans of the 160- and 128-plane levels when the level has < 100 000 rows (32-event batches):
# the row threshold of the eight-wave plans
make -C sparseeventid_amd/csrc -j16 -B EXTRA=-DSEI_EXPERIMENTS > /dev/null 2>&1
for ev in 32 16; do
for plan in "" "1,1,1,160,2,8" "1,1,1,160,5,8" "2,1,2,160,5,4" "2,1,1,160,2,4"; do
  echo "== events $ev SEI_FWD_PLAN=[$plan]"
  SEI_FWD_PLAN="$plan" python tools/sweep_band.py --events $ev --levels 4 --plans tile --rounds 2 2>&1 | grep -v amdgpu | tail -1
done
for plan in "" "1,1,1,128,2,8" "2,1,2,128,2,4"; do
  echo "== events $ev SEI_FWD_PLAN=[$plan]"
  SEI_FWD_PLAN="$plan" python tools/sweep_band.py --events $ev --levels 3 --plans tile --rounds 2 2>&1 | grep -v amdgpu | tail -1
done
done
make -C sparseeventid_amd/csrc -j16 -B > /dev/null 2>&1
