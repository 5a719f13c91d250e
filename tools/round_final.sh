#!/bin/bash
# The round's closing run on ONE box: full GPU test suite, smoke(), then tools/round_profiles.sh (bench line with the opt-in leg and
# the CPU baseline, rocprofv3 kernel stats, serial step, HBM traffic, MFMA-busy pass).   SEI_ROUND=r05 bash tools/round_final.sh
R=${SEI_ROUND:-r05}; mkdir -p gpurun_out/$R
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/$R/${R}_gpu_tests.log 2>&1; rc=$?
echo "gpu tests rc=$rc: $(tail -1 gpurun_out/$R/${R}_gpu_tests.log)"
[ $rc -ne 0 ] && { tail -30 gpurun_out/$R/${R}_gpu_tests.log; exit $rc; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/round_profiles.sh
