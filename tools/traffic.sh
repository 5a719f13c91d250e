#!/bin/bash
# HBM traffic of the conv kernels for bench.py's `roofline.traffic`: separate rocprofv3 --pmc passes (FETCH_SIZE,
# WRITE_SIZE cannot share a pass on gfx950) of the bench command, averaged per launch.  Run on the GPU box:
#   gpurun -- "SEI_COMMIT=$(git rev-parse --short HEAD) bash tools/traffic.sh"   (the box has no .git)
export SEI_GEN_WORKERS=1 TMPDIR=/tmp
out=gpurun_out/traffic; rm -rf $out; mkdir -p $out
cmd="python3 bench.py --steps 1 --warmup 1 --cpu-events 0 --no-prof --batches 1 --no-optin-leg"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/f -- $cmd > $out/f.log 2>&1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/w -- $cmd > $out/w.log 2>&1
python3 - <<'PY'
import csv, glob, json, os
from collections import defaultdict
def load(d, name):
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != name: continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k] += float(row["Counter_Value"]); cnt[k] += 1
    return acc, cnt
fa, fc = load("gpurun_out/traffic/f", "FETCH_SIZE")
wa, wc = load("gpurun_out/traffic/w", "WRITE_SIZE")
# FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3's derived metric
conv = [k for k in fa if any(n in k for n in ("conv_tile_mfma_kernel", "conv_round_mfma_kernel", "conv_mround_mfma_kernel", "conv_wave_mfma_kernel", "conv_band_mfma_kernel"))]
dw = [k for k in fa if "conv_dw_mfma_kernel" in k]
def per_launch(keys):
    n = sum(fc[k] for k in keys)
    return n, sum(fa[k] for k in keys) * 1024 / max(n, 1), sum(wa[k] for k in keys) * 1024 / max(sum(wc[k] for k in keys), 1)
n, f, w = per_launch(conv)
nd, fd, wd = per_launch(dw)
res = {"commit": os.environ.get("SEI_COMMIT"),
       "source": "tools/traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 --cpu-events 0 --no-prof --batches 1 --no-optin-leg",
       "workload": "dune3d train 64 events x 50000 voxels 900^3 tracks",
       "kernel": "conv_round_mfma_kernel + conv_mround_mfma_kernel + conv_tile_mfma_kernel + conv_wave_mfma_kernel (all forward + dX launches)",
       "launches": n, "fetch_size_bytes_per_launch": f, "write_size_bytes_per_launch": w,
       "correction": "gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads: traffic = 2*FETCH_SIZE + WRITE_SIZE (an upper bound: the 64-B row segments of the gathers may be counted exactly)",
       "traffic_bytes_per_launch": 2 * f + w,
       "per_variant_raw_fetch_write_mb": {k: [fa[k] * 1024 / fc[k] / 1e6, wa[k] * 1024 / max(wc[k], 1) / 1e6, fc[k]] for k in sorted(conv + dw)},
       "conv_dw_mfma_kernel": {"launches": nd, "fetch_size_bytes_per_launch": fd, "write_size_bytes_per_launch": wd,
                               "traffic_bytes_per_launch": 2 * fd + wd}}
json.dump(res, open("gpurun_out/%s_conv_fwd_hbm_traffic.json" % os.environ.get("SEI_ROUND", "r05"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $out/f $out/w
