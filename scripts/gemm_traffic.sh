#!/bin/bash
# HBM read traffic of the fp64 GEMM on the contraction shapes of scripts/perf_gemm.py, XCD-aware tile
# order against the plain launch order (default; LOCUS_HIP_GEMM_XCD=1 selects the XCD-aware one).  Run on the GPU box; one counter
# pass per order.  Prints KiB per kernel dispatch (FETCH_SIZE; x2 on gfx950, MI355X_MICROARCH.md).
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/gemm_traffic
rm -rf $OUT
mkdir -p $OUT
LOCUS_HIP_GEMM_XCD=1 timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/xcd -- python3 scripts/perf_gemm.py > $OUT/xcd.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/plain -- python3 scripts/perf_gemm.py > $OUT/plain.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for order in ("xcd", "plain"):
    f = glob.glob(f"gpurun_out/gemm_traffic/{order}/**/*counter_collection.csv", recursive=True)[0]
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "gemm_f64" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            key = (r["Kernel_Name"].split("(")[0], r["Grid_Size"])
            rows.setdefault(key, []).append(float(r["Counter_Value"]))
    for (k, gs), v in rows.items():
        v.sort()
        print(f"{order:6s} {k:60s} grid {gs:>9s} calls {len(v):3d}  HBM read {2 * v[len(v)//2] * 1024 / 1e9:8.3f} GB")
PY
