#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers rest on (run on the GPU box):
#   1. --kernel-trace --stats : per-kernel durations of the default bench workload
#   2. --pmc FETCH_SIZE       : HBM read side   (own pass)
#   3. --pmc WRITE_SIZE       : HBM write side  (own pass)
# Output under gpurun_out/prof/; scripts/summarise_profile.py turns it into profiles/*.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf $OUT
mkdir -p $OUT
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --to-tol 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -20
# 4. matrix-core utilisation (own pass): MFMA busy cycles against CU busy cycles, per kernel
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 $ARGS > $OUT/mfma.log 2>&1 || echo "mfma pass failed (see mfma.log)"
