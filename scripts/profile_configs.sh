#!/bin/bash
# Per-config rocprofv3 evidence (run on the GPU box): for each of C2 / C4 / C5 a short driver run
# (scripts/perf_drivers.py, 4 iterations) under
#   1. --kernel-trace --stats   2. --pmc FETCH_SIZE   3. --pmc WRITE_SIZE   (one pass per run)
# Output under gpurun_out/prof_<cfg>/; scripts/summarise_configs.py turns it into profiles/*.
set -e
export TMPDIR=/tmp
for CFG in ${@:-c2 c4 c5}; do
  OUT=$PWD/gpurun_out/prof_$CFG
  rm -rf $OUT
  mkdir -p $OUT
  ARGS="scripts/perf_drivers.py $CFG 4"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
  echo "$CFG done"
done
