#!/bin/bash
# rocprofv3 kernel-trace summaries of the sweep at the shard sizes of C3 (n = 1000, p = 50 000):
# d_loc = 5000 / 2500 / 1250 = what each of 2 / 4 / 8 GPUs sweeps.  Output gpurun_out/prof_shard_<d>/
set -e
export TMPDIR=/tmp
for D in ${@:-5000 2500 1250}; do
  OUT=$PWD/gpurun_out/prof_shard_$D
  rm -rf $OUT
  mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/perf_sweep.py 1000 50000 $D auto > $OUT/trace.log 2>&1
  grep cfg= $OUT/trace.log
done
