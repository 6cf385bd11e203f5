#!/bin/bash
# Sweep time at the shard sizes of C3 (n = 1000, p = 50 000): d_loc = 10000 / 5000 / 2500 / 1250 = 1 / 2 / 4 / 8 GPUs
set -e
for d in 1250 2500 5000 10000; do
  timeout -k 10 300 python3 scripts/perf_sweep.py 1000 50000 $d "$@"
done
