#!/bin/bash
# Logit look-ahead kernel: who finishes the reduction (RED = 1 vector waves, 2 chain wave), per
# geometry; sweep time in ms at p = 20 000.  Run on the GPU box.
run() { timeout -k 10 100 python3 scripts/perf_sweep.py "$@" 2>/dev/null | grep -o "ms= *[0-9.]*" | tr '\n' ' '; }
for cr in 1 2 4; do
  for npl in 1 2 3 4 5 6 8; do
    [ $cr = 4 ] && [ $npl = 8 ] && continue
    n=$((384 * npl - 8)); d=$((250 * cr))
    echo "CR=$cr NPL=$npl n=$n d=$d: RED1 / RED2 $(run $n 20000 $d logit -3,$cr,1,$npl -3,$cr,2,$npl)"
  done
done
