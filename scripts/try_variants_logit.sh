#!/bin/bash
# Times logit look-ahead shapes for the library variants build_tmp/liblocus_hip_v*.so (experiments;
# the default library first).  Run on the GPU box.
run() { timeout -k 10 100 python3 scripts/perf_sweep.py "$@" 2>/dev/null | grep -o "ms= *[0-9.]*"; }
for lib in default build_tmp/liblocus_hip_v*.so; do
  if [ "$lib" = default ]; then unset LOCUS_HIP_LIB; else export LOCUS_HIP_LIB=$PWD/$lib; fi
  echo "$lib: c4 $(run 2000 20000 500 logit auto) | n1000 d256 $(run 1000 20000 256 logit auto) | n500 d1000 $(run 500 20000 1000 logit auto)"
done
