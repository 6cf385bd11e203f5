#!/bin/bash
# Times chain-bound sweep shapes for the library variants build_tmp/liblocus_hip_v*.so (experiments;
# the default library first).  Run on the GPU box.
run() { timeout -k 10 100 python3 scripts/perf_sweep.py "$@" 2>/dev/null | grep -o "ms= *[0-9.]*"; }
for lib in default build_tmp/liblocus_hip_v*.so; do
  if [ "$lib" = default ]; then unset LOCUS_HIP_LIB; else export LOCUS_HIP_LIB=$PWD/$lib; fi
  echo "$lib: d1250 $(run 1000 50000 1250 auto) | c2 $(run 1000 20000 1000 auto) | n200 $(run 200 20000 1000 auto)"
done
