for mode in odd even; do
  if [ $mode = even ]; then export LOCUS_HIP_MFMA_EVEN=1; else unset LOCUS_HIP_MFMA_EVEN; fi
  echo "== $mode"
  timeout 100 python3 scripts/perf_sweep.py 1000 50000 4096 -1,2,2,0 | grep cfg=
  timeout 100 python3 scripts/perf_sweep.py 1000 50000 2000 -1,1,2,0 | grep cfg=
  timeout 100 python3 scripts/perf_sweep.py 1000 50000 2500 -1,1,1,0 | grep cfg=
  timeout 100 python3 scripts/perf_sweep.py 1000 20000 1000 -1,1,3,0 | grep cfg=
  timeout 100 python3 scripts/perf_sweep.py 500 20000 1000 -1,1,2,0 -1,1,3,0 | grep cfg=
  timeout 100 python3 scripts/perf_sweep.py 2000 20000 500 -1,1,3,0 -1,1,2,0 | grep cfg=
done
