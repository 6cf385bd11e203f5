# usage: bash scripts/gemm_sweep.sh "ENV=1 ..." ["ENV2=..." ...]: scripts/perf_gemm.py under each environment
set -e
for env in "$@"; do
  echo "== $env"
  env $env timeout -k 10 100 python scripts/perf_gemm.py 2>&1 | grep TFLOP
done
