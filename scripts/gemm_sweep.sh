set -e
for env in "X=1" "LOCUS_HIP_GEMM_BK=3" "LOCUS_HIP_GEMM_KS=1" "LOCUS_HIP_GEMM_TW=4 LOCUS_HIP_GEMM_KS=2" "LOCUS_HIP_GEMM_SPLITK=1" ; do
  echo "== $env"
  env $env timeout -k 10 100 python scripts/perf_gemm.py 2>&1 | grep TFLOP
done
