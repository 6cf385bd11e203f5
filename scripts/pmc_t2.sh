# FETCH_SIZE of the T = 2 sweep for the library variants given as arguments (paths to .so files)
set -e
export TMPDIR=/tmp
for lib in "$@"; do
  out=$PWD/gpurun_out/pmc_t2_$(basename $lib .so)
  rm -rf $out
  LOCUS_HIP_LIB=$lib timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 scripts/perf_sweep.py 1000 16000 8192 -1,2,0,0 > $out.log 2>&1
  python3 - "$out" "$lib" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
tot = {}
for r in csv.DictReader(open(f)):
    if "mfma_sweep_kernel" in r["Kernel_Name"]:
        tot.setdefault(r["Dispatch_Id"], 0.0)
        tot[r["Dispatch_Id"]] += float(r["Counter_Value"])
v = sorted(tot.values())
print(sys.argv[2].split("/")[-1], "FETCH_SIZE KiB per dispatch: min %.3e median %.3e" % (v[0], v[len(v)//2]))
PY
  grep "cfg=" $out.log | cut -c1-90
done
