#!/bin/bash
# usage: scripts/pmc_sweep.sh "<perf_sweep args>" ; PMC passes on the sweep kernel
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
ARGS="scripts/perf_sweep.py $1"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc/p*/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Kernel_Name'] for k in ('sweep_kernel', 'sweep1_kernel', 'logit_la_kernel')):
            acc[r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
for k,v in sorted(acc.items()):
    last=v[-1]
    print(f"{k:28s} value={last[0]:.4e} dur_ms={last[1]/1e6:.2f} (n={len(v)})")
PY
