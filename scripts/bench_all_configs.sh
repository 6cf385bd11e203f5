#!/bin/bash
# The four bench lines (BASELINE.json configs C2-C5) of one run -> gpurun_out/bench_lines.json
# usage: bash scripts/bench_all_configs.sh [extra bench.py flags]
set -e
out=gpurun_out/bench_lines.json
: > "$out"
for c in c2 c3 c4 c5; do
  echo "== $c" >&2
  timeout -k 10 500 python3 bench.py --config $c "$@" > gpurun_out/bench_$c.log 2> gpurun_out/bench_$c.err
  grep '^{"metric"' gpurun_out/bench_$c.log >> "$out"
done
python3 - <<'PY'
import json
for ln in open("gpurun_out/bench_lines.json"):
    r = json.loads(ln)
    t = r.get("time_to_elbo_tol") or {}
    print(r["config"]["name"], "ms/step %.2f" % r["ms_per_step"], "value %.3e" % r["value"],
          "kernel_ms %.2f" % r["roofline"]["kernel_ms"], "frac %.3f" % r["roofline"]["frac"],
          "cpu %.3e" % r["cpu_baseline"]["value"], "to_tol", t.get("iterations"), "%.2fs" % t.get("seconds", 0),
          "monotone", r["config"]["elbo_monotone"])
PY
