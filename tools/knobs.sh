#!/bin/bash
# tools/knobs.sh STEPS "ENV=.. ENV=.." ["..." ...] -- bench.py under each environment setting, one line per run.  GPU box only.
STEPS=$1; shift
for v in "$@"; do
  env $v timeout -k 10 300 python -u bench.py --steps $STEPS --warmup 2 --no-cpu-baseline > gpurun_out/knob.json 2> gpurun_out/knob.err || { echo "$v FAILED"; tail -3 gpurun_out/knob.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/knob.json")); r = d["roofline"]
print("%-44s %6.0f Mrays/s %6.1f ms/step drain %4.0f ms | %s frac %.3f | ms %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["drain_ms"], r["kernel"], r["frac"],
      {k[:3]: round(v) for k, v in d["kernel_ms"].items()}), flush=True)
PY
done
