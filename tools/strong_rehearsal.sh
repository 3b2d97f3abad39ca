#!/bin/bash
# Strong-scaling rehearsal on one GPU: rank 0's share of an N-rank run of configs[3] (one frame of 256 spp at 3840x2160
# per step), unpipelined (one handle) and with frames in flight.  usage: tools/strong_rehearsal.sh OUTDIR STEPS "SHARES" "IN_FLIGHT"
set -u
OUT=${1:-gpurun_out/strong}; STEPS=${2:-10}; SHARES=${3:-"8"}; FLIGHT=${4:-"1 3"}
mkdir -p "$OUT"
for n in $SHARES; do
  for f in $FLIGHT; do
    timeout -k 10 400 python3 bench.py --scaling strong --total-spp 256 --width 3840 --height 2160 --share-of "$n" --frames-in-flight "$f" \
      --steps "$STEPS" --warmup 1 --no-cpu-baseline ${EXTRA:-} > "$OUT/strong_${n}_f${f}.json" 2> "$OUT/strong_${n}_f${f}.err" || { echo "share $n in-flight $f FAILED"; tail -5 "$OUT/strong_${n}_f${f}.err"; exit 1; }
    python3 - "$OUT/strong_${n}_f${f}.json" "$n" "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fr = d.get("frames", {})
print("share-of %s frames-in-flight %s: %.1f ms/frame (steady state %.1f)  %.0f Mrays/s  lanes %s slots %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"],
      fr.get("steady_state_ms_per_frame", 0), d["value"], d["config"]["lanes_per_gpu"], d["config"]["slots"]))
PY
  done
done
