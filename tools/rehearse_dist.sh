#!/bin/bash
# The multi-rank code path of bench.py with ONE rank under torch.distributed.run (process group on RCCL, barriers, the max / sum
# all-reduces, the accumulator reduce by torch and inside the library, frames in flight on a shared communicator): what the driver
# launches with N ranks, as far as a one-GPU box can run it.
set -u
OUT=${1:-gpurun_out/dist}; mkdir -p "$OUT"
run() { # name, bench args...
  local name=$1; shift
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 "$@" --no-cpu-baseline \
    > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "$name FAILED"; tail -8 "$OUT/$name.err"; exit 1; }
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("%-22s %8.0f Mrays/s %7.1f ms/step n_gpus %d scaling %s reduce %s in-flight %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["n_gpus"], d["scaling"], d["config"]["reduce"], d["config"].get("frames_in_flight")))
PY
}
run weak_torch --steps 6 --warmup 2 &&
run weak_rtmi --steps 6 --warmup 2 --reduce rtmi &&
run strong_torch --scaling strong --total-spp 64 --steps 8 --warmup 1 &&
run strong_rtmi --scaling strong --total-spp 64 --steps 8 --warmup 1 --reduce rtmi
