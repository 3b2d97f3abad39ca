#!/bin/bash
# tools/sweep.sh -- GPU box: correctness first, then bench variants (one JSON line each in gpurun_out/sweep.log)
#   tools/sweep.sh "VAR=val VAR2=val" "VAR=val" ...   (each argument is one environment to run bench.py under)
set -u
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/sweep_tests.log 2>&1 || { tail -40 gpurun_out/sweep_tests.log; exit 1; }
tail -2 gpurun_out/sweep_tests.log
: > gpurun_out/sweep.log
run() { echo "## $*" >> gpurun_out/sweep.log; env $* timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep '^{' >> gpurun_out/sweep.log || exit 1; }
if [ $# -eq 0 ]; then set -- "RTMI_REFILL=16"; fi
for e in "$@"; do run "$e"; done
python3 - <<'PY'
import json
for l in open('gpurun_out/sweep.log'):
    if l.startswith('##'): print(l.strip(), end='  ')
    else:
        j=json.loads(l); k=j['kernel_ms']
        print('Mrays/s %.0f ms/step %.1f ext %.1f shade %.1f conn %.1f frac %.3f' % (j['value'], j['ms_per_step'], k['extend'], k['shade'], k['connect'], j['roofline']['frac']))
PY
