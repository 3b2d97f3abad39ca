#!/bin/bash
# tools/sweep4.sh "BENCH-ARGS | ENV..." ... -- bench.py with its own arguments under each environment (no tests)
set -u
for spec in "$@"; do
  args=${spec%%|*}; e=${spec#*|}
  echo "run: $spec" >&2
  env $e timeout -k 10 400 python -u bench.py --no-cpu-baseline $args 2>&1 | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); k=j['kernel_ms']
print('%-70s Mrays/s %.0f ms/step %.1f Mpaths/s %.0f ext %.1f shade %.1f conn %.1f other %.1f' % ('$spec', j['value'], j['ms_per_step'], j['mpaths_per_s'], k['extend'], k['shade'], k['connect'], k['other']))" | tee -a gpurun_out/sweep4.log || exit 1
done
