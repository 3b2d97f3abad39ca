#!/bin/bash
# tools/sweep3.sh "STEPS ENV..." ... -- bench.py with its own step count under each environment (no tests)
set -u
for spec in "$@"; do
  K=${spec%% *}; e=${spec#* }
  echo "run: steps $K env $e" >&2
  env $e timeout -k 10 400 python -u bench.py --no-cpu-baseline --steps $K --warmup 1 2>&1 | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); k=j['kernel_ms']
print('%-60s Mrays/s %.0f ms/step %.1f Mpaths/s %.0f ext %.1f shade %.1f conn %.1f other %.1f' % ('$spec', j['value'], j['ms_per_step'], j['mpaths_per_s'], k['extend'], k['shade'], k['connect'], k['other']))" | tee -a gpurun_out/sweep3.log || exit 1
done
