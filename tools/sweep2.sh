#!/bin/bash
# tools/sweep2.sh STEPS "ENV..." ... -- bench.py under each environment, one summary line each (no tests)
set -u
K=$1; shift
for e in "$@"; do
  echo "run: $e" >&2
  env $e timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps $K --warmup 1 2>&1 | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); k=j['kernel_ms']
print('%-50s Mrays/s %.0f ms/step %.1f Mpaths/s %.0f ext %.1f shade %.1f conn %.1f other %.1f' % ('$e', j['value'], j['ms_per_step'], j['mpaths_per_s'], k['extend'], k['shade'], k['connect'], k['other']))" || exit 1
done
