#!/bin/bash
# tools/torchrun_probe.sh -- bench.py launched as the driver launches it for N>1 (torch.distributed.run), with one rank, under a few environments
set -u
run() { echo "## $*"; env "$@" timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('Mrays/s %.0f ms/step %.1f' % (j['value'], j['ms_per_step']))"; }
echo "## direct OMP_NUM_THREADS=1"; OMP_NUM_THREADS=1 timeout -k 10 300 python -u bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-160
run A=1
run GPU_MAX_HW_QUEUES=24
run RTMI_SLOTS=8
run GPU_MAX_HW_QUEUES=32 RTMI_SLOTS=12
