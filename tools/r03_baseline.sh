#!/bin/bash
# tools/r03_baseline.sh -- round-3 reference point on one box: GPU tests, image checksums, the driver's bench command, and the same
# bench on one lane with the two walks on one stream (kernel_ms then are each kernel's own time).  GPU box only.
set -u
mkdir -p gpurun_out/r03
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03/pytest.log 2>&1 || { tail -20 gpurun_out/r03/pytest.log; exit 1; }
tail -3 gpurun_out/r03/pytest.log
timeout -k 10 300 python -u tools/checksum.py > gpurun_out/r03/checksum.txt 2>&1 || { tail gpurun_out/r03/checksum.txt; exit 1; }
cat gpurun_out/r03/checksum.txt
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench20.json 2> gpurun_out/r03/bench20.err || { tail gpurun_out/r03/bench20.err; exit 1; }
python tools/show_bench.py gpurun_out/r03/bench20.json
RTMI_OVERLAP=0 timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --hw-queues 0 --no-cpu-baseline > gpurun_out/r03/bench20_serial.json 2> gpurun_out/r03/bench20_serial.err || { tail gpurun_out/r03/bench20_serial.err; exit 1; }
python tools/show_bench.py gpurun_out/r03/bench20_serial.json
