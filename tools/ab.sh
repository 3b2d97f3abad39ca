#!/bin/bash
# tools/ab.sh STEPS LIB... -- the same box, the same bench, one library build after the other (RTMI_LIBRARY = file name inside
# raytracer_amd/): image checksums, then the serial configuration (one lane, walks on one stream: kernel_ms are each kernel's
# own time) and the driver's configuration.  GPU box only.
set -u
STEPS=$1; shift
mkdir -p gpurun_out/ab
for lib in "$@"; do
  export RTMI_LIBRARY=$lib
  echo "== $lib"
  timeout -k 10 300 python -u tools/checksum.py > gpurun_out/ab/$lib.sum 2>&1 || { tail gpurun_out/ab/$lib.sum; exit 1; }
  cat gpurun_out/ab/$lib.sum
  RTMI_OVERLAP=0 timeout -k 10 300 python -u bench.py --steps $STEPS --warmup 3 --hw-queues 0 --no-cpu-baseline --mesh-view-steps 0 > gpurun_out/ab/$lib.serial.json 2> gpurun_out/ab/$lib.serial.err || { tail gpurun_out/ab/$lib.serial.err; exit 1; }
  python tools/show_bench.py gpurun_out/ab/$lib.serial.json
  timeout -k 10 300 python -u bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --mesh-view-steps 0 > gpurun_out/ab/$lib.json 2> gpurun_out/ab/$lib.err || { tail gpurun_out/ab/$lib.err; exit 1; }
  python tools/show_bench.py gpurun_out/ab/$lib.json
done
