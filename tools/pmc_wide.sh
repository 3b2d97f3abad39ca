#!/bin/bash
# tools/pmc_wide.sh TAG -- SQ counter pass + kernel trace of a short bench run (environment inherited, e.g. RTMI_WIDE=1)
set -u
TAG=${1:-wide}
OUT=gpurun_out/pmcw_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1 || { echo "pmc sq failed"; tail -5 $OUT/pmc_sq.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -E "k_trace|k_shade|^kernel|^==" $OUT/summary.txt | grep -v "#"
