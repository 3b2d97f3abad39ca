#!/bin/bash
# tools/profile_round.sh TAG [bench args...] -- run on the GPU box (via gpurun): rocprofv3 kernel trace + stats of
# `python3 bench.py <args>` and the PMC passes of the same command, each in a run of its own (no trace domains beside --pmc;
# at most 8 SQ / 4 TCC / 4 TCP / 2 TA / 2 GRBM counters per pass).  Raw output under gpurun_out/prof_TAG/, summaries in
# gpurun_out/prof_TAG/{summary.txt,traffic.json,pmc_ta.txt,pmc_latency.txt}: copy what is to be judged into profiles/.
set -u
TAG=${1:-r03}; shift
ARGS="${*:---steps 20 --warmup 5}"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py $ARGS --no-cpu-baseline --mesh-view-steps 0"
TRACED="$BENCH"
echo "$BENCH" > $OUT/command.txt
pass() { # name counters...
  local name=$1; shift
  echo "pass $name"
  timeout -s KILL 400 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $BENCH > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $OUT/$name.log; return 1; }
}
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $TRACED > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
grep "^{\"metric\"" $OUT/trace.log > $OUT/bench.json
pass pmc_fetch FETCH_SIZE || exit 1
pass pmc_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
pass pmc_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU || exit 1
pass pmc_sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT || exit 1
pass pmc_ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE || exit 1
pass pmc_a TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE || exit 1
pass pmc_b TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum || exit 1
python3 tools/trace_union.py $OUT $OUT/union.json > $OUT/union.txt 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
python3 tools/summarize_prof.py $OUT "$TRACED" > $OUT/summary.txt 2>&1
python3 tools/traffic.py $OUT $OUT/traffic.json "$BENCH (tools/profile_round.sh)" > $OUT/traffic.log 2>&1
python3 tools/pmc_tables.py $OUT > $OUT/pmc_tables.txt 2>&1
# the raw per-dispatch tables are tens of MiB per pass (gpurun_out/ travels back only below 64 MiB): keep the summaries
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
grep -E "^k_|^kernel|^==" $OUT/summary.txt | grep -v "#" | head -40
cat $OUT/union.txt
cat $OUT/pmc_tables.txt
