#!/bin/bash
# Collects the rocprofv3 evidence for one bench configuration on the GPU box.
# usage: tools/profile.sh <tag> [bench args...]      (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --cpu-seconds 0 --no-extra-legs $*"
REPO=$PWD
cd /tmp
# 1) kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
# 2) counters, one small group per pass (TCC: FETCH_SIZE costs 3 slots, WRITE_SIZE 2)
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/pmc$i.log 2>&1
done
cd $REPO
find $OUT -name "*.csv" | head -40
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep what profiles/ keeps (summary, kernel stats, the profiled bench line); the raw per-dispatch CSVs are tens of megabytes
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/trace_kernel_stats.csv 2>/dev/null
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_line_profiled.json      # the bench's JSON line (rocprofv3 writes its own log lines after it)
[ -z "${KEEP_RAW:-}" ] && rm -rf $OUT/trace $OUT/pmc[0-9]*
