#!/bin/bash
# kernel-trace + stats of the default bench command (short run); summary -> gpurun_out/prof_<tag>/
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p $OUT
REPO=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-frames-leg $* > $OUT/trace.log 2>&1
echo "rc=$?"; cd $REPO
tail -1 $OUT/trace.log | cut -c1-400
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; grep -v "copyBuffer\|fillBuffer" $OUT/summary.txt | head -30
