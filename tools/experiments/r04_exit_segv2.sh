#!/bin/bash
OUT=$PWD/gpurun_out/r04_exit_segv; mkdir -p $OUT
REPO=$PWD; export TMPDIR=/tmp; cd /tmp
for v in "0" "1" "2" "4" "8" "2 -1"; do
  n=$(echo $v | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/min_$n -o t -- python3 $REPO/tools/experiments/r04_exit_segv_min.py $v > $OUT/min_$n.log 2>&1
  echo "torch only, streams/priority $v: rc=$?"
  rm -rf $OUT/min_$n
done
