#!/bin/bash
# usage: tools/pmc_one.sh <tag> "<counters>" [bench args]  -- ONE guarded rocprofv3 --pmc pass over the bench, summary printed
TAG=$1; CNT=$2; shift; shift
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT
REPO=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 100 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-frames-leg "$@" > $OUT/log.txt 2>&1
echo "$TAG rc=$?"
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "align_fused" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for n, cs in acc.items():
        for c, v in cs.items():
            print("  %-40s %-26s n=%d avg=%.6g" % (n, c, len(v), sum(v) / len(v)))
PY
