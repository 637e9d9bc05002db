#!/bin/bash
OUT=$PWD/gpurun_out/prof_upload; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $REPO/tools/measure_upload.py > $OUT/log.txt 2>&1
cd $REPO; python3 - "$OUT" <<'PY'
import csv, glob, sys, os
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print("%-60s calls=%6s avg_us=%9.1f total_ms=%8.2f" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6))
PY
