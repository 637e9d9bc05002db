#!/bin/bash
# tools/exhaustive/line_fetch.hip under rocprofv3 (kernel trace + one PMC pass) -> gpurun_out/r04_line_fetch/
OUT=$PWD/gpurun_out/r04_line_fetch; mkdir -p $OUT
BIN=$PWD/tools/exhaustive/bin/line_fetch
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT -o lf -- $BIN > $OUT/log.txt 2>&1
cd - > /dev/null
python3 - $OUT <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out = sys.argv[1]
dur = {}
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"][:30], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
acc = defaultdict(dict)
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
for d in sorted(acc, key=int):
    n, us = dur.get(d, ("?", 0))
    print("dispatch %3s %-30s %9.1f us  " % (d, n, us) + "  ".join("%s=%d" % (k.replace("TCC_", "").replace("_sum", ""), v) for k, v in sorted(acc[d].items())))
PY
cat $OUT/log.txt | tail -3
