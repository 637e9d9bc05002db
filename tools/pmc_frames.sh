#!/bin/bash
# Guarded rocprofv3 counter passes over the frame-preprocessing tool (each pass its own run and timeout).
OUT=$PWD/gpurun_out/pmc_frames; rm -rf $OUT; mkdir -p $OUT
REPO=$PWD; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  cd /tmp
  timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/tools/bench_frames.py --batch 64 --reps 1 --pinned > $OUT/p$i.log 2>&1
  echo "pass $i ($grp): rc=$?"
  cd $REPO
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, os
from collections import defaultdict
for f in sorted(glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        if n.startswith("dvo::") and "align" not in n and "enlist" not in n:
            acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for n, cs in sorted(acc.items()):
        for c, v in cs.items():
            v = sorted(v)
            print("%-42s %-22s n=%d max=%.6g sum=%.6g" % (n, c, len(v), v[-1], sum(v)))
PY
