#!/bin/bash
# Guarded rocprofv3 counter passes over the default bench (each its own run, own timeout).
TAG=${1:-r01}
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT
REPO=$PWD; export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  cd /tmp
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-frames-leg > $OUT/p$i.log 2>&1
  echo "pass $i ($grp): rc=$?"
  cd $REPO
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, os
from collections import defaultdict
for f in sorted(glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        if "align_fused" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:34]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for n, cs in acc.items():
        for c, v in cs.items():
            print("%-36s %-26s n=%d avg=%.6g" % (n, c, len(v), sum(v) / len(v)))
PY
