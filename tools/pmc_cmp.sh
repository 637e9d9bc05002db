#!/bin/bash
# usage: tools/pmc_cmp.sh <tag> [bench args]  -- guarded rocprofv3 --pmc passes (memory side + issue side) over the bench
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p $OUT
REPO=$PWD; export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TA_BUSY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  cd /tmp
  timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-frames-leg $* > $OUT/p$i.log 2>&1
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
            print("%-36s %-32s n=%d avg=%.6g" % (n, c, len(v), sum(v) / len(v)))
PY
