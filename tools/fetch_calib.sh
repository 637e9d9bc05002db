#!/bin/bash
OUT=$PWD/gpurun_out/fetch_calib; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"; do
  cd /tmp; timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$(echo $grp | cut -c1-8) -o pmc -- $REPO/tools/exhaustive/bin/fetch_calib > $OUT/log.txt 2>&1; echo "rc=$?"; cd $REPO
done
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if "stream16" in row["Kernel_Name"] or "gather16" in row["Kernel_Name"]:
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            print("dispatch %-3s %-26s %-24s %14.0f   %.0f us" % (row["Dispatch_Id"], row["Kernel_Name"][:26], row["Counter_Name"], float(row["Counter_Value"]), dur))
PY
