#!/bin/bash
# kernel trace of the tiled path at one rank: which kernels an iteration consists of (is there an RCCL kernel for a one-rank in-place
# all-reduce?), graph replay and direct submission (-> profiles/r04_tiled/)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_tiled; mkdir -p $OUT; REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/graph -o t -- python3 $REPO/bench.py --mode tiled --cpu-seconds 0 --steps 50 --warmup 2 > $OUT/graph.log 2>&1
DVO_TILED_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/direct -o t -- python3 $REPO/bench.py --mode tiled --cpu-seconds 0 --steps 50 --warmup 2 > $OUT/direct.log 2>&1
cd $REPO
for v in graph direct; do
  f=$(find $OUT/$v -name "*kernel_stats.csv" | head -1); cp $f $OUT/${v}_kernel_stats.csv
  grep '^{' $OUT/$v.log | tail -1 > $OUT/${v}_bench_line.json
  echo "== $v"; head -12 $OUT/${v}_kernel_stats.csv | cut -c1-160
  t=$(find $OUT/$v -name "*kernel_trace.csv" | head -1)
  python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 60 dispatches: name, start-to-start gap, duration
prev = None
out = []
for r in rows[-60:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append("%-60s dur %6.2f us  gap-from-prev-end %6.2f us" % (r["Kernel_Name"][:60], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
    prev = e
print("\n".join(out[-24:]))
PY
  rm -rf $OUT/$v
done
