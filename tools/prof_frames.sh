#!/bin/bash
# kernel trace of the frame-preprocessing tool (rows f1+f2); outputs under gpurun_out/prof_frames
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_frames
mkdir -p $O
python3 $R/tools/bench_frames.py --batch 256 --distinct 256 --pinned > $O/pinned.json 2>&1
[ "$1" = "full" ] && python3 $R/tools/bench_frames.py --batch 256 > $O/pageable.json 2>&1
[ "$1" = "full" ] && python3 $R/tools/bench_frames.py --batch 256 --pinned --width 320 --height 240 > $O/pinned_320.json 2>&1
rm -rf $O/trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o frames -- python3 $R/tools/bench_frames.py --batch 256 --pinned --reps 3 > $O/trace.log 2>&1
python3 $R/tools/summarize_prof.py $O > $O/summary.txt 2>&1 || true
for f in $O/*.json; do echo $f; tail -c 1800 $f; echo; done; grep -v "^==" $O/summary.txt | grep "n=" | cut -c1-160 | head -30
