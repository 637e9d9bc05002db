#!/bin/bash
# processes segfault inside exit() under rocprofv3 on the team path and on the frame path.  Which library's exit
# handler is it, and which call of ours arms it?  -> gpurun_out/r04_exit_segv/
OUT=$PWD/gpurun_out/r04_exit_segv; mkdir -p $OUT
REPO=$PWD; export TMPDIR=/tmp; cd /tmp
try() {  # name env... -- program args
  name=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  for e in $envs; do export $e; done
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o t -- python3 "$@" > $OUT/$name.log 2>&1
  echo "$name rc=$?"
  for e in $envs; do unset ${e%%=*}; done
  rm -rf $OUT/$name
}
[ -n "$PART2" ] || try teams_coop DVO_DUMP_MAPS=1 DVO_TEAM_COOP_LAUNCH=1 -- $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-extra-legs --batch 32
[ -n "$PART2" ] || try teams_plain DVO_DUMP_MAPS=1 -- $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-extra-legs --batch 32
[ -n "$PART2" ] || try noteams DVO_DUMP_MAPS=1 -- $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-extra-legs --batch 32 --team 1
[ -n "$PART2" ] || try batch1024 DVO_DUMP_MAPS=1 -- $REPO/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-extra-legs --batch 1024
[ -n "$PART2" ] || try frames_pinned -- $REPO/tools/bench_frames.py --batch 64 --reps 1 --pinned
[ -n "$PART2" ] || try frames_pageable -- $REPO/tools/bench_frames.py --batch 64 --reps 1
cd $REPO
for f in $OUT/*.log; do echo "== $f"; grep -n "SIGSEGV\|PC: @" $f | head -3; done
# second part: the frame path with plain copy streams
cd /tmp
try frames_plain_streams DVO_COPY_STREAM_PRIORITY=0 -- $REPO/tools/bench_frames.py --batch 64 --reps 1
cd $REPO
grep -c "SIGSEGV" $OUT/frames_plain_streams.log
