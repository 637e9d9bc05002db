#!/bin/bash
# round-3 GPU session driver: tools/r03_gpu.sh <tag> <what...>   what in: tests bench trace frames pmc others
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for what in "$@"; do
case $what in
tests)
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log ;;
testsall)
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -40 $O/pytest.log ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 $O/smoke.log ;;
bench)
  timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-3000 $O/bench.json; tail -3 $O/bench.err ;;
benchquick)
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extra-legs > $O/benchquick.json 2> $O/benchquick.err; echo "bench rc=$?"; cut -c1-2500 $O/benchquick.json; tail -3 $O/benchquick.err ;;
trace)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-extra-legs > $O/trace.log 2>&1)
  python3 tools/summarize_prof.py $O > $O/trace_summary.txt 2>&1; grep "calls=" $O/trace_summary.txt | head -12 ;;
frames)
  tools/prof_frames.sh > $O/frames.log 2>&1; cp -r gpurun_out/prof_frames $O/ 2>/dev/null; tail -40 $O/frames.log ;;
tiled)
  timeout 600 python bench.py --mode tiled --steps 20 --warmup 2 > $O/tiled.json 2> $O/tiled.err; echo "tiled rc=$?"; cut -c1-2500 $O/tiled.json; tail -3 $O/tiled.err
  timeout 600 python bench.py --mode tiled --steps 20 --warmup 2 --width 640 --height 480 --levels 4 > $O/tiled_c2.json 2> $O/tiled_c2.err; echo "tiled c2 rc=$?"; cut -c1-1500 $O/tiled_c2.json ;;
*) echo "unknown step $what" ;;
esac
done
