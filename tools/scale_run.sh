#!/bin/bash
# round 6 (VERDICT r5 next #6a): ONE command for the session on an 8-GPU node -- BASELINE configs[1] (weak scaling of the batch
# mode), configs[3] (256 pairs in total, strong scaling, no collective) and configs[4] (one 4096x3072x5 frame tiled over the GPUs,
# an all-reduce of 32 doubles per iteration) at 1, 2, 4 and 8 GPUs, one table at the end (absolute numbers, roofline fraction of the
# dominant kernel, ranks that really talked over RCCL).  Every bench.py call starts its own ranks (python -m torch.distributed.run)
# before anything touches a GPU.
#
#   tools/scale_run.sh [--dry-run] [--gpus "1 2 4 8"] [--out DIR]
#
# --dry-run: for a box with ONE GPU -- the batch lines with --ranks-share-gpu (the N ranks share device 0 and meet over gloo: the
# launcher, the barriers, the MAX over ranks and rank 0's line run, the numbers are NOT scaling measurements) at a small batch, the
# tiled line at one rank.  Exit code 0 only if every line parsed and passed its parity check.
GPUS="1 2 4 8"; OUT=gpurun_out/scale_run; DRY=0
while [ $# -gt 0 ]; do
  case "$1" in
    --dry-run) DRY=1;;
    --gpus) shift; GPUS="$1";;
    --out) shift; OUT="$1";;
    *) echo "usage: $0 [--dry-run] [--gpus \"1 2 4 8\"] [--out DIR]" >&2; exit 2;;
  esac
  shift
done
cd "$(dirname "$0")/.." || exit 2
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
common="--cpu-seconds 2 --no-cpu-all-cores --no-extra-legs"      # a short CPU leg: the in-run parity check rides on it (rank count 1 only)
run() {   # name, bench arguments...
  local name=$1; shift
  echo "== $name: python bench.py $*" >&2
  timeout 1800 python bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "   $name: exit code $?" >&2
}
for n in $GPUS; do
  if [ $DRY = 1 ]; then
    share=""; [ "$n" != 1 ] && share="--ranks-share-gpu"
    run weak_$n   --gpus $n --batch 1024 --steps 5 --warmup 2 $common $share
    run strong_$n --gpus $n --total-pairs 256 --steps 20 --warmup 3 $common $share
    [ "$n" = 1 ] && run tiled_$n --mode tiled --gpus 1 --steps 50 --warmup 3 $common
  else
    run weak_$n   --gpus $n --steps 20 --warmup 3 $common
    run strong_$n --gpus $n --total-pairs 256 --steps 200 --warmup 5 $common
    run tiled_$n  --mode tiled --gpus $n --steps 200 --warmup 5 $common
    run tiled_h_$n --mode tiled --gpus $n --steps 200 --warmup 5 --normal-matrix $common      # configs[4] as worded: H all-reduced too
  fi
done
python3 - "$OUT" <<'PY' | tee "$OUT/table.txt"
import glob, json, os, sys
out = sys.argv[1]
rows, bad = [], 0
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    name = os.path.basename(f)[:-5]
    try:
        d = json.loads([ln for ln in open(f).read().splitlines() if ln.startswith('{"metric"')][-1])
    except Exception as e:
        rows.append((name, "NO LINE (%s)" % e)); bad += 1
        continue
    pc = d.get("parity_check")
    ok = pc.get("pass") if pc else None                       # ranks > 1 carry no parity check (bench.py runs it on one rank only)
    bad += 1 if ok is False else 0
    rf = d.get("roofline", {})
    rows.append((name, "%-44s n_gpus %d  rccl_ranks %s  %12.1f %s  %9.4f ms/step  roofline %.3f (%s)  scaling %-6s parity %s%s" % (
        d["config"].get("workload", "")[:44], d["n_gpus"], d.get("rccl_ranks"), d["value"], d["unit"], d["ms_per_step"], rf.get("frac", float("nan")),
        rf.get("bound"), d.get("scaling"), "n/a" if ok is None else ("pass" if ok else "FAIL"), "  [ranks share ONE gpu: not a scaling measurement]" if d["config"].get("ranks_share_one_gpu") else "")))
base = {}
for name, line in rows:
    print("%-12s %s" % (name, line))
print("\nscaling efficiency is the driver's to compute (value(N) / (N x value(1)) for weak lines, value(N) / value(1) / N for strong ones); table: %s/table.txt" % out)
sys.exit(1 if bad else 0)
PY
