#!/bin/bash
B="timeout 100 python bench.py --cpu-seconds 0 --steps 10 --warmup 2"
short() { python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-52s %9.0f aligns/s  kernel %.3f ms' % (sys.argv[1], d['value'], d['roofline']['kernel_ms']))" "$1"; }
for v in "" _prio1; do
  DVO_LIB_VARIANT=$v $B 2>/dev/null | short "lib=$v block=512"
  DVO_LIB_VARIANT=$v $B 2>/dev/null | short "lib=$v block=512 (repeat)"
done
sed -i 's/os.environ\["DVO_LIB_VARIANT"\] = "_stamps"/os.environ["DVO_LIB_VARIANT"] = os.environ.get("STAMPS_VARIANT", "_stamps")/' tools/exp7.py
STAMPS_VARIANT=_stamps_prio1 timeout 100 python tools/exp7.py 256 512 155000 1 2>&1 | grep -v amdgpu
