#!/bin/bash
# disassembles the gfx950 code object of dvo_fused.hip inside lib/libdvo_amd<variant>.so: tools/disasm_fused.sh [variant] > out.s
LIB=$(dirname $0)/../rgbd_odometry_amd/lib/libdvo_amd${1:-}.so
TMP=$(mktemp -d)
python3 - "$LIB" "$TMP" <<'PY'
import re, struct, sys
data = open(sys.argv[1], "rb").read()
n = 0
for m in re.finditer(b"\x7fELF\x02\x01\x01", data):
    o = m.start()
    if o == 0 or struct.unpack_from("<H", data, o + 18)[0] != 224:
        continue
    shoff = struct.unpack_from("<Q", data, o + 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", data, o + 0x3A)
    open("%s/co_%d.elf" % (sys.argv[2], n), "wb").write(data[o:o + shoff + shentsize * shnum])
    n += 1
PY
for f in $TMP/co_*.elf; do
  if /opt/rocm/lib/llvm/bin/llvm-readelf -s $f 2>/dev/null | grep -q align_fused2_kernel; then
    /opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $f
  fi
done
rm -rf $TMP
