#!/usr/bin/env python3
"""Instruction mix of the steady-state point loops of align_fused2_kernel<256,false> (the loops that hold 6 = 3 rounds x 2 or 4 = 2 x 2
dwordx3 gathers) out of lib/libdvo_amd<variant>.so.  usage: tools/hotloop_stats.py [variant] [block]"""
import re, subprocess, sys, os
from collections import Counter
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = sys.argv[1] if len(sys.argv) > 1 else ""
block = sys.argv[2] if len(sys.argv) > 2 else "256"
txt = subprocess.run([os.path.join(root, "tools", "disasm_fused.sh"), variant], capture_output=True, text=True).stdout
ins, on = [], False
for line in txt.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
    if m:
        on = ("align_fused2_kernelILi%sELb0E" % block) in m.group(1)
        continue
    if on:
        m = re.search(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):(.*)$", line)
        if m: ins.append((int(m.group(3), 16), m.group(1), m.group(2), m.group(4)))
idx = {x[0]: i for i, x in enumerate(ins)}
base = ins[0][0]
for i, (a, op, args, rest) in enumerate(ins):
    if not op.startswith("s_cbranch"): continue
    t = re.search(r"\+0x([0-9a-f]+)>", rest)
    if not t: continue
    tgt = base + int(t.group(1), 16)
    if tgt > a or tgt not in idx: continue
    body = ins[idx[tgt]:i + 1]
    ng = sum(1 for x in body if x[1] == "global_load_dwordx3")
    if ng not in (4, 6) or len(body) > 1200: continue
    rounds = ng // 2
    c = Counter(x[1] for x in body)
    valu = sum(n for o, n in c.items() if o.startswith("v_"))
    print("loop %x..%x: %d instructions, %d rounds: VALU %.1f / round, LDS %.1f, SALU %.1f, VMEM %.1f, scratch %d" % (
        tgt, a, len(body), rounds, valu / rounds, sum(n for o, n in c.items() if o.startswith("ds_")) / rounds,
        sum(n for o, n in c.items() if o.startswith("s_") and o != "s_waitcnt") / rounds,
        sum(n for o, n in c.items() if o.startswith("global_") or o.startswith("buffer_")) / rounds, sum(n for o, n in c.items() if o.startswith("scratch"))))
    if "-v" in sys.argv:
        for o, n in c.most_common(60): print("   %4d %s" % (n, o))
