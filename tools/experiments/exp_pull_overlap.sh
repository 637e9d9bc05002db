#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ovprof; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ovprof -o t -- python3 $R/tools/experiments/exp_pull_overlap.py 2>&1 | grep "ms per step"
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/ovprof/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')) for r in csv.DictReader(open(f))]
rows.sort()
# last step only: from the last but one align kernel's end
aligns = [r for r in rows if 'align_fused' in r[2]]
t_lo = aligns[-2][1]; t_hi = aligns[-1][1]
rows = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
pull = [r for r in rows if 'gather' in r[2]]; other = [r for r in rows if 'gather' not in r[2]]
def union(iv):
    iv = sorted(iv); out = []
    for a, b in iv:
        if out and a <= out[-1][1]: out[-1][1] = max(out[-1][1], b)
        else: out.append([a, b])
    return out
up, uo = union([(a, b) for a, b, _ in pull]), union([(a, b) for a, b, _ in other])
def total(u): return sum(b - a for a, b in u)
def inter(u, v):
    i = j = 0; t = 0
    while i < len(u) and j < len(v):
        a, b = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if a < b: t += b - a
        if u[i][1] < v[j][1]: i += 1
        else: j += 1
    return t
print("step span %.3f ms; pull kernels busy %.3f ms in %d launches; other kernels busy %.3f ms; both at once %.3f ms; neither %.3f ms" % (
    (t_hi - t_lo) / 1e6, total(up) / 1e6, len(pull), total(uo) / 1e6, inter(up, uo) / 1e6, ((t_hi - t_lo) - total(union([tuple(x) for x in up + uo]))) / 1e6))
for a, b, n in pull[:12]: print("  pull  %.3f .. %.3f ms (%.3f)" % ((a - t_lo) / 1e6, (b - t_lo) / 1e6, (b - a) / 1e6))
# what runs between the end of the first chunk's pull and the start of the next chunk's
g0, g1 = pull[1][1], pull[2][0]
for a, b, n in other:
    if b > g0 - 100000 and a < g1 + 100000: print("    %.3f .. %.3f  %s" % ((a - t_lo) / 1e6, (b - t_lo) / 1e6, n[:50]))
PY
