#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/trace_fused; rm -rf $O; mkdir -p $O
python3 $R/tools/trace_frames_fused.py fused 2>&1 | grep rep
python3 $R/tools/trace_frames_fused.py split 2>&1 | grep rep
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- python3 $R/tools/trace_frames_fused.py fused > $O/log.txt 2>&1
ls $O
python3 - <<PY
import csv,glob
O="$O"
ev=[]
for f in glob.glob(O+"/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"K",r["Kernel_Name"].split("(")[0][-30:]))
for f in glob.glob(O+"/**/*memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"C",r.get("Direction","")))
ev.sort()
# last align kernel marks the end of a rep; take the window between the 3rd and 4th align kernels
al=[e for e in ev if "align_fused" in e[3]]
print("n events",len(ev),"aligns",len(al))
a0,a1=al[-2][1],al[-1][1]
win=[e for e in ev if e[0]>=a0 and e[1]<=a1]
span=(a1-a0)/1e6
kb=sum(e[1]-e[0] for e in win if e[2]=="K")/1e6
cb=sum(e[1]-e[0] for e in win if e[2]=="C")/1e6
print("window ms %.2f kernel busy %.2f copy busy (sum over queues) %.2f n_copies %d"%(span,kb,cb,sum(1 for e in win if e[2]=="C")))
cs=[e for e in win if e[2]=="C"]
print("copies span ms %.2f first at %.2f last end %.2f"%((cs[-1][1]-cs[0][0])/1e6,(cs[0][0]-a0)/1e6,(cs[-1][1]-a0)/1e6))
ks=[e for e in win if e[2]=="K"]
# kernel timeline coarse: print every kernel > 100us and gaps > 200us
prev=a0
for e in ks:
    gap=(e[0]-prev)/1e3
    if gap>150: print("  gap %.0f us before %s at %.2f ms"%(gap,e[3],(e[0]-a0)/1e6))
    prev=max(prev,e[1])
PY
