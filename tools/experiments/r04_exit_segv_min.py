"""Does a process that never loads the engine die the same way at exit under rocprofv3?  N extra streams with a little work each.
usage: r04_exit_segv_min.py <n_streams> [priority]"""
import sys
import torch
n = int(sys.argv[1])
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
x = torch.ones(1 << 20, device="cuda")
ss = [torch.cuda.Stream(priority=prio) for _ in range(n)]
ys = []
for s in ss:
    with torch.cuda.stream(s):
        ys.append(x * 2)
h = torch.empty(1 << 20, pin_memory=True)
for i, s in enumerate(ss):
    with torch.cuda.stream(s):
        ys[i].copy_(h, non_blocking=True)
torch.cuda.synchronize()
print("ok", n, prio, flush=True)
if len(sys.argv) > 3 and n >= 2:            # cross-stream event waits (what the frame path's upload pipeline does)
    ev = torch.cuda.Event(enable_timing=False)
    with torch.cuda.stream(ss[0]):
        z = x * 3
        ev.record(ss[0])
    ss[1].wait_event(ev)
    with torch.cuda.stream(ss[1]):
        w = z + 1
    torch.cuda.synchronize()
    print("cross-stream wait done", flush=True)
