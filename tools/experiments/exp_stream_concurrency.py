#!/usr/bin/env python3
"""do kernels of different HIP streams run concurrently on this box?  torch.cuda._sleep spins one thread for n cycles"""
import time, torch
n = 5_000_000
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.empty(1 << 28, device="cuda")
def big(k=20):
    for _ in range(k): x.mul_(1.0001)
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    return best
def sleeps(two):
    with torch.cuda.stream(s1): torch.cuda._sleep(n)
    with torch.cuda.stream(s2 if two else s1): torch.cuda._sleep(n)
def big_and_sleep():
    with torch.cuda.stream(s1): big()
    with torch.cuda.stream(s2): torch.cuda._sleep(n)
def sleep_then_big():
    with torch.cuda.stream(s2): torch.cuda._sleep(n)
    with torch.cuda.stream(s1): big()
print("one sleep                : %.2f ms" % timed(lambda: torch.cuda._sleep(n)))
print("two sleeps on one stream : %.2f ms" % timed(lambda: sleeps(False)))
print("two sleeps on two streams: %.2f ms" % timed(lambda: sleeps(True)))
print("20 big kernels           : %.2f ms" % timed(big))
print("big on s1 then sleep on s2 (submission order): %.2f ms" % timed(big_and_sleep))
print("sleep on s2 then big on s1 (submission order): %.2f ms" % timed(sleep_then_big))
