import torch, time
for mb in (1, 8, 64, 256):
    h = torch.empty(mb << 20, dtype=torch.uint8, pin_memory=True)
    d = torch.empty(mb << 20, dtype=torch.uint8, device='cuda')
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = max(1, 512 // mb)
    for _ in range(n): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("H2D %4d MB chunks: %.1f GB/s" % (mb, n * mb / 1024 / dt))
# many 0.9 MB images (a 640x480 BGR frame) from separate pinned buffers, spread over 1 / 2 / 4 / 8 streams
imgs = [torch.empty(921600, dtype=torch.uint8, pin_memory=True) for _ in range(256)]
dst = torch.empty(256 * 921600, dtype=torch.uint8, device='cuda')
for ns in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    def go():
        for i, h in enumerate(imgs):
            with torch.cuda.stream(streams[i % ns]):
                dst[i * 921600:(i + 1) * 921600].copy_(h, non_blocking=True)
        torch.cuda.synchronize()
    go()
    t0 = time.perf_counter()
    for _ in range(4): go()
    dt = (time.perf_counter() - t0) / 4
    print("H2D 256 x 0.9 MB images on %d stream(s): %.2f ms  %.1f GB/s" % (ns, 1e3 * dt, 256 * 921600 / 1e9 / dt))
