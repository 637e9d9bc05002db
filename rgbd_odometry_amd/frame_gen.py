"""Synthetic camera frames (BGR8 + depth in metres) for the frame-preprocessing rows f1/f2.

numpy only, seeded; piecewise-constant shapes (rectangles, discs, thick lines) over a smooth background plus
mild noise, so that Canny(150,100) finds a few percent of edge pixels like a real indoor frame.  A "now" frame
is the reference image shifted by a few pixels (what a small camera rotation does to first order)."""
import numpy as np


def camera_frame(seed: int, rows: int = 480, cols: int = 640, shift=(0, 0), noise: int = 3, holes: bool = True):
    rng = np.random.default_rng(seed)
    R, Cc = rows + 32, cols + 32                       # margin so that shifted crops stay inside
    yy, xx = np.mgrid[0:R, 0:Cc]
    img = np.empty((R, Cc, 3), np.float32)
    for ch in range(3):
        img[..., ch] = 110 + 40 * np.sin(xx / (37.0 + 5 * ch)) * np.cos(yy / (29.0 + 3 * ch))
    n_shapes = max(6, (rows * cols) // 4000)
    for _ in range(n_shapes):
        kind = rng.integers(0, 3)
        col = rng.integers(0, 256, 3).astype(np.float32)
        cy, cx = rng.integers(0, R), rng.integers(0, Cc)
        if kind == 0:
            h, w = rng.integers(6, max(8, R // 4)), rng.integers(6, max(8, Cc // 4))
            img[max(cy - h // 2, 0):cy + h // 2 + 1, max(cx - w // 2, 0):cx + w // 2 + 1] = col
        elif kind == 1:
            r = rng.integers(4, max(6, R // 6))
            m = (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
            img[m] = col
        else:
            ang = rng.uniform(0, np.pi)
            d = np.abs((xx - cx) * np.sin(ang) - (yy - cy) * np.cos(ang))
            ln = np.abs((xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang))
            img[(d <= rng.integers(1, 4)) & (ln <= rng.integers(10, max(12, Cc // 3)))] = col
    if noise:
        img += rng.integers(-noise, noise + 1, img.shape).astype(np.float32)
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    depth = (2.0 + 0.3 * np.sin(xx / 40.0) + 0.2 * np.cos(yy / 30.0)).astype(np.float32)
    if holes:
        hole = rng.random((R, Cc)) < 0.01
        depth[hole] = 0.0
        depth[rng.random((R, Cc)) < 0.002] = np.nan
    oy, ox = 16 + int(shift[0]), 16 + int(shift[1])
    return np.ascontiguousarray(img[oy:oy + rows, ox:ox + cols]), np.ascontiguousarray(depth[oy:oy + rows, ox:ox + cols])
