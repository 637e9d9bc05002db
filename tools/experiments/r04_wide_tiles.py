"""How often would a one-byte-per-pixel now form (tile-local ranks, 7 x 18 stored pixels per 128-byte line) meet a tile whose
ranks span more than a byte?  Edge maps of random 60-pixel segments at several densities, exact distance transform, ranks of the
distinct squared distances (what the compact form stores), rank span per tile.  CPU only (scipy).  -> DESIGN.md section 6."""
import numpy as np
from scipy.ndimage import distance_transform_edt

rng = np.random.default_rng(0)
H, W = 480, 640
for dens in (0.2, 0.05, 0.02, 0.005):
    e = np.zeros((H, W), bool)
    for _ in range(int(dens * H * W / 60)):
        y, x, a = rng.integers(0, H), rng.integers(0, W), rng.uniform(0, np.pi)
        t = np.arange(60)
        e[(y + t * np.sin(a)).astype(int) % H, (x + t * np.cos(a)).astype(int) % W] = True
    d = distance_transform_edt(~e)
    d2 = np.rint(d * d).astype(np.int64)
    vals, rk = np.unique(d2, return_inverse=True)
    rk = rk.reshape(H, W)
    wide, tot, nearest = 0, 0, []
    for ty in range(0, H, 5):
        for tx in range(0, W, 16):
            ys, xs = slice(max(ty - 1, 0), min(ty + 6, H)), slice(max(tx - 1, 0), min(tx + 17, W))
            tot += 1
            if rk[ys, xs].max() - rk[ys, xs].min() > 254:
                wide += 1
                nearest.append(np.sqrt(d2[ys, xs].min()))
    print(f"edge pixels {e.mean():.3f}  palette {len(vals):5d}  wide tiles {wide:4d}/{tot} = {wide / tot:.3f}"
          f"  nearest edge of a wide tile >= {min(nearest) if nearest else float('nan'):.1f} px  mean distance {d.mean():.1f} px")
