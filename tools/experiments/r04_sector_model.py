#!/usr/bin/env python3
"""Offline model of the look-up requests of one iteration of the packed kernel (compact now form, dvo_palette.h): how many
distinct 64-byte sectors / 128-byte lines the points of a level touch, per wave instruction (64 consecutive points of the
block-ordered list) and per level, for the bench's synthetic scenes at the true pose.  CPU only."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from rgbd_odometry_amd import SynthScene

W, H, NL = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480, 4)
seeds = [1000, 1001, 1002, 1003]
tot = {}
for seed in seeds:
    sc = SynthScene(W, H, NL, seed)
    for l, L in enumerate(sc.levels):
        rows, cols = L.rows, L.cols
        edge = L.ref_edge.reshape(cols, rows)          # [xx][yy]
        depth = L.ref_depth.reshape(cols, rows)
        xs, ys = np.nonzero((edge > 0) & (depth > 100.0))
        # block order: (xx>>4) outer, (yy>>4), then xx&15, then yy
        key = ((xs >> 4) * 4096 + (ys >> 4)) * 16 + (xs & 15)
        order = np.lexsort((ys, key))
        xs, ys = xs[order], ys[order]
        s = 2.0 ** -l
        fx, fy, cx, cy = sc.fx * s, sc.fy * s, sc.cx * s, sc.cy * s
        Z = depth[xs, ys] / 1000.0
        P = np.stack([Z * (xs - cx) / fx, Z * (ys - cy) / fy, Z])
        Pn = sc.R_true.T @ (P - sc.t_true[:, None])
        u = fx * Pn[0] / Pn[2] + cx
        v = fy * Pn[1] / Pn[2] + cy
        vis = (u >= 0) & (u < cols) & (v >= 0) & (v < rows)
        px, py = np.floor(u).astype(int), np.floor(v).astype(int)
        line = (px >> 2) * 100000 + py // 6
        sect = line * 2 + ((px >> 1) & 1)
        sq44 = (px >> 2) * 100000 + (py >> 2)           # a 4x4 sector without apron (64 B)
        sq84 = (px // 5) * 100000 + (py >> 4)           # byte ranks: 7 x 18 bytes per line, 5 x 16 interior
        sq44 = (px // 9) * 100000 + (py // 9)           # byte ranks: 11 x 11 bytes per line, 9 x 9 interior
        if l == 0 and seed == seeds[0]:
            for (w, h) in ((5, 16), (16, 5), (6, 14), (7, 12), (8, 10), (9, 9)):
                print("   level 0 interior %2d x %2d: %d lines" % (w, h, len(np.unique((px[vis] // w) * 100000 + py[vis] // h))))
        n = len(xs)
        d = tot.setdefault(l, dict(n=0, isect=0, iline=0, iboth=0, dsect=0, dline=0, d44=0, d84=0, i44=0, i84=0, pix=0))
        d["n"] += n
        for c0 in range(0, n, 64):
            m = vis[c0:c0 + 64]
            ss = np.unique(sect[c0:c0 + 64][m]); ll = np.unique(line[c0:c0 + 64][m])
            d["isect"] += len(ss); d["iline"] += len(ll)
            d["iboth"] += 2 * len(ll) - len(ss) if len(ss) else 0      # lines with both sectors touched by this instruction: len(ss) - len(ll)
            d["i44"] += len(np.unique(sq44[c0:c0 + 64][m])); d["i84"] += len(np.unique(sq84[c0:c0 + 64][m]))
        d["dsect"] += len(np.unique(sect[vis])); d["dline"] += len(np.unique(line[vis]))
        d["d44"] += len(np.unique(sq44[vis])); d["d84"] += len(np.unique(sq84[vis]))
        d["pix"] += len(np.unique(px[vis] * 100000 + py[vis]))
print("per level, averages over %d scenes (%dx%d)" % (len(seeds), W, H))
print("lvl  points  pixels | per instr: sectors lines (both-halves lines) | per level: sectors lines | bytes 9x9: instr level | bytes 5x16: instr level")
S = {}
for l, d in sorted(tot.items()):
    k = len(seeds)
    both = d["isect"] - d["iline"]
    print("%3d %7d %7d | %8d %6d (%5d) | %8d %6d | %6d %6d | %6d %6d" % (l, d["n"] / k, d["pix"] / k, d["isect"] / k, d["iline"] / k, both / k,
          d["dsect"] / k, d["dline"] / k, d["i44"] / k, d["d44"] / k, d["i84"] / k, d["d84"] / k))
    for key in d: S[key] = S.get(key, 0) + d[key] / k
print("sum %7d %7d | %8d %6d (%5d) | %8d %6d | %6d %6d | %6d %6d" % (S["n"], S["pix"], S["isect"], S["iline"], S["isect"] - S["iline"], S["dsect"], S["dline"], S["i44"], S["d44"], S["i84"], S["d84"]))
