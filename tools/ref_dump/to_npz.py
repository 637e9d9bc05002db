#!/usr/bin/env python3
"""Text dumps of the two drivers of tools/ref_dump -> the .npz files the skipping tests read.

    python tools/ref_dump/to_npz.py <outputs.txt> tests/golden/reference_golden.npz                    (ref_dump.cpp: runIterations)
    python tools/ref_dump/to_npz.py --frames <outputs.txt> tests/golden/reference_frames_golden.npz    (frames_dump.cpp: row f1)
"""
import sys

import numpy as np


def parse(path):
    out = {}
    toks = open(path).read().split("\n")
    i, case = 0, None
    while i < len(toks):
        w = toks[i].split()
        i += 1
        if not w:
            continue
        if w[0] == "case":
            case = w[1]
        elif w[0] == "level":
            l, n, best, ratio = int(w[1]), int(w[2]), int(w[3]), float.fromhex(w[4])
            out[f"{case}_L{l}_energy"] = np.array([float.fromhex(x) for x in toks[i].split()], np.float32); i += 1
            assert len(out[f"{case}_L{l}_energy"]) == n
            out[f"{case}_L{l}_best"] = np.array(best, np.int32)
            out[f"{case}_L{l}_ratio"] = np.array(ratio, np.float32)
            last = l
        elif w[0] == "final":
            head = int(w[1])
            eps = np.array([float.fromhex(x) for x in toks[i].split()], np.float32); i += 1
            rep = np.array([float.fromhex(x) for x in toks[i].split()], np.float32).reshape(head, 3); i += 1
            out[f"{case}_final_eps_head"], out[f"{case}_final_reproj_head"] = eps, rep      # the last level run overwrites: level 0
        elif w[0] == "pose":
            v = np.array([float.fromhex(x) for x in w[1:]], np.float64)
            out[f"{case}_R"] = v[:9].reshape(3, 3, order="F")
            out[f"{case}_t"] = v[9:12]
    return out


def parse_frames(path):
    """frames_dump.cpp's text: per frame and level the edge map (run-length, column-major int32 0 / 255) and dt / gx / gy (float32)"""
    out, frame, level, shape = {}, None, None, None
    for line in open(path):
        w = line.split()
        if not w:
            continue
        if w[0] == "frame":
            frame = w[1]
            out[f"{frame}_levels"] = np.array(int(w[2]), np.int32)
        elif w[0] == "level":
            level, shape = int(w[1]), (int(w[2]), int(w[3]))
            out[f"{frame}_L{level}_shape"] = np.array(shape, np.int32)
        elif w[0] == "edge":
            v = np.array([int(x) for x in w[1:]], np.int64).reshape(-1, 2)
            e = np.repeat(v[:, 0], v[:, 1]).astype(np.int32)
            assert e.size == shape[0] * shape[1], (frame, level, e.size, shape)
            out[f"{frame}_L{level}_edge"] = e
        elif w[0] in ("dt", "gx", "gy"):
            a = np.array([float.fromhex(x) for x in w[1:]], np.float32)
            assert a.size == shape[0] * shape[1], (frame, level, w[0], a.size, shape)
            out[f"{frame}_L{level}_{w[0]}"] = a
    return out


if __name__ == "__main__":
    frames = len(sys.argv) > 1 and sys.argv[1] == "--frames"
    if frames:
        del sys.argv[1]
    d = parse_frames(sys.argv[1]) if frames else parse(sys.argv[1])
    np.savez_compressed(sys.argv[2], **d)
    print("wrote %d arrays to %s" % (len(d), sys.argv[2]))
