#!/usr/bin/env python3
"""Text dump of tools/ref_dump/ref_dump.cpp -> tests/golden/reference_golden.npz (keys as in oracle_golden.npz).

    python tools/ref_dump/to_npz.py <outputs.txt> tests/golden/reference_golden.npz
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


if __name__ == "__main__":
    d = parse(sys.argv[1])
    np.savez_compressed(sys.argv[2], **d)
    print("wrote %d arrays to %s" % (len(d), sys.argv[2]))
