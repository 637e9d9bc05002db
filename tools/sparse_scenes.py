#!/usr/bin/env python3
"""The `sparse_scenes` leg of bench.py on its own (round 5): scenes with 0.5 / 1 / 2 % edge pixels confined to the left half of the
frame, at 640x480x4, 1920x1080x5 and 4096x3072x5 -- which levels get the compact form, which are refused and why, aligns/s, roofline
fraction.  usage: python tools/sparse_scenes.py [--quick] > gpurun_out/sparse_scenes.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
stream = torch.cuda.Stream()
res = bench.sparse_scenes_leg(stream, quick="--quick" in sys.argv)
print(json.dumps(res, indent=1))
for r in res:
    if "error" in r:
        print("#", r["workload"], "ERROR", r["error"], file=sys.stderr); continue
    print("# %-70s %9.0f aligns/s  alg. rate / HBM peak %.3f  refused %d partial %d of %d %s  info %s  fallback %s" % (r["workload"], r["aligns_per_s"], r["algorithmic_rate_over_hbm_peak"],
          r["levels_refused"], r.get("levels_partial", 0), r["levels_total"], r["refusal_reasons"], r["compact_info_per_level_scene0"],
          r.get("exact_fallback_ran_pair0")), file=sys.stderr)
