#!/usr/bin/env python3
"""HBM-side traffic of the default bench launch from rocprofv3 PMC passes -> profiles/pmc_traffic.json.

Run on the GPU box (python3 tools/update_pmc_traffic.py [tag]).  Three separate, guarded --pmc passes (FETCH_SIZE,
WRITE_SIZE, TCC_EA0_RDREQ + TCC hit/req) over `bench.py --steps 3`, as MI355X_MICROARCH.md prescribes (counters in
their own runs, --kernel-trace only).  The record carries the hash of the kernel sources (bench.kernel_source_hash), so
bench.py reports `traffic: null` instead of a stale number once the kernel changes.
Read bytes = TCC_EA0_RDREQ x 128: on gfx950 EVERY L2 -> fabric read request is a whole 128-byte line, for sparse gathers as
for streams (round 4, tools/exhaustive/line_fetch.hip, profiles/r04_line_fetch: the other half of a line fetched for one
4-byte load is an L2 hit), while FETCH_SIZE tallies every request at 64 bytes.  FETCH_SIZE is kept in the record as reported
(rounds 1-3 took it at face value: half the real read traffic).
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
extra = sys.argv[2:]          # extra bench.py arguments, e.g. --batch 1024 (the record key follows them)
out_dir = os.path.join(ROOT, "gpurun_out", "pmc_traffic_" + tag)
os.makedirs(out_dir, exist_ok=True)
os.environ["TMPDIR"] = "/tmp"
groups = {"fetch": "FETCH_SIZE", "write": "WRITE_SIZE", "req": "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum"}
vals = {}
for name, cnt in groups.items():
    d = os.path.join(out_dir, name)
    cmd = ["timeout", "240", "rocprofv3", "--pmc"] + cnt.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc",
           "--", "python3", os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--no-extra-legs"] + extra
    r = subprocess.run(cmd, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    open(os.path.join(out_dir, name + ".log"), "w").write(r.stdout)
    print(name, "rc", r.returncode)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "align_fused" in row["Kernel_Name"]:
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in vals.items()}
print(avg)
import bench
fetch_reported, write = avg["FETCH_SIZE"] * 1024.0, avg["WRITE_SIZE"] * 1024.0      # rocprofv3 reports KiB
fetch = avg["TCC_EA0_RDREQ_sum"] * 128.0       # whole lines (see the module docstring)
rec = {
    "hbm_bytes_per_launch": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write),
    "fetch_size_as_reported_bytes": int(fetch_reported),
    "l2_read_requests": int(avg["TCC_EA0_RDREQ_sum"]), "l2_requests": int(avg.get("TCC_REQ_sum", 0)), "l2_hits": int(avg.get("TCC_HIT_sum", 0)),
    "kernel_source_sha256": bench.kernel_source_hash(),
    "source": ("tools/update_pmc_traffic.py %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ, separate passes over bench.py "
               "--steps 3 %s; read bytes = TCC_EA0_RDREQ x 128: every L2 -> fabric read is a whole line, profiles/r04_line_fetch; "
               "FETCH_SIZE tallies them at 64 bytes)") % (tag, " ".join(extra)),
}
path = os.path.join(ROOT, "gpurun_out", "pmc_traffic.json")
allrec = {}
for src in (os.path.join(ROOT, "profiles", "pmc_traffic.json"), path):      # several calls of one GPU session accumulate
    try:
        allrec.update(json.load(open(src)))
    except Exception:
        pass
sys.argv = ["bench.py"] + extra
a = bench.parse_args()
key = ("tiled_%dx%dx%dx%d" % (a.width, a.height, a.levels, a.iters)) if a.mode == "tiled" else "%dx%dx%dx%d_b%d" % (a.width, a.height, a.levels, a.iters, a.batch)
allrec[key] = rec
# records measured on other kernel sources describe a kernel that no longer exists: drop them (tests/test_bench_record.py)
allrec = {k: v for k, v in allrec.items() if v.get("kernel_source_sha256") == rec["kernel_source_sha256"]}
json.dump(allrec, open(path, "w"), indent=2)
print("wrote", path, "(copy to profiles/pmc_traffic.json)")
print(json.dumps(rec, indent=2))
