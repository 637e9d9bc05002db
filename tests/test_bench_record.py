"""The committed measurement record must describe the committed kernels (CPU only, no GPU call): bench.py pastes roofline.traffic
from profiles/pmc_traffic.json only when that record carries the hash of the current kernel sources -- a kernel edit without a new
PMC pass on the GPU would silently turn the line's `traffic` into null at the end of the round."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_for_record_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)          # defines functions only; main() runs under __main__
    return m


def test_pmc_traffic_record_matches_the_kernel_sources():
    bench = _bench_module()
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    h = bench.kernel_source_hash()
    default_key = "640x480x4x10_b40000"                        # the default line's workload (bench.py --batch default)
    assert default_key in rec
    stale = [k for k, v in rec.items() if v.get("kernel_source_sha256") != h]
    assert not stale, ("profiles/pmc_traffic.json was measured on other kernel sources (%s, now %s): run "
                                      "tools/r04_record.sh traffic bench on the GPU box and copy gpurun_out/pmc_traffic.json" % (rec[default_key].get("kernel_source_sha256"), h))
    for k, v in rec.items():
        assert abs(v["hbm_bytes_per_launch"] - (v["fetch_bytes"] + v["write_bytes"])) <= 2, k
        if "l2_read_requests" in v and k not in stale:
            # a miss fetches a whole line (profiles/r04_line_fetch); both are means over launches, truncated to integers separately
            assert abs(v["fetch_bytes"] - 128 * v["l2_read_requests"]) <= max(128, 1e-6 * v["fetch_bytes"]), k


def test_committed_bench_line_keeps_the_contract():
    line = [l for l in open(os.path.join(ROOT, "profiles", "r04_final", "bench_driver_flags.json")) if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "aligns/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["peak"] == 8000.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert r["kernel_ms"] <= d["ms_per_step"]                                  # the launch is inside the step
    assert abs(d["value"] - d["config"]["pairs_per_gpu"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["sample"]
    assert d["parity_check"]["pass"] is True
