"""The register budget of the fused alignment kernels, read from the code objects inside lib/libdvo_amd.so.

The throughput configuration of the packed kernel (rgbd_odometry_amd/csrc/dvo_fused.hip) runs two waves per SIMD -- one
512-thread workgroup or two 256-thread workgroups per compute unit -- which gives a wave at most 256 vector registers.  A
kernel that needs one more silently halves its occupancy (or spills to scratch), so the budget is pinned here.  CPU test: the
numbers are in the ELF notes of the embedded gfx950 code object (llvm-readelf from the ROCm LLVM).
"""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "rgbd_odometry_amd", "lib", "libdvo_amd.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def _kernel_resources(tmp_path):
    data = open(LIB, "rb").read()
    out = {}
    for m in re.finditer(b"\x7fELF\x02\x01\x01", data):
        o = m.start()
        if o == 0 or struct.unpack_from("<H", data, o + 18)[0] != 224:        # EM_AMDGPU
            continue
        shoff = struct.unpack_from("<Q", data, o + 0x28)[0]
        shentsize, shnum = struct.unpack_from("<HH", data, o + 0x3A)
        f = tmp_path / ("co_%d.elf" % o)
        f.write_bytes(data[o:o + shoff + shentsize * shnum])
        notes = subprocess.run([READELF, "--notes", str(f)], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
            out[name.group(1)] = dict(vgpr=get("vgpr_count"), vgpr_spill=get("vgpr_spill_count"),
                                      agpr=int(re.match(r"\s*(\d+)", blk).group(1)), scratch=get("private_segment_fixed_size"))
    return out


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(READELF)), reason="library or llvm-readelf missing")
def test_packed_kernel_fits_two_waves_per_simd(tmp_path):
    res = _kernel_resources(tmp_path)
    fused2 = {k: v for k, v in res.items() if "align_fused2_kernel" in k}
    assert len(fused2) >= 7, sorted(res)[:10]
    for block in (256, 512):
        for team in (0, 1):
            k = [n for n in fused2 if "ILi%dELb%dELb0E" % (block, team) in n]
            assert len(k) == 1, k
            r = fused2[k[0]]
            assert r["vgpr"] + r["agpr"] <= 256, (block, r)       # two waves per SIMD (512 registers per lane and SIMD)
            # round 5: the update's constants live in LDS (dvo_device_math.h: UpdConst) instead of being hoisted into registers for
            # the whole kernel -- that hoisting was what spilled in rounds 2-4 (108 bytes of scratch on the throughput shape): none now
            assert r["scratch"] == 0 and r["vgpr_spill"] == 0, (block, team, r)
    # DVO_FLAG_NORMAL_MATRIX on the packed kernel (512 threads): 42 more accumulator registers.  Loop-invariant values may sit in
    # scratch (stored at kernel start, reloaded once per level or before the final pass); what must never happen is a scratch
    # access inside a loop over points: test_no_scratch_access_inside_the_point_loops covers this instantiation too
    for block in (256, 512):                                  # the 256-thread shape with H is auto-selected for large batches too
        k = [n for n in fused2 if "ILi%dELb0ELb1E" % block in n]
        assert len(k) == 1, (block, k)
        r = fused2[k[0]]
        assert r["vgpr"] + r["agpr"] <= 256 and r["scratch"] <= 320, (block, r)


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(READELF)), reason="library or llvm-readelf missing")
def test_step_launch_kernels_of_the_tiled_schedule_do_not_spill(tmp_path):
    """round 5: the packed point loop inside the step launch of the tiled / wide schedule, and the one-launch form of a small level:
    512 threads, no scratch (one workgroup per CU, or a single workgroup: the register budget is 256, neither needs half of it)"""
    res = _kernel_resources(tmp_path)
    for frag, budget in (("tiled_step_pk_kernelILb0E", 128), ("tiled_step_pk_kernelILb1E", 256), ("tiled_level_solo_kernel", 256)):      # a single workgroup: only "no scratch" matters
        k = [n for n in res if frag in n]
        assert len(k) == 1, (frag, k)
        r = res[k[0]]
        assert r["scratch"] == 0 and r["vgpr_spill"] == 0 and r["vgpr"] + r["agpr"] <= budget, (frag, r)       # with H: 42 more accumulators


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="library or llvm-objdump missing")
def test_no_scratch_access_inside_the_point_loops(tmp_path):
    """Disassembles the gfx950 code object and checks, for the throughput instantiations of the packed kernel, that no
    scratch_load / scratch_store sits inside a loop body of up to 8 KB of code (a backward branch and its target): the
    per-point loops are 3-6 KB each, the level / iteration loops that legitimately reload spilled loop-invariant values span
    tens of KB.  A spill inside a point loop would be paid per round of points."""
    data = open(LIB, "rb").read()
    text = ""
    for m in re.finditer(b"\x7fELF\x02\x01\x01", data):          # one embedded code object per .hip source
        o = m.start()
        if o == 0 or struct.unpack_from("<H", data, o + 18)[0] != 224:
            continue
        shoff = struct.unpack_from("<Q", data, o + 0x28)[0]
        shentsize, shnum = struct.unpack_from("<HH", data, o + 0x3A)
        f = tmp_path / ("co_%d.elf" % o)
        f.write_bytes(data[o:o + shoff + shentsize * shnum])
        text += subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(f)], capture_output=True, text=True, check=True).stdout
    assert "align_fused2_kernel" in text
    cur, kernels = None, {}
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        m = re.search(r"^\s+(\S+).*// ([0-9A-F]+):", line)
        if cur and m:
            tgt = re.search(r"<%s\+0x([0-9a-f]+)>" % re.escape(cur), line)
            kernels[cur].append((int(m.group(2), 16), m.group(1), tgt.group(1) if tgt else None))
    checked = 0
    for name, ins in kernels.items():
        if "align_fused2_kernel" not in name or not ins or "ILi1024E" in name:
            continue
        base = ins[0][0]
        loops = [(base + int(t, 16), a) for a, op, t in ins if op.startswith("s_cbranch") and t is not None and base + int(t, 16) <= a]
        inner = [(lo, hi) for lo, hi in loops if hi - lo <= 8192]
        scratch = [a for a, op, _ in ins if op.startswith("scratch_")]
        # a backward branch is not always a loop: the compiler also lays shared tail blocks out BEFORE the code that jumps to them.
        # A scratch access counts only if it lies on a cycle of the region: reachable from the branch target and reaching the
        # backward branch, both without leaving [lo, hi].
        index = {a: i for i, (a, _, _) in enumerate(ins)}

        def reaches(src, dst, lo, hi):
            seen, todo = set(), [src]
            while todo:
                a = todo.pop()
                if a == dst:
                    return True
                if a in seen or not (lo <= a <= hi) or a not in index:
                    continue
                seen.add(a)
                i = index[a]
                _, op, t = ins[i]
                if t is not None and (op.startswith("s_cbranch") or op == "s_branch"):
                    todo.append(base + int(t, 16))
                if op not in ("s_branch", "s_endpgm", "s_setpc_b64") and i + 1 < len(ins):
                    todo.append(ins[i + 1][0])
            return False

        bad = [(hex(a), (hex(lo), hex(hi))) for a in scratch for lo, hi in inner
               if lo <= a <= hi and reaches(lo, a, lo, hi) and reaches(a, hi, lo, hi)]
        assert sum(1 for lo, hi in inner if hi - lo >= 2500) >= 4, (name, len(loops))            # the point loops were found at all
        assert not bad, (name, bad[:5])
        checked += 1
    assert checked >= 4
