"""CPU checks of the C-ABI library: it loads, exports every symbol include/dvo_amd.h declares, its
parameter defaults are the reference's literals, and without a HIP device it fails loudly (no fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "dvo_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dvo_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from rgbd_odometry_amd import capi
    assert _declared_functions() == sorted(capi.C_ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    from rgbd_odometry_amd import capi
    lib = capi.load_library()
    for name in _declared_functions():
        assert hasattr(lib, name), name


def test_param_defaults_are_the_reference_literals():
    from rgbd_odometry_amd import capi
    lib = capi.load_library()
    p = capi.DvoParams()
    assert lib.dvo_params_default(ctypes.byref(p)) == 0
    assert p.beta == 0.5 and p.precond_rot == 0.5 and p.reg_lambda == 0.05          # :653, :725, :742
    assert p.step_a == 9.0 and p.step_b == 1.0e-2                                   # :773
    assert (p.step_decay_after, p.step_decay_offset) == (5, 4)                      # :773
    assert p.trust_radius == np.float32(0.003) and p.psi_norm_stop == np.float32(1.0e-7)   # :24-25 (float members)
    assert p.enable_rotationize == 1 and p.enable_l2_reg == 1 and p.interpolate_dt == 0    # SolveDVO.h:97,107,112


def test_struct_layout_matches_header():
    """the ctypes mirror must have the size the C compiler gives struct dvo_params"""
    from rgbd_odometry_amd import capi
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write('#include <stdio.h>\n#include <stddef.h>\n#include "dvo_amd.h"\nint main(){printf("%zu %zu %zu %zu", '
                             'sizeof(dvo_params), sizeof(dvo_image), offsetof(dvo_image, dtype), offsetof(dvo_params, canny_threshold1));return 0;}')
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        size, img_size, img_dtype_off, canny_off = (int(x) for x in subprocess.check_output([exe]).split())
    assert ctypes.sizeof(capi.DvoParams) == size
    assert ctypes.sizeof(capi.DvoImage) == img_size and capi.DvoImage.dtype.offset == img_dtype_off
    assert capi.DvoParams.canny_threshold1.offset == canny_off


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    from rgbd_odometry_amd import DvoContext, DvoError
    from rgbd_odometry_amd.capi import DVO_ERR_NO_DEVICE
    with pytest.raises(DvoError) as ei:
        DvoContext(1)
    assert ei.value.code == DVO_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_product_does_not_reference_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "rgbd_odometry_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in text and "dvo_oracle" not in text and "libdvo_oracle" not in text, f
