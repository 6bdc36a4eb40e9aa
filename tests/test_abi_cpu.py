"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol that
include/ptcore.h declares, and fails loudly without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptcore.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from pathtracer_amd import ptcore
    lib = ptcore.load()
    names = declared_symbols()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ptcore.h but not exported"
    assert sorted(ptcore.EXPORTS) == names


def test_struct_layouts_match_header():
    from pathtracer_amd import ptcore, scenes
    assert ptcore.RAY_DTYPE.itemsize == 28 and ptcore.HIT_DTYPE.itemsize == 12 and ptcore.NODE_DTYPE.itemsize == 32
    assert scenes.MATERIAL_DTYPE.itemsize == 24
    assert C.sizeof(ptcore.CameraState) == 8 * 12 + 16
    assert C.sizeof(ptcore.Shard) == 20
    assert C.sizeof(ptcore.Counters) == 120


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pathtracer_amd import ptcore
    with pytest.raises(ptcore.PtError) as e:
        ptcore.PathTracer(0)
    assert e.value.status == ptcore.PT_ERR_NO_DEVICE
    lib = ptcore.load()
    h = C.c_void_p()
    assert lib.pt_create(C.c_int(-1), C.byref(h)) == ptcore.PT_ERR_NO_DEVICE  # no "CPU device"
    assert lib.pt_strerror(ptcore.PT_ERR_NO_DEVICE).decode() == "no usable HIP device"


def test_product_does_not_reference_oracle():
    """The product path may not import, include or link anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "pathtracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"oracle[/.]|liborc|from oracle|import oracle", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
