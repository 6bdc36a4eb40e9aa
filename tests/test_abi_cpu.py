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
    assert C.sizeof(ptcore.Counters) == 136
    from oracle import orc
    assert C.sizeof(orc.Counters) == 136   # the oracle fills the header's pt_counters too


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


def test_header_is_plain_c(tmp_path):
    """include/ptcore.h is the boundary a C, cgo or bindgen caller sees: it must compile as strict C99 on its own,
    and the structs that cross it keep the sizes the bindings (bindings/rust/ptcore-sys, pathtracer_amd/ptcore.py) assume."""
    import os
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('''#include "ptcore.h"
#include <stdio.h>
int main(void) {
    printf("%u %u %u %u %u\\n", (unsigned)sizeof(pt_ray), (unsigned)sizeof(pt_hit), (unsigned)sizeof(pt_bvh_node),
           (unsigned)sizeof(pt_material), (unsigned)sizeof(pt_shard));
    return 0;
}
''')
    exe = str(tmp_path / "abi")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", exe], check=True)
    sizes = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    import ctypes
    from pathtracer_amd import ptcore as pc, scenes
    assert sizes == [28, pc.HIT_DTYPE.itemsize, pc.NODE_DTYPE.itemsize, scenes.MATERIAL_DTYPE.itemsize, ctypes.sizeof(pc.Shard)]


def test_fastdiv_is_exact(tmp_path):
    """pt_pixelmap.h FastDiv: floor(n / d) for n < 2^31 as multiply-high + shift (the slot -> pixel mapping of the kernels
    and of pt_shard_map).  Every d up to 5000, powers of two and their neighbours, random d; n at 0, around multiples
    of d, at 2^31 - 1 and random."""
    import os
    import subprocess
    src = tmp_path / "fd.cpp"
    src.write_text(r'''
#include "pt_pixelmap.h"
#include <cstdio>
#include <random>
using namespace pt;
static uint32_t fd(uint32_t n, FastDiv f) { return f.m ? (uint32_t)(((uint64_t)n * f.m) >> 32) >> f.sh : n; }
int main() {
    std::mt19937_64 rng(1);
    unsigned long long bad = 0, checked = 0;
    auto check = [&](uint32_t d) {
        FastDiv f = fastdiv_make(d);
        uint32_t ns[] = {0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 0x7fffffffu, 0x7ffffffeu, 0x7fffffffu / d * d, 0x7fffffffu / d * d - 1};
        for (uint32_t n : ns) if (n < 0x80000000u) { bad += fd(n, f) != n / d; ++checked; }
        for (int k = 0; k < 300; ++k) { uint32_t n = (uint32_t)(rng() & 0x7fffffffu); bad += fd(n, f) != n / d; ++checked; }
        for (int q = 0; q < 100; ++q) { uint64_t n = (uint64_t)(rng() % (0x80000000ull / d + 1)) * d; for (int o = -1; o <= 1; ++o) { uint64_t m = n + o; if (m < 0x80000000ull) { bad += fd((uint32_t)m, f) != (uint32_t)m / d; ++checked; } } }
    };
    for (uint32_t d = 1; d <= 5000; ++d) check(d);
    for (int k = 0; k < 20000; ++k) check((uint32_t)(rng() % 0x7fffffffu) + 1);
    for (int l = 0; l < 31; ++l) { check(1u << l); check((1u << l) + 1); if (l) check((1u << l) - 1); }
    std::printf("%llu %llu\n", checked, bad);
    return bad != 0;
}
''')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fd")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "pathtracer_amd", "csrc"), str(src), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.split()[1] == "0", r.stdout
