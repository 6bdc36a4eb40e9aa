"""The boundary's declarations stay in step: include/ptcore.h is valid C (not only C++), and the Rust `-sys` crate a
maintainer of the reference would depend on (bindings/rust/ptcore-sys, not compilable here: no rustc in the image)
declares every entry point of the header with the same number of arguments and mirrors the struct sizes."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ptcore.h")
RUST = os.path.join(ROOT, "bindings", "rust", "ptcore-sys", "src", "lib.rs")


def _c_prototypes():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"\b(pt_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def _rust_prototypes():
    text = re.sub(r"//[^\n]*", "", open(RUST).read())
    out = {}
    for m in re.finditer(r"pub fn (pt_[a-z0-9_]+)\s*\(([^)]*)\)", text):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else args.count(":")
    return out


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "use_header.c"
    src.write_text('#include "ptcore.h"\n#include "ptcore_detmath.h"\n'
                   "int main(void) { pt_shard s = {PT_SHARD_ALL, 0, 1, 0, 0}; pt_counters c; (void)c;\n"
                   "  return (int)s.kind + (sizeof(pt_ray) == 28 && sizeof(pt_hit) == 12 && sizeof(pt_bvh_node) == 32 && sizeof(pt_material) == 24\n"
                   "                        && sizeof(pt_camera_state) == 112 && sizeof(pt_shard) == 20 && sizeof(pt_counters) == 136 ? 0 : 1); }\n")
    exe = tmp_path / "use_header"
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                        str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([str(exe)]).returncode == 0


def test_detmath_compiled_as_c_gives_the_same_bits(tmp_path):
    """include/ptcore_detmath.h in a plain C host (gcc -std=c11) against the C++ build inside the oracle library."""
    import numpy as np
    from oracle import orc
    xs = np.array([0.0, 0.3, 1.0, 2.5, 3.1415927, 4.7, 6.2831855, -0.75, -3.3], np.float32)
    src = tmp_path / "dm.c"
    src.write_text('#include <stdio.h>\n#include "ptcore_detmath.h"\nint main(void) {\n  const float xs[] = {%s};\n'
                   "  for (int i = 0; i < %d; ++i) { float s, c; pt_sincosf(xs[i], &s, &c);\n"
                   '    printf("%%08x %%08x %%08x %%08x %%08x\\n", pt_f2u(s), pt_f2u(c), pt_f2u(pt_expf(-xs[i] * xs[i])), pt_f2u(pt_acosf(xs[i] * 0.15f)), pt_f2u(pt_atan2f(xs[i], 1.5f - xs[i]))); }\n'
                   "  return 0; }\n" % (", ".join("%.9ef" % x for x in xs), len(xs)))
    exe = tmp_path / "dm"
    r = subprocess.run(["gcc", "-std=c11", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.array([[int(w, 16) for w in ln.split()] for ln in subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines()], np.uint32)
    want = np.stack([orc.math_fn(0, xs), orc.math_fn(1, xs), orc.math_fn(2, (-xs * xs).astype(np.float32)), orc.math_fn(3, (xs * np.float32(0.15)).astype(np.float32)),
                     orc.math_fn(4, xs, (np.float32(1.5) - xs).astype(np.float32))], axis=1).view(np.uint32)
    assert np.array_equal(got, want)


def test_rust_sys_crate_declares_every_entry_point():
    c, rust = _c_prototypes(), _rust_prototypes()
    assert len(c) >= 36
    assert sorted(c) == sorted(rust), set(c) ^ set(rust)
    for name, n in c.items():
        assert rust[name] == n, (name, n, rust[name])


def test_rust_struct_sizes_match():
    """Field-by-field size of the #[repr(C)] mirrors (all fields are 4 or 8 bytes wide and naturally aligned)."""
    text = open(RUST).read()
    width = {"u32": 4, "f32": 4, "u64": 8, "f64": 8}
    want = {"pt_material": 24, "pt_ray": 28, "pt_hit": 12, "pt_bvh_node": 32, "pt_camera_state": 112, "pt_shard": 20, "pt_counters": 136}
    for name, size in want.items():
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, text, flags=re.S).group(1)
        total = 0
        for ty in re.findall(r"pub \w+: ([^,\n]+),", body):
            m = re.fullmatch(r"\[(\w+); (\d+)\]", ty.strip())
            total += width[m.group(1)] * int(m.group(2)) if m else width[ty.strip()]
        assert total == size, (name, total, size)
