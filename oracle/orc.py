"""ctypes binding of liborc.so, the CPU oracle.

TEST INFRASTRUCTURE ONLY: import from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never from pathtracer_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pt_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


RAY_DTYPE = np.dtype([("o", "<f4", (3,)), ("d", "<f4", (3,)), ("tmax", "<f4")])
HIT_DTYPE = np.dtype([("prim", "<u4"), ("t", "<f4"), ("inside", "<u4")])
NODE_DTYPE = np.dtype([("min", "<f4", (3,)), ("left_first", "<u4"), ("max", "<f4", (3,)), ("count", "<u4")])
MISS = 0xFFFFFFFF


class CameraState(C.Structure):
    _fields_ = [(n, C.c_float * 3) for n in ("origin", "target", "direction", "right", "up", "p1", "p2", "p3")] + [
        ("focal_distance", C.c_float), ("lens_size", C.c_float), ("width", C.c_uint32), ("height", C.c_uint32)]

    def as_dict(self):
        d = {n: np.array(getattr(self, n)[:], dtype=np.float32) for n in
             ("origin", "target", "direction", "right", "up", "p1", "p2", "p3")}
        d["focal_distance"] = np.float32(self.focal_distance)
        d["lens_size"] = np.float32(self.lens_size)
        d["width"], d["height"] = self.width, self.height
        return d


class Counters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_any", C.c_uint64),
                ("node_tests", C.c_uint64), ("prim_tests", C.c_uint64), ("extend_launches", C.c_uint64),
                ("extend_ms", C.c_double), ("render_ms", C.c_double), ("tail_rays", C.c_uint64),
                ("tail_node_tests", C.c_uint64), ("tail_prim_tests", C.c_uint64), ("tail_launches", C.c_uint64),
                ("tail_ms", C.c_double), ("extend_ms_sum", C.c_double), ("tail_ms_sum", C.c_double),
                ("redo_paths", C.c_uint64), ("redo_restarts", C.c_uint64)]   # pt_counters of include/ptcore.h, 136 bytes

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def as_rays(a) -> np.ndarray:
    """(n,7) float32 or RAY_DTYPE array -> contiguous RAY_DTYPE array."""
    a = np.asarray(a)
    if a.dtype == RAY_DTYPE:
        return np.ascontiguousarray(a)
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 7)
    return a.view(RAY_DTYPE).reshape(-1)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_num_lights.restype = C.c_uint64
        _lib.orc_tonemap.restype = C.c_uint8
        _lib.orc_tonemap.argtypes = [C.c_float]
    return _lib


class Oracle:
    MATH_DET, MATH_LIBM = 0, 1

    def __init__(self, math_mode: int = 0):
        self.l = lib()
        self.h = C.c_void_p(self.l.orc_create())
        self.l.orc_set_math(self.h, C.c_int(math_mode))
        self.width = self.height = 0

    def close(self):
        if self.h:
            self.l.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene
    def set_materials(self, mats):
        mats = np.ascontiguousarray(mats)
        assert mats.dtype.itemsize == 24
        self.l.orc_set_materials(self.h, _ptr(mats), C.c_uint32(len(mats)))

    def set_triangles(self, pos, nrm, mat_id):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 9)
        nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1, 9)
        mat_id = np.ascontiguousarray(mat_id, np.uint32)
        self.l.orc_set_triangles(self.h, _ptr(pos), _ptr(nrm), _ptr(mat_id), C.c_uint64(len(pos)))

    def set_spheres(self, cr, mat_id):
        cr = np.ascontiguousarray(cr, np.float32).reshape(-1, 4)
        mat_id = np.ascontiguousarray(mat_id, np.uint32)
        self.l.orc_set_spheres(self.h, _ptr(cr), _ptr(mat_id), C.c_uint64(len(cr)))

    def set_sky(self, sky):
        sky = np.ascontiguousarray(sky, np.float32)
        h, w = sky.shape[:2]
        self.l.orc_set_sky(self.h, _ptr(sky), C.c_uint32(w), C.c_uint32(h))

    def build(self) -> int:
        return self.l.orc_build(self.h)

    def bvh(self):
        n_nodes, n_idx, depth = C.c_uint64(), C.c_uint64(), C.c_uint32()
        self.l.orc_bvh_sizes(self.h, C.byref(n_nodes), C.byref(n_idx), C.byref(depth))
        nodes = np.zeros(n_nodes.value, dtype=NODE_DTYPE)
        idx = np.zeros(n_idx.value, dtype=np.uint64)
        self.l.orc_get_bvh(self.h, _ptr(nodes), _ptr(idx))
        return nodes, idx, depth.value

    def num_lights(self) -> int:
        return int(self.l.orc_num_lights(self.h))

    # ---- camera
    def set_camera(self, origin, target, width, height, lens_size=0.04):
        o = (C.c_float * 3)(*origin)
        t = (C.c_float * 3)(*target)
        rc = self.l.orc_set_camera(self.h, o, t, C.c_uint32(width), C.c_uint32(height), C.c_float(lens_size))
        assert rc == 0, rc
        self.width, self.height = width, height

    def camera(self):
        s = CameraState()
        self.l.orc_get_camera(self.h, C.byref(s))
        return s.as_dict()

    def generate(self, xy, sample, seed):
        xy = np.ascontiguousarray(xy, np.uint32).reshape(-1, 2)
        rays = np.zeros(len(xy), dtype=RAY_DTYPE)
        self.l.orc_generate(self.h, _ptr(xy), C.c_uint64(len(xy)), C.c_uint32(sample), C.c_uint64(seed), _ptr(rays))
        return rays

    # ---- seams
    def trace(self, rays, any_hit=False, with_counters=False):
        rays = as_rays(rays)
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        cnt = np.zeros(5, dtype=np.uint64)
        rc = self.l.orc_trace(self.h, _ptr(rays), C.c_uint64(len(rays)), C.c_int(1 if any_hit else 0), _ptr(hits), _ptr(cnt))
        assert rc == 0, rc
        if with_counters:
            return hits, dict(rays=int(cnt[0]), node_tests=int(cnt[1]), prim_tests=int(cnt[2]),
                              aabb_calls=int(cnt[3]), max_stack=int(cnt[4]))
        return hits

    def trace_stack_hist(self, rays):
        rays = as_rays(rays)
        hist = np.zeros(64, dtype=np.uint64)
        self.l.orc_trace_stack_hist(self.h, _ptr(rays), C.c_uint64(len(rays)), _ptr(hist))
        return hist

    def sample_skybox(self, dirs):
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros_like(dirs)
        self.l.orc_sample_skybox(self.h, _ptr(dirs), C.c_uint64(len(dirs)), _ptr(out))
        return out

    def sample_pixels(self, xy, sample, depth, seed, nee_mode=0):
        xy = np.ascontiguousarray(xy, np.uint32).reshape(-1, 2)
        out = np.zeros((len(xy), 3), dtype=np.float32)
        rc = self.l.orc_sample_pixels(self.h, _ptr(xy), C.c_uint64(len(xy)), C.c_uint32(sample), C.c_uint32(depth),
                                      C.c_uint64(seed), C.c_int(nee_mode), _ptr(out))
        assert rc == 0, rc
        return out

    # ---- frame loop
    def render(self, spp, depth=32, seed=0, sample_begin=0, nthreads=0, nee_mode=0):
        rc = self.l.orc_render(self.h, C.c_uint32(spp), C.c_uint32(depth), C.c_uint64(seed), C.c_uint32(sample_begin),
                               C.c_uint32(nthreads), C.c_int(nee_mode))
        assert rc == 0, rc

    def reset_accum(self):
        self.l.orc_reset_accum(self.h)

    def read_accum(self):
        out = np.zeros((self.height, self.width, 3), dtype=np.float32)
        spp = C.c_float()
        self.l.orc_read_accum(self.h, _ptr(out), C.byref(spp))
        return out, spp.value

    def read_rgb8(self):
        out = np.zeros((self.height, self.width, 3), dtype=np.uint8)
        self.l.orc_read_rgb8(self.h, _ptr(out), C.c_uint64(self.width * 3))
        return out

    def counters(self):
        c = Counters()
        hist = np.zeros(64, dtype=np.uint64)
        self.l.orc_get_counters(self.h, C.byref(c), _ptr(hist))
        d = c.as_dict()
        d["stack_hist"] = hist
        return d

    def reset_counters(self):
        self.l.orc_reset_counters(self.h)


# ---- unit seams -------------------------------------------------------------------------
def _ray(o, d, tmax=np.inf):
    r = np.zeros(1, dtype=RAY_DTYPE)
    r["o"][0], r["d"][0], r["tmax"][0] = o, d, tmax
    return r


def aabb_intersect(mn, mx, o, d, tmax=np.inf):
    mn = np.asarray(mn, np.float32)
    mx = np.asarray(mx, np.float32)
    r = _ray(o, d, tmax)
    t0, t1 = C.c_float(), C.c_float()
    hit = lib().orc_aabb_intersect(_ptr(mn), _ptr(mx), _ptr(r), C.byref(t0), C.byref(t1))
    return (t0.value, t1.value) if hit else None


def triangle_intersect(p, n, o, d, tmax=np.inf):
    p = np.ascontiguousarray(p, np.float32).reshape(9)
    n = np.ascontiguousarray(n, np.float32).reshape(9)
    r = _ray(o, d, tmax)
    out = np.zeros(9, np.float32)
    hit = lib().orc_triangle_intersect(_ptr(p), _ptr(n), _ptr(r), _ptr(out))
    if not hit:
        return None
    return dict(t=out[0], u=out[1], v=out[2], a=out[3], inside=bool(out[4]), normal=out[5:8].copy(), area=out[8])


def sphere_intersect(cr, o, d, tmax=np.inf):
    cr = np.ascontiguousarray(cr, np.float32).reshape(4)
    r = _ray(o, d, tmax)
    out = np.zeros(6, np.float32)
    hit = lib().orc_sphere_intersect(_ptr(cr), _ptr(r), _ptr(out))
    if not hit:
        return None
    return dict(t=out[0], inside=bool(out[1]), normal=out[2:5].copy(), area=out[5])


def philox(key, ctr):
    key = np.asarray(key, np.uint32)
    ctr = np.asarray(ctr, np.uint32)
    out = np.zeros(4, np.uint32)
    lib().orc_philox(_ptr(key), _ptr(ctr), _ptr(out))
    return out


def math_fn(fn, x, y=None, mode=0):
    """fn: 0 sin, 1 cos, 2 exp, 3 acos, 4 atan2(x, y), 5 cbrt."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(np.zeros_like(x) if y is None else y, np.float32)
    out = np.zeros_like(x)
    lib().orc_math(C.c_int(fn), C.c_int(mode), _ptr(x), _ptr(y), C.c_uint64(x.size), _ptr(out))
    return out


def tonemap(x) -> int:
    return int(lib().orc_tonemap(C.c_float(x)))


def from_scene(scene, sky=None, math_mode=0) -> Oracle:
    """scene: (cr, mat_id, mats) for spheres or (pos, nrm, mat_id, mats) for triangles."""
    o = Oracle(math_mode)
    if len(scene) == 3:
        cr, mat_id, mats = scene
        o.set_materials(mats)
        o.set_spheres(cr, mat_id)
    else:
        pos, nrm, mat_id, mats = scene
        o.set_materials(mats)
        o.set_triangles(pos, nrm, mat_id)
    if sky is not None:
        o.set_sky(sky)
    rc = o.build()
    assert rc == 0, f"oracle build failed: {rc}"
    return o
