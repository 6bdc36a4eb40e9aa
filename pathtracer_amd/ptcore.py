"""ctypes binding of libptcore.so (include/ptcore.h) — the MI355X path-tracing core.

Host-side mirror of the reference's Scene / Camera / frame-loop surface for the hot path:

    reference (Rust)                                   here
    Scene::<Sphere>::default_scene()  scene.rs:98      PathTracer.set_spheres(...) + build()
    Scene::<Triangle>::scene(path)    scene.rs:35      PathTracer.set_triangles(...) + build()
    Camera::new(w, h, scene)          camera.rs:89     PathTracer.set_camera(origin, target, w, h)
    camera.generate / camera.sample   main.rs:157-159  PathTracer.render(spp, max_depth=32)
    accum / vec_to_rgb                main.rs:100,160  PathTracer.read_accum() / read_rgb8()
    bvh.intersect_closest / _any      bvh.rs:180,227   PathTracer.trace(rays, any_hit)

There is no CPU fallback: loading fails loudly if the HIP library is missing, and
PathTracer() raises if no gfx950 device is usable.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptcore.so")

RAY_DTYPE = np.dtype([("o", "<f4", (3,)), ("d", "<f4", (3,)), ("tmax", "<f4")])
HIT_DTYPE = np.dtype([("prim", "<u4"), ("t", "<f4"), ("inside", "<u4")])
NODE_DTYPE = np.dtype([("min", "<f4", (3,)), ("left_first", "<u4"), ("max", "<f4", (3,)), ("count", "<u4")])
MISS = 0xFFFFFFFF

PT_OK = 0
PT_ERR_INVALID, PT_ERR_STATE, PT_ERR_NO_DEVICE, PT_ERR_HIP, PT_ERR_LIMIT, PT_ERR_BVH = -1, -2, -3, -4, -5, -6
SHARD_ALL, SHARD_TILES = 0, 1
RENDER_TIMED, RENDER_COUNT = 1, 2
OPT_BATCH_PATHS, OPT_LANES, OPT_TAIL_MIN_PATHS, OPT_TAIL_FRACTION, OPT_TAIL_STREAMS, OPT_TAIL_PATHS_PER_LANE, OPT_SIMPLE_EXTEND, OPT_BULK_STREAMS, OPT_TAIL_MODE, OPT_FIRST_CHECK_ROUND, OPT_RENDER_MODE, OPT_MERGE_QUEUED_CALLS, OPT_DEVICE_BUILD, OPT_REDO_CAPACITY, OPT_SKIP_REPEATED_MISS, OPT_WAVES_PER_SIMD = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15

EXPORTS = [
    "pt_strerror", "pt_last_error", "pt_create", "pt_destroy", "pt_set_triangles", "pt_set_spheres", "pt_set_materials",
    "pt_set_sky", "pt_build", "pt_get_build_info", "pt_set_bvh", "pt_bvh_build", "pt_get_bvh_sizes", "pt_get_bvh", "pt_set_camera", "pt_get_camera", "pt_generate", "pt_render", "pt_render_async", "pt_wait",
    "pt_shard_map", "pt_shard_slab_floats", "pt_shard_pack", "pt_shard_unpack", "pt_reset_accum", "pt_read_accum", "pt_write_accum", "pt_read_rgb8", "pt_read_rgb8_latest", "pt_accum_device_ptr", "pt_add_external_samples", "pt_trace", "pt_trace_noalloc",
    "pt_sample_skybox", "pt_set_option", "pt_get_counters", "pt_get_round_stats", "pt_reset_counters",
]


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


class Shard(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("rank", C.c_uint32), ("nranks", C.c_uint32), ("tile", C.c_uint32),
                ("sample_begin", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_any", C.c_uint64),
                ("node_tests", C.c_uint64), ("prim_tests", C.c_uint64), ("extend_launches", C.c_uint64),
                ("extend_ms", C.c_double), ("render_ms", C.c_double), ("tail_rays", C.c_uint64),
                ("tail_node_tests", C.c_uint64), ("tail_prim_tests", C.c_uint64), ("tail_launches", C.c_uint64),
                ("tail_ms", C.c_double), ("extend_ms_sum", C.c_double), ("tail_ms_sum", C.c_double),
                ("redo_paths", C.c_uint64), ("redo_restarts", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class PtError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"ptcore status {status}: {text}")
        self.status = status


_lib = None
_torch_first = True  # False when libptcore.so was loaded into a process that had not imported torch yet (see dist.py)


def load() -> C.CDLL:
    """Load libptcore.so; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C pathtracer_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        global _torch_first
        _torch_first = "torch" in sys.modules
        lib = C.CDLL(os.environ.get("PTCORE_LIB", LIB_PATH))  # PTCORE_LIB: A/B builds of the same ABI
        lib.pt_strerror.restype = C.c_char_p
        lib.pt_last_error.restype = C.c_char_p
        lib.pt_last_error.argtypes = [C.c_void_p]
        lib.pt_destroy.restype = None
        lib.pt_destroy.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def as_rays(a) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype == RAY_DTYPE:
        return np.ascontiguousarray(a)
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 7)
    return a.view(RAY_DTYPE).reshape(-1)


def bvh_build_host(geom, prim_kind=0):
    """BVH::new on the host, no device needed (pt_bvh_build).  Returns (nodes, indices, max_depth)."""
    lib = load()
    w = 9 if prim_kind == 0 else 4
    geom = np.ascontiguousarray(geom, np.float32).reshape(-1, w)
    n = len(geom)
    nodes = np.zeros(max(2 * n - 1, 1), dtype=NODE_DTYPE)
    idx = np.zeros(n, dtype=np.uint64)
    nn, depth = C.c_uint64(), C.c_uint32()
    rc = lib.pt_bvh_build(C.c_int(prim_kind), _ptr(geom), C.c_uint64(n), _ptr(nodes), C.c_uint64(len(nodes)), _ptr(idx),
                          C.byref(nn), C.byref(depth))
    if rc != PT_OK:
        raise PtError(rc, lib.pt_strerror(rc).decode())
    return nodes[: nn.value].copy(), idx, depth.value


def shard_map(width, height, tile=8, rank=0, nranks=1):
    """(n, 2) uint32 pixel coordinates owned by a shard, in device order (pt_shard_map, no device)."""
    lib = load()
    n = C.c_uint64()
    rc = lib.pt_shard_map(C.c_uint32(width), C.c_uint32(height), C.c_uint32(tile), C.c_uint32(rank), C.c_uint32(nranks),
                          None, C.c_uint64(0), C.byref(n))
    if rc != PT_OK:
        raise PtError(rc, lib.pt_strerror(rc).decode())
    xy = np.zeros((n.value, 2), dtype=np.uint32)
    lib.pt_shard_map(C.c_uint32(width), C.c_uint32(height), C.c_uint32(tile), C.c_uint32(rank), C.c_uint32(nranks),
                     _ptr(xy), C.c_uint64(n.value), C.byref(n))
    return xy


def shard_slab_floats(width, height, tile, rank, nranks):
    """floats in the compact slab of a tile shard (pt_shard_slab_floats, no device)."""
    lib = load()
    n = C.c_uint64()
    rc = lib.pt_shard_slab_floats(C.c_uint32(width), C.c_uint32(height), C.c_uint32(tile), C.c_uint32(rank), C.c_uint32(nranks), C.byref(n))
    if rc != PT_OK:
        raise PtError(rc, lib.pt_strerror(rc).decode())
    return n.value


class PathTracer:
    """One context on one GPU (a `pt_ctx`)."""

    def __init__(self, device: int = 0):
        self.l = load()
        h = C.c_void_p()
        rc = self.l.pt_create(C.c_int(device), C.byref(h))
        if rc != PT_OK:
            raise PtError(rc, self.l.pt_strerror(rc).decode())
        self.h = h
        self.width = self.height = 0

    def _ck(self, rc):
        if rc != PT_OK:
            raise PtError(rc, self.l.pt_last_error(self.h).decode() or self.l.pt_strerror(rc).decode())

    def close(self):
        if getattr(self, "h", None):
            self.l.pt_destroy(self.h)
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
        self._ck(self.l.pt_set_materials(self.h, _ptr(mats), C.c_uint32(len(mats))))

    def set_triangles(self, pos, nrm, mat_id):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 9)
        nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1, 9)
        mat_id = np.ascontiguousarray(mat_id, np.uint32)
        self._ck(self.l.pt_set_triangles(self.h, _ptr(pos), _ptr(nrm), _ptr(mat_id), C.c_uint64(len(pos))))

    def set_spheres(self, cr, mat_id):
        cr = np.ascontiguousarray(cr, np.float32).reshape(-1, 4)
        mat_id = np.ascontiguousarray(mat_id, np.uint32)
        self._ck(self.l.pt_set_spheres(self.h, _ptr(cr), _ptr(mat_id), C.c_uint64(len(cr))))

    def set_sky(self, sky):
        sky = np.ascontiguousarray(sky, np.float32)
        h, w = sky.shape[:2]
        self._ck(self.l.pt_set_sky(self.h, _ptr(sky), C.c_uint32(w), C.c_uint32(h)))

    def set_scene(self, scene, sky=None):
        """scene: (cr, mat_id, mats) for spheres or (pos, nrm, mat_id, mats) for triangles."""
        if len(scene) == 3:
            cr, mat_id, mats = scene
            self.set_materials(mats)
            self.set_spheres(cr, mat_id)
        else:
            pos, nrm, mat_id, mats = scene
            self.set_materials(mats)
            self.set_triangles(pos, nrm, mat_id)
        if sky is not None:
            self.set_sky(sky)
        self.build()
        return self

    def build(self):
        self._ck(self.l.pt_build(self.h))

    def set_bvh(self, nodes, indices):
        nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
        indices = np.ascontiguousarray(indices, dtype=np.uint64)
        self._ck(self.l.pt_set_bvh(self.h, _ptr(nodes), C.c_uint64(len(nodes)), _ptr(indices), C.c_uint64(len(indices))))

    def build_info(self):
        """(tree built on the GPU?, milliseconds of the tree build alone) of the last pt_build."""
        on_dev, ms = C.c_uint32(), C.c_double()
        self._ck(self.l.pt_get_build_info(self.h, C.byref(on_dev), C.byref(ms)))
        return bool(on_dev.value), ms.value

    def bvh(self):
        n_nodes, n_idx, depth = C.c_uint64(), C.c_uint64(), C.c_uint32()
        self._ck(self.l.pt_get_bvh_sizes(self.h, C.byref(n_nodes), C.byref(n_idx), C.byref(depth)))
        nodes = np.zeros(n_nodes.value, dtype=NODE_DTYPE)
        idx = np.zeros(n_idx.value, dtype=np.uint64)
        self._ck(self.l.pt_get_bvh(self.h, _ptr(nodes), _ptr(idx)))
        return nodes, idx, depth.value

    # ---- camera
    def set_camera(self, origin, target, width, height, lens_size=0.04):
        o = (C.c_float * 3)(*origin)
        t = (C.c_float * 3)(*target)
        self._ck(self.l.pt_set_camera(self.h, o, t, C.c_uint32(width), C.c_uint32(height), C.c_float(lens_size)))
        self.width, self.height = width, height

    def camera(self):
        s = CameraState()
        self._ck(self.l.pt_get_camera(self.h, C.byref(s)))
        return s.as_dict()

    def generate(self, xy, sample, seed):
        xy = np.ascontiguousarray(xy, np.uint32).reshape(-1, 2)
        rays = np.zeros(len(xy), dtype=RAY_DTYPE)
        self._ck(self.l.pt_generate(self.h, _ptr(xy), C.c_uint64(len(xy)), C.c_uint32(sample), C.c_uint64(seed), _ptr(rays)))
        return rays

    # ---- frame loop
    def render(self, spp, max_depth=32, seed=0, shard=None, flags=0):
        sp = C.byref(shard) if shard is not None else None
        self._ck(self.l.pt_render(self.h, C.c_uint32(spp), C.c_uint32(max_depth), C.c_uint64(seed), sp, C.c_uint32(flags)))

    def render_async(self, spp, max_depth=32, seed=0, shard=None, flags=0):
        """Queue a render call and return; calls overlap on the device, sums stay in call order (pt_render_async)."""
        sp = C.byref(shard) if shard is not None else None
        self._ck(self.l.pt_render_async(self.h, C.c_uint32(spp), C.c_uint32(max_depth), C.c_uint64(seed), sp, C.c_uint32(flags)))

    def wait(self):
        self._ck(self.l.pt_wait(self.h))

    def reset_accum(self):
        self._ck(self.l.pt_reset_accum(self.h))

    def read_accum(self):
        out = np.zeros((self.height, self.width, 3), dtype=np.float32)
        spp = C.c_float()
        self._ck(self.l.pt_read_accum(self.h, _ptr(out), C.byref(spp)))
        return out, spp.value

    def write_accum(self, accum, spp):
        """Put back a sum read earlier with read_accum (resume); rendering continues at sample int(spp)."""
        a = np.ascontiguousarray(accum, np.float32)
        assert a.shape == (self.height, self.width, 3)
        self._ck(self.l.pt_write_accum(self.h, _ptr(a), C.c_float(spp)))

    def read_rgb8(self, pitch=None):
        pitch = pitch or self.width * 3
        out = np.zeros((self.height, pitch), dtype=np.uint8)
        self._ck(self.l.pt_read_rgb8(self.h, _ptr(out), C.c_uint64(pitch)))
        return out[:, : self.width * 3].reshape(self.height, self.width, 3)

    def read_rgb8_latest(self, pitch=None, out=None):
        """(image, samples added so far) without waiting for queued render calls (pt_read_rgb8_latest).
        `out`: a (height, pitch) uint8 array to fill instead of a new one (a frame loop reuses its texture buffer)."""
        pitch = pitch or self.width * 3
        if out is None:
            out = np.zeros((self.height, pitch), dtype=np.uint8)
        elif out.dtype != np.uint8 or out.shape != (self.height, pitch) or not out.flags.c_contiguous:
            raise ValueError("read_rgb8_latest: out must be a C-contiguous (height, pitch) uint8 array")
        spp = C.c_float()
        self._ck(self.l.pt_read_rgb8_latest(self.h, _ptr(out), C.c_uint64(pitch), C.byref(spp)))
        return out[:, : self.width * 3].reshape(self.height, self.width, 3), spp.value

    def accum_device_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._ck(self.l.pt_accum_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def shard_pack(self, shard, slab_ptr, cap_floats):
        """accumulator pixels of a tile shard -> compact slab in DEVICE memory (pt_shard_pack)."""
        self._ck(self.l.pt_shard_pack(self.h, C.byref(shard), C.c_void_p(slab_ptr), C.c_uint64(cap_floats)))

    def shard_unpack(self, shard, slab_ptr, n_floats):
        """compact slab in DEVICE memory -> the shard's pixels of the accumulator (pt_shard_unpack)."""
        self._ck(self.l.pt_shard_unpack(self.h, C.byref(shard), C.c_void_p(slab_ptr), C.c_uint64(n_floats)))

    def add_external_samples(self, spp):
        self._ck(self.l.pt_add_external_samples(self.h, C.c_float(spp)))

    # ---- seams
    def trace(self, rays, any_hit=False):
        rays = as_rays(rays)
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        self._ck(self.l.pt_trace(self.h, _ptr(rays), C.c_uint64(len(rays)), C.c_int(1 if any_hit else 0), _ptr(hits)))
        return hits

    def trace_noalloc(self, rays, any_hit=False):
        """pt_trace_noalloc: same results as trace(), no device allocation (<= 16384 rays per call)."""
        rays = as_rays(rays)
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        self._ck(self.l.pt_trace_noalloc(self.h, _ptr(rays), C.c_uint64(len(rays)), C.c_int(1 if any_hit else 0), _ptr(hits)))
        return hits

    def sample_skybox(self, dirs):
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros_like(dirs)
        self._ck(self.l.pt_sample_skybox(self.h, _ptr(dirs), C.c_uint64(len(dirs)), _ptr(out)))
        return out

    def set_option(self, option, value):
        self._ck(self.l.pt_set_option(self.h, C.c_int(option), C.c_uint64(value)))

    def counters(self):
        c = Counters()
        self._ck(self.l.pt_get_counters(self.h, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        self._ck(self.l.pt_reset_counters(self.h))

    def round_stats(self):
        sizes = np.zeros(4096, np.uint32)
        n, worst = C.c_uint32(), C.c_uint64()
        self._ck(self.l.pt_get_round_stats(self.h, _ptr(sizes), C.c_uint32(len(sizes)), C.byref(n), C.byref(worst)))
        return sizes[: n.value].copy(), worst.value


class AccumView:
    """Exposes the device accumulator through __cuda_array_interface__ so torch can wrap it
    (zero copy) for the RCCL reduce / gather that combines ranks."""

    def __init__(self, pt: PathTracer):
        ptr, n = pt.accum_device_ptr()
        self.__cuda_array_interface__ = {
            "shape": (pt.height, pt.width, 3), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}
        self._keepalive = pt
