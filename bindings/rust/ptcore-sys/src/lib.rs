//! Raw binding of `include/ptcore.h` — one `extern "C"` declaration per entry point, `#[repr(C)]` mirrors of its
//! structs.  The reference crate (ReinierMaas/PathTracer, edition-2015 nightly Rust) would depend on this crate and
//! replace the body of its frame loop (`src/main.rs:143-170`) with `pt_render` / `pt_read_rgb8`; INTEGRATION.md shows
//! the patch.  NOT COMPILED in the image this repository is built in (no rustc there): `tests/test_bindings_cpu.py`
//! keeps the list of functions and their argument counts in step with the header.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct pt_ctx {
    _private: [u8; 0],
}

pub const PT_OK: c_int = 0;
pub const PT_ERR_INVALID: c_int = -1;
pub const PT_ERR_STATE: c_int = -2;
pub const PT_ERR_NO_DEVICE: c_int = -3;
pub const PT_ERR_HIP: c_int = -4;
pub const PT_ERR_LIMIT: c_int = -5;
pub const PT_ERR_BVH: c_int = -6;

pub const PT_MAT_DIFFUSE: u32 = 0;
pub const PT_MAT_DIELECTRIC: u32 = 1;
pub const PT_MAT_EMISSIVE: u32 = 2;
pub const PT_MISS: u32 = 0xFFFF_FFFF;
pub const PT_SHARD_ALL: u32 = 0;
pub const PT_SHARD_TILES: u32 = 1;
pub const PT_RENDER_DEFAULT: u32 = 0;
pub const PT_RENDER_TIMED: u32 = 1;
pub const PT_RENDER_COUNT: u32 = 2;

/// `enum Material`, src/material.rs:5-18
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct pt_material {
    pub kind: u32,
    pub a: f32,
    pub b: f32,
    pub rgb: [f32; 3],
}

/// `Ray`, src/ray.rs:14-18
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct pt_ray {
    pub o: [f32; 3],
    pub d: [f32; 3],
    pub tmax: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct pt_hit {
    pub prim: u32,
    pub t: f32,
    pub inside: u32,
}

/// `BVHNode`, src/bvh.rs:10-14 (make it `#[repr(C)]` + `pub` in the reference to hand `bvh_nodes` over as is)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct pt_bvh_node {
    pub min: [f32; 3],
    pub left_first: u32,
    pub max: [f32; 3],
    pub count: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pt_camera_state {
    pub origin: [f32; 3],
    pub target: [f32; 3],
    pub direction: [f32; 3],
    pub right: [f32; 3],
    pub up: [f32; 3],
    pub p1: [f32; 3],
    pub p2: [f32; 3],
    pub p3: [f32; 3],
    pub focal_distance: f32,
    pub lens_size: f32,
    pub width: u32,
    pub height: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct pt_shard {
    pub kind: u32,
    pub rank: u32,
    pub nranks: u32,
    pub tile: u32,
    pub sample_begin: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pt_counters {
    pub paths: u64,
    pub rays_closest: u64,
    pub rays_any: u64,
    pub node_tests: u64,
    pub prim_tests: u64,
    pub extend_launches: u64,
    pub extend_ms: f64,
    pub render_ms: f64,
    pub tail_rays: u64,
    pub tail_node_tests: u64,
    pub tail_prim_tests: u64,
    pub tail_launches: u64,
    pub tail_ms: f64,
    pub extend_ms_sum: f64,
    pub tail_ms_sum: f64,
    pub redo_paths: u64,
    pub redo_restarts: u64,
}

extern "C" {
    pub fn pt_strerror(status: c_int) -> *const c_char;
    pub fn pt_last_error(ctx: *const pt_ctx) -> *const c_char;
    pub fn pt_create(device: c_int, out: *mut *mut pt_ctx) -> c_int;
    pub fn pt_destroy(ctx: *mut pt_ctx);
    pub fn pt_set_triangles(ctx: *mut pt_ctx, pos: *const f32, nrm: *const f32, mat_id: *const u32, n: u64) -> c_int;
    pub fn pt_set_spheres(ctx: *mut pt_ctx, cr: *const f32, mat_id: *const u32, n: u64) -> c_int;
    pub fn pt_set_materials(ctx: *mut pt_ctx, m: *const pt_material, count: u32) -> c_int;
    pub fn pt_set_sky(ctx: *mut pt_ctx, rgb: *const f32, w: u32, h: u32) -> c_int;
    pub fn pt_build(ctx: *mut pt_ctx) -> c_int;
    pub fn pt_set_bvh(ctx: *mut pt_ctx, nodes: *const pt_bvh_node, n_nodes: u64, indices: *const u64, n_indices: u64) -> c_int;
    pub fn pt_bvh_build(prim_kind: c_int, geom: *const f32, n: u64, nodes: *mut pt_bvh_node, nodes_cap: u64, indices: *mut u64, n_nodes_out: *mut u64, max_depth_out: *mut u32) -> c_int;
    pub fn pt_get_build_info(ctx: *const pt_ctx, on_device: *mut u32, build_ms: *mut f64) -> c_int;
    pub fn pt_get_bvh_sizes(ctx: *const pt_ctx, n_nodes: *mut u64, n_indices: *mut u64, max_depth: *mut u32) -> c_int;
    pub fn pt_get_bvh(ctx: *const pt_ctx, nodes: *mut pt_bvh_node, indices: *mut u64) -> c_int;
    pub fn pt_set_camera(ctx: *mut pt_ctx, origin: *const f32, target: *const f32, width: u32, height: u32, lens_size: f32) -> c_int;
    pub fn pt_get_camera(ctx: *const pt_ctx, out: *mut pt_camera_state) -> c_int;
    pub fn pt_generate(ctx: *mut pt_ctx, xy: *const u32, n: u64, sample: u32, seed: u64, rays: *mut pt_ray) -> c_int;
    pub fn pt_render(ctx: *mut pt_ctx, spp: u32, max_depth: u32, seed: u64, shard: *const pt_shard, flags: u32) -> c_int;
    pub fn pt_render_async(ctx: *mut pt_ctx, spp: u32, max_depth: u32, seed: u64, shard: *const pt_shard, flags: u32) -> c_int;
    pub fn pt_wait(ctx: *mut pt_ctx) -> c_int;
    pub fn pt_shard_map(width: u32, height: u32, tile: u32, rank: u32, nranks: u32, xy: *mut u32, cap: u64, n_out: *mut u64) -> c_int;
    pub fn pt_reset_accum(ctx: *mut pt_ctx) -> c_int;
    pub fn pt_read_accum(ctx: *mut pt_ctx, rgb: *mut f32, spp_out: *mut f32) -> c_int;
    pub fn pt_write_accum(ctx: *mut pt_ctx, rgb: *const f32, spp: f32) -> c_int;
    pub fn pt_read_rgb8(ctx: *mut pt_ctx, rgb: *mut u8, pitch_bytes: u64) -> c_int;
    pub fn pt_read_rgb8_latest(ctx: *mut pt_ctx, rgb: *mut u8, pitch_bytes: u64, spp_out: *mut f32) -> c_int;
    pub fn pt_accum_device_ptr(ctx: *mut pt_ctx, ptr: *mut *mut c_void, n_floats: *mut u64) -> c_int;
    pub fn pt_add_external_samples(ctx: *mut pt_ctx, spp: f32) -> c_int;
    pub fn pt_shard_slab_floats(width: u32, height: u32, tile: u32, rank: u32, nranks: u32, n_floats: *mut u64) -> c_int;
    pub fn pt_shard_pack(ctx: *mut pt_ctx, shard: *const pt_shard, slab_dev: *mut f32, cap_floats: u64) -> c_int;
    pub fn pt_shard_unpack(ctx: *mut pt_ctx, shard: *const pt_shard, slab_dev: *const f32, n_floats: u64) -> c_int;
    pub fn pt_trace(ctx: *mut pt_ctx, rays: *const pt_ray, n: u64, any_hit: c_int, hits: *mut pt_hit) -> c_int;
    pub fn pt_trace_noalloc(ctx: *mut pt_ctx, rays: *const pt_ray, n: u64, any_hit: c_int, hits: *mut pt_hit) -> c_int;
    pub fn pt_sample_skybox(ctx: *mut pt_ctx, dirs: *const f32, n: u64, rgb: *mut f32) -> c_int;
    pub fn pt_set_option(ctx: *mut pt_ctx, option: c_int, value: u64) -> c_int;
    pub fn pt_get_counters(ctx: *mut pt_ctx, out: *mut pt_counters) -> c_int;
    pub fn pt_get_round_stats(ctx: *mut pt_ctx, queue_sizes: *mut u32, cap: u32, n_rounds: *mut u32, max_node_tests: *mut u64) -> c_int;
    pub fn pt_reset_counters(ctx: *mut pt_ctx) -> c_int;
}
