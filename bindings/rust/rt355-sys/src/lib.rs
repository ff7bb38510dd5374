//! Raw declarations of `include/rt355.h`.  Layouts are checked at start-up
//! against `rt_abi_sizes` by `check_layout()`.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_NO_DEVICE: c_int = -2;
pub const RT_ERR_HIP: c_int = -3;
pub const RT_ERR_DEPTH: c_int = -4;
pub const RT_ERR_NOMEM: c_int = -5;
pub const RT_ERR_LIMIT: c_int = -6;
pub const RT_MISS: u32 = 0xFFFF_FFFF;
pub const RT_PRIM_REVERSE: u32 = 0x100;

pub const RT_SPHERE: u32 = 0;
pub const RT_MOVING_SPHERE: u32 = 1;
pub const RT_TRIANGLE: u32 = 2;
pub const RT_RECT: u32 = 3;
pub const RT_BOX: u32 = 4;
pub const RT_GSPHERE: u32 = 5;
pub const RT_MEDIUM: u32 = 6;

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_ray { pub o: [f32; 3], pub d: [f32; 3], pub t_max: f32, pub time: f32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_hit { pub t: f32, pub prim: u32, pub u: f32, pub v: f32, pub n: [f32; 3], pub flags: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_prim_ref { pub kind: u32, pub index: u32, pub material: u32, pub xform: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_sphere { pub c: [f32; 3], pub r: f32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_moving_sphere { pub c0: [f32; 3], pub r: f32, pub c1: [f32; 3], pub time0: f32, pub time1: f32, pub _pad: [f32; 3] }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_rect { pub axis: u32, pub k: f32, pub a0: f32, pub a1: f32, pub b0: f32, pub b1: f32, pub _pad: [u32; 2] }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_box { pub lo: [f32; 3], pub _p0: f32, pub hi: [f32; 3], pub _p1: f32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_gsphere { pub o2w: [f32; 12], pub w2o: [f32; 12], pub radius: f32, pub z_min: f32, pub z_max: f32, pub phi_max_deg: f32, pub _pad: [u32; 4] }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_xform { pub o2w: [f32; 12], pub w2o: [f32; 12] }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_medium { pub boundary_kind: u32, pub boundary_index: u32, pub xform: u32, pub density: f32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_material { pub kind: u32, pub texture: u32, pub fuzz: f32, pub ior: f32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_texture { pub kind: u32, pub rgb: [f32; 3], pub a: u32, pub b: u32, pub scale: f32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_distant_light { pub w: [f32; 3], pub rgb: [f32; 3], pub _pad: [u32; 2] }
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct rt_image { pub rgb8: *const u8, pub w: u32, pub h: u32 }

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct rt_scene_desc {
    pub prims: *const rt_prim_ref,
    pub spheres: *const rt_sphere,
    pub moving_spheres: *const rt_moving_sphere,
    pub tri_verts: *const f32,
    pub tri_uvs: *const f32,
    pub tri_normals: *const f32,
    pub rects: *const rt_rect,
    pub boxes: *const rt_box,
    pub gspheres: *const rt_gsphere,
    pub xforms: *const rt_xform,
    pub media: *const rt_medium,
    pub materials: *const rt_material,
    pub textures: *const rt_texture,
    pub images: *const rt_image,
    pub perlin_vec: *const f32,
    pub perlin_perm: *const i32,
    pub n_prims: u32, pub n_spheres: u32, pub n_moving_spheres: u32, pub n_tris: u32,
    pub n_rects: u32, pub n_boxes: u32, pub n_gspheres: u32, pub n_xforms: u32,
    pub n_media: u32, pub n_materials: u32, pub n_textures: u32, pub n_images: u32,
    pub background: [f32; 3],
    pub n_distant_lights: u32,
    pub distant_lights: *const rt_distant_light,
}

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_build_opts { pub max_prims_in_node: u32, pub split_method: u32, pub flags: u32, pub device: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_linear_node { pub lo: [f32; 3], pub hi: [f32; 3], pub child_offset: u32, pub primitive_count: u16, pub axis: u8, pub _pad: u8 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_scene_info { pub n_prims: u32, pub n_nodes: u32, pub max_depth: u32, pub stack_slots: u32, pub device_bytes: u64, pub build_ms: f64, pub n_enclosing: u32, pub enclosing: [u32; 4], pub n_balls: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_camera { pub raster_to_camera: [f32; 16], pub camera_to_world: [f32; 12], pub lens_radius: f32, pub focal_distance: f32, pub shutter_open: f32, pub shutter_close: f32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_filter { pub kind: u32, pub radius: [f32; 2], pub a: f32, pub b: f32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct rt_render_opts { pub width: u32, pub height: u32, pub spp: u32, pub max_depth: u32, pub seed: u64, pub shard_index: u32, pub shard_count: u32, pub max_paths: u32, pub flags: u32, pub filter: rt_filter, pub sampler: u32, pub sampler_arg: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct rt_stats {
    pub paths: u64, pub segments: u64, pub nodes_visited: u64, pub prims_tested: [u64; 8],
    pub trace_launches: u64, pub trace_ms: f64, pub total_ms: f64, pub batches: u32, pub max_bounce: u32,
    pub bounce_rays: [u64; 64], pub occlusion_rays: u64,
}

#[repr(C)] pub struct rt_scene { _opaque: [u8; 0] }

extern "C" {
    pub fn rt_scene_create(desc: *const rt_scene_desc, opts: *const rt_build_opts, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene);
    pub fn rt_scene_world_bound(scene: *const rt_scene, out: *mut f32) -> c_int;
    pub fn rt_scene_get_info(scene: *const rt_scene, out: *mut rt_scene_info) -> c_int;
    pub fn rt_scene_get_nodes(scene: *const rt_scene, out: *mut rt_linear_node, cap: usize) -> i64;
    pub fn rt_scene_get_prim_order(scene: *const rt_scene, out: *mut u32, cap: usize) -> i64;
    pub fn rt_scene_upload(scene: *mut rt_scene, device: c_int) -> c_int;
    pub fn rt_intersect(scene: *mut rt_scene, rays: *const rt_ray, n: usize, hits: *mut rt_hit, device: c_int) -> c_int;
    pub fn rt_intersect_dev(scene: *mut rt_scene, d_rays: *const rt_ray, n: usize, d_hits: *mut rt_hit, device: c_int, stream: *mut c_void) -> c_int;
    pub fn rt_intersect_dev_noalloc(scene: *mut rt_scene, d_rays: *const rt_ray, n: usize, d_hits: *mut rt_hit, device: c_int, stream: *mut c_void) -> c_int;
    pub fn rt_intersect_count(scene: *mut rt_scene, d_rays: *const rt_ray, n: usize, d_hits: *mut rt_hit, device: c_int, out: *mut u64) -> c_int;
    pub fn rt_occluded(scene: *mut rt_scene, rays: *const rt_ray, n: usize, out: *mut u8, device: c_int) -> c_int;
    pub fn rt_occluded_dev(scene: *mut rt_scene, d_rays: *const rt_ray, n: usize, d_out: *mut u8, device: c_int, stream: *mut c_void) -> c_int;
    pub fn rt_generate_rays(cam: *const rt_camera, opts: *const rt_render_opts, first_path: u64, n: usize, rays: *mut rt_ray, device: c_int) -> c_int;
    pub fn rt_render(scene: *mut rt_scene, cam: *const rt_camera, opts: *const rt_render_opts, rgba_out: *mut f32, stats: *mut rt_stats, device: c_int) -> c_int;
    pub fn rt_render_dev(scene: *mut rt_scene, cam: *const rt_camera, opts: *const rt_render_opts, d_rgba_out: *mut f32, stats: *mut rt_stats, device: c_int, stream: *mut c_void) -> c_int;
    pub fn rt_filter_table(filter: *const rt_filter, out: *mut f32) -> c_int;
    pub fn rt_sobol_tables(width: u32, height: u32, m32: *mut u32, vdc: *mut u64, inv: *mut u64) -> c_int;
    pub fn rt_film_resolve_dev(d_rgba: *mut f32, width: u32, height: u32, device: c_int, stream: *mut c_void) -> c_int;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_abi_sizes(out: *mut u32, cap: usize) -> c_int;
}

/// Compares `size_of` of every struct with what the loaded library was built with.
pub fn check_layout() -> Result<(), String> {
    use std::mem::size_of;
    let mine = [
        size_of::<rt_ray>(), size_of::<rt_hit>(), size_of::<rt_prim_ref>(), size_of::<rt_sphere>(),
        size_of::<rt_moving_sphere>(), size_of::<rt_rect>(), size_of::<rt_box>(), size_of::<rt_gsphere>(),
        size_of::<rt_xform>(), size_of::<rt_medium>(), size_of::<rt_material>(), size_of::<rt_texture>(),
        size_of::<rt_image>(), size_of::<rt_scene_desc>(), size_of::<rt_build_opts>(),
        size_of::<rt_linear_node>(), size_of::<rt_scene_info>(), size_of::<rt_camera>(),
        size_of::<rt_filter>(), size_of::<rt_render_opts>(), size_of::<rt_stats>(),
        size_of::<rt_distant_light>(),
    ];
    let mut theirs = [0u32; 64];
    let n = unsafe { rt_abi_sizes(theirs.as_mut_ptr(), theirs.len()) } as usize;
    if n != mine.len() { return Err(format!("library reports {n} structs, bindings know {}", mine.len())); }
    for (i, (a, b)) in mine.iter().zip(&theirs[..n]).enumerate() {
        if *a != *b as usize { return Err(format!("struct #{i}: {a} bytes here, {b} in the library")); }
    }
    Ok(())
}
