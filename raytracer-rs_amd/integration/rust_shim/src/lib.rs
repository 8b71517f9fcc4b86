//! UNCOMPILED SOURCE — written against include/mi355rt.h; there is no rustc in this repository's
//! build environment, so this file has never been compiled or run.
//!
//! Drop-in for the reference crate `raytracer_lib`: the modules `scene` (data model + COLLADA loader)
//! and `vecmath` stay the reference's own Rust code (copy them from the reference crate unchanged:
//! scene/, vecmath.rs — they are host-side and cold); `raytracer/` (render core, octree, film) is
//! replaced by the FFI below.  Public surface kept (lib.rs:5-27 of the reference):
//!   create_raytracer, create_raytracer_from_file, RayTracer{camera, film, trace_frame_additive,
//!   get_tonemapped_pixels}, stats::Stats, DEFAULT_TRIANGLES_PER_LEAF.
#![allow(non_camel_case_types)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_int};

mod scene;      // the reference's scene/ directory, unchanged
mod vecmath;    // the reference's vecmath.rs, unchanged
pub mod stats;  // the reference's stats.rs, unchanged

use scene::loaders::{colladaloader::ColladaLoader, SceneLoader};
use scene::{color::Diffuse, Scene};

pub const DEFAULT_TRIANGLES_PER_LEAF: usize = 70;

// ---- C ABI (include/mi355rt.h) -------------------------------------------------------------------
#[repr(C)] pub struct mi355rt_handle { _private: [u8; 0] }
#[repr(C)] struct mi355rt_material { kind: u32, rgb: [f32; 3], tex_id: u32 }
#[repr(C)] struct mi355rt_light { pos: [f32; 3], color: [f32; 3] }
#[repr(C)] struct mi355rt_texture { width: u32, height: u32, rgb: *const f32 }
#[repr(C)] struct mi355rt_scene_desc {
    tri_verts: *const f32, tri_geom: *const u32, ntri: u32,
    materials: *const mi355rt_material, nmaterials: u32,
    lights: *const mi355rt_light, nlights: u32,
    textures: *const mi355rt_texture, ntextures: u32,
    camera_orientation: [f32; 16], camera_fov_deg: f32,
}
#[repr(C)] struct mi355rt_config {
    width: u32, height: u32, triangles_per_leaf: u32, recursions: u32, spread: u32, flags: u32,
    seed: u64, device: i32, stripe_rows: u32, stripe_rank: u32, stripe_world: u32, samples_per_pass: u32,
    device_count: u32,      // > 1: the handle drives that many GPUs of this process (rows dealt in stripes, gathered on the first)
}
extern "C" {
    fn mi355rt_default_config(cfg: *mut mi355rt_config);
    fn mi355rt_create(scene: *const mi355rt_scene_desc, cfg: *const mi355rt_config, out: *mut *mut mi355rt_handle) -> c_int;
    fn mi355rt_destroy(h: *mut mi355rt_handle);
    fn mi355rt_last_error(h: *const mi355rt_handle) -> *const c_char;
    fn mi355rt_trace_frame_additive(h: *mut mi355rt_handle) -> u32;
    fn mi355rt_get_tonemapped_pixels(h: *mut mi355rt_handle, out: *mut u32, n: usize) -> c_int;
    fn mi355rt_film_clear(h: *mut mi355rt_handle) -> c_int;
    fn mi355rt_camera_move_rel(h: *mut mi355rt_handle, x: f32, y: f32, z: f32) -> c_int;
    fn mi355rt_camera_add_x_angle(h: *mut mi355rt_handle, radians: f32) -> c_int;
    fn mi355rt_camera_add_y_angle(h: *mut mi355rt_handle, radians: f32) -> c_int;
}

fn last_error(h: *const mi355rt_handle) -> String {
    unsafe { CStr::from_ptr(mi355rt_last_error(h)).to_string_lossy().into_owned() }
}

/// `pub camera` of the reference's RayTracer (mod.rs:38): the three methods main.rs:125-161 calls.
pub struct Camera { h: *mut mi355rt_handle }
impl Camera {
    pub fn move_rel(&mut self, x: f32, y: f32, z: f32) { unsafe { mi355rt_camera_move_rel(self.h, x, y, z); } }
    pub fn add_x_angle(&mut self, radians: f32) { unsafe { mi355rt_camera_add_x_angle(self.h, radians); } }
    pub fn add_y_angle(&mut self, radians: f32) { unsafe { mi355rt_camera_add_y_angle(self.h, radians); } }
}
/// `pub film` of the reference's RayTracer (mod.rs:41): main.rs:126-162 calls clear().
pub struct Film { h: *mut mi355rt_handle }
impl Film {
    pub fn clear(&mut self) { unsafe { mi355rt_film_clear(self.h); } }
}

pub struct RayTracer {
    h: *mut mi355rt_handle,
    width: usize,
    height: usize,
    pub camera: Camera,
    pub film: Film,
}
// created on the main thread, moved into the tracer thread (main.rs:183,194-196); the library binds its
// HIP device on every call and is not re-entrant per handle, exactly the reference's usage.
unsafe impl Send for RayTracer {}

impl RayTracer {
    pub fn trace_frame_additive(&mut self) -> u32 {
        let n = unsafe { mi355rt_trace_frame_additive(self.h) };
        if n == 0 { panic!("{}", last_error(self.h)); }      // the reference's runtime failures are panics too
        n
    }
    pub fn get_tonemapped_pixels(&self) -> Vec<u32> {
        let mut out = vec![0u32; self.width * self.height];
        let rc = unsafe { mi355rt_get_tonemapped_pixels(self.h, out.as_mut_ptr(), out.len()) };
        if rc != 0 { panic!("{}", last_error(self.h)); }
        out
    }
}
impl Drop for RayTracer {
    fn drop(&mut self) { unsafe { mi355rt_destroy(self.h) } }
}

pub fn create_raytracer(collada_doc: &str, triangles_per_leaf: usize, width: usize, height: usize) -> Result<RayTracer, String> {
    let scene = ColladaLoader::from_str(collada_doc, None, width, height).map_err(|e| e.to_string())?;
    build_raytracer(scene, triangles_per_leaf, width, height)
}

pub fn create_raytracer_from_file(collada_filename: String, triangles_per_leaf: usize, width: usize, height: usize) -> Result<RayTracer, String> {
    let scene = ColladaLoader::from_file(collada_filename, width, height).map_err(|e| e.to_string())?;
    build_raytracer(scene, triangles_per_leaf, width, height)
}

/// lib.rs:29-44 of the reference, with the octree build + RayTracer::new_with_intersector replaced by
/// one mi355rt_create call.  Needs two small accessors added to the reference's scene code:
/// `Camera::orientation_and_fov() -> ([f32; 16], f32)` (the arguments from_orientation_matrix received)
/// and `Texture::{width, height, data}` getters.
fn build_raytracer(scene: Scene, triangles_per_leaf: usize, width: usize, height: usize) -> Result<RayTracer, String> {
    let mut tri_verts: Vec<f32> = Vec::new();
    let mut tri_geom: Vec<u32> = Vec::new();
    let mut materials: Vec<mi355rt_material> = Vec::new();
    for (gi, geom) in scene.geometries.iter().enumerate() {
        for v in &geom.transformed_vertices { tri_verts.extend_from_slice(&[v.x, v.y, v.z]); }
        tri_geom.extend(std::iter::repeat(gi as u32).take(geom.transformed_vertices.len() / 3));
        materials.push(match &geom.material.diffuse {
            Diffuse::Color(rgb) => mi355rt_material { kind: 0, rgb: [rgb.r, rgb.g, rgb.b], tex_id: 0 },
            Diffuse::TextureId(id) => mi355rt_material { kind: 1, rgb: [0.0; 3], tex_id: *id as u32 },
        });
    }
    let lights: Vec<mi355rt_light> = scene.lights.iter()
        .map(|l| mi355rt_light { pos: [l.pos.x, l.pos.y, l.pos.z], color: [l.color.r, l.color.g, l.color.b] }).collect();
    let texel_store: Vec<Vec<f32>> = scene.textures.iter()
        .map(|t| t.data().iter().flat_map(|c| [c.r, c.g, c.b]).collect()).collect();
    let textures: Vec<mi355rt_texture> = scene.textures.iter().zip(&texel_store)
        .map(|(t, px)| mi355rt_texture { width: t.width() as u32, height: t.height() as u32, rgb: px.as_ptr() }).collect();
    let (orientation, fov_deg) = scene.cameras.get(0).ok_or("scene has no camera")?.orientation_and_fov();

    let desc = mi355rt_scene_desc {
        tri_verts: tri_verts.as_ptr(), tri_geom: tri_geom.as_ptr(), ntri: tri_geom.len() as u32,
        materials: materials.as_ptr(), nmaterials: materials.len() as u32,
        lights: lights.as_ptr(), nlights: lights.len() as u32,
        textures: textures.as_ptr(), ntextures: textures.len() as u32,
        camera_orientation: orientation, camera_fov_deg: fov_deg,
    };
    let mut cfg: mi355rt_config = unsafe { std::mem::zeroed() };
    unsafe { mi355rt_default_config(&mut cfg) };
    cfg.width = width as u32; cfg.height = height as u32; cfg.triangles_per_leaf = triangles_per_leaf as u32;
    cfg.seed = std::time::SystemTime::now().duration_since(std::time::UNIX_EPOCH).map(|d| d.as_nanos() as u64).unwrap_or(1);
    // cfg.flags = 0: the library intersects with the reference's DEFAULT intersector semantics (OctTreeIntersector built
    // with triangles_per_leaf, as lib.rs:29-44 wires it), exactly.  cfg.device_count > 1 spreads the rows over that many
    // GPUs of this process; the RayTracer stays one object.
    if let Ok(n) = std::env::var("MI355RT_GPUS") { if let Ok(n) = n.parse::<u32>() { cfg.device_count = n.max(1); } }

    let mut h: *mut mi355rt_handle = std::ptr::null_mut();
    let rc = unsafe { mi355rt_create(&desc, &cfg, &mut h) };
    if rc != 0 { return Err(last_error(std::ptr::null())); }
    Ok(RayTracer { h, width, height, camera: Camera { h }, film: Film { h } })
}
