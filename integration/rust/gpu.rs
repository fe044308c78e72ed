//! src/gpu.rs — binding of include/rt.h (ABI version 1) and the replacement of the render closure
//! main.rs:152-174.  Field order and types mirror the C header exactly; `FlatScene` owns the arrays.
#![allow(dead_code)]
use crate::cam::Camera;
use crate::hittable::Hittable;
use crate::alias::Color;
use image::RgbImage;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const RT_ABI_VERSION: u32 = 3;
pub const RT_REF_LEAF: u32 = 0x8000_0000;
pub const RT_REF_NONE: u32 = 0xFFFF_FFFF;
pub const fn make_leaf(kind: u32, index: u32) -> u32 { RT_REF_LEAF | (kind << 28) | index }
pub const LEAF_SPHERE: u32 = 0;
pub const LEAF_MSPHERE: u32 = 1;
pub const LEAF_CUBE: u32 = 2;
pub const LEAF_COMPLEX: u32 = 3;
pub const OP_TRANSLATE: u8 = 0;
pub const OP_ROTATE: u8 = 1;
pub const OP_MEDIUM: u8 = 2;
pub const MAT_LAMBERTIAN: u8 = 0;
pub const MAT_DIELECTRIC: u8 = 1;
pub const MAT_METAL: u8 = 2;
pub const MAT_DIFFUSE_LIGHT: u8 = 3;
pub const MAT_ISOTROPIC: u8 = 4;
pub const TEX_SOLID: u8 = 0;
pub const TEX_CHECKER: u8 = 1;
pub const TEX_IMAGE: u8 = 2;
pub const TEX_NORMAL: u8 = 3;
pub const TEX_UV: u8 = 4;
pub const LIGHT_SPHERE: u8 = 0;
pub const LIGHT_OTHER: u8 = 1;
pub const LIGHTS_NONE: u32 = 0;
pub const LIGHTS_SINGLE: u32 = 1;
pub const LIGHTS_LIST: u32 = 2;

#[repr(C)]
pub struct RtSceneDesc {
    pub abi_version: u32,
    pub bvh_height: u32,
    pub n_nodes: u32,
    pub root: u32,
    pub node_box: *const f64,   // [n_nodes][6]
    pub node_left: *const u32,
    pub node_right: *const u32,
    pub n_spheres: u32,
    pub sphere: *const f64,     // [n][4]
    pub sphere_material: *const u32,
    pub sphere_object: *const u32,
    pub n_mspheres: u32,
    pub msphere: *const f64,    // [n][9]
    pub msphere_material: *const u32,
    pub msphere_object: *const u32,
    pub n_cubes: u32,
    pub cube: *const f64,       // [n][6]
    pub cube_material: *const u32,
    pub cube_object: *const u32,
    pub n_complex: u32,
    pub complex_body: *const u32,
    pub complex_op_first: *const u32,
    pub complex_op_count: *const u32,
    pub complex_material: *const u32,
    pub complex_object: *const u32,
    pub n_ops: u32,
    pub op_tag: *const u8,
    pub op_data: *const f64,    // [n_ops][4]
    pub n_materials: u32,
    pub mat_tag: *const u8,
    pub mat_texture: *const u32,
    pub mat_params: *const f64, // [n][4]
    pub n_textures: u32,
    pub tex_tag: *const u8,
    pub tex_color: *const f64,  // [n][3]
    pub tex_odd: *const u32,
    pub tex_even: *const u32,
    pub tex_texel_offset: *const u64,
    pub tex_width: *const u32,
    pub tex_height: *const u32,
    pub n_texel_bytes: u64,
    pub texels: *const u8,
    pub lights_kind: u32,
    pub n_lights: u32,
    pub light_tag: *const u8,
    pub light_sphere: *const f64, // [n_lights][4]
    pub light_box: *const f64,    // [n_lights][6], RT_LIGHT_CUBE (extension) only: null here
    pub background: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtCamera {
    pub origin: [f64; 3],
    pub lower_left: [f64; 3],
    pub horizontal: [f64; 3],
    pub vertical: [f64; 3],
    pub u: [f64; 3],
    pub v: [f64; 3],
    pub lens_radius: f64,
    pub time_open: f64,
    pub time_close: f64,
}

#[repr(C)]
pub struct RtParams {
    pub width: u32,
    pub height: u32,
    pub spp: u32,
    pub max_depth: u32,
    pub seed: u64,
    pub sample_offset: u64,
    pub spp_per_launch: u32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Default)]
pub struct RtStats {
    pub rays: u64,
    pub light_probes: u64,
    pub medium_probes: u64,
    pub nodes_visited: u64,
    pub sphere_tests: u64,
    pub sphere_hits: u64,
    pub cube_tests: u64,
    pub cube_hits: u64,
    pub rotate_visits: u64,
    pub translate_visits: u64,
    pub medium_visits: u64,
    pub closest_hits: u64,
    pub diffuse_vertices: u64,
    pub specular_vertices: u64,
    pub emitter_hits: u64,
    pub misses: u64,
    pub nonfinite_samples: u64,
    pub launches: u32,
    pub reserved: u32,
    pub kernel_ms: f64,
}

pub enum RtScene {}
pub enum RtMulti {}

extern "C" {
    pub fn rt_scene_create(device: c_int, desc: *const RtSceneDesc, out: *mut *mut RtScene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut RtScene);
    pub fn rt_render(scene: *mut RtScene, cam: *const RtCamera, p: *const RtParams, accum_rgb: *mut f32,
                     stats: *mut RtStats) -> c_int;
    pub fn rt_render_device(scene: *mut RtScene, cam: *const RtCamera, p: *const RtParams, accum_rgb_device: *mut f32,
                            accumulate: c_int, stream: *mut c_void, stats: *mut RtStats) -> c_int;
    pub fn rt_render_multi(n_dev: c_int, devices: *const c_int, desc: *const RtSceneDesc, cam: *const RtCamera,
                           p: *const RtParams, accum_rgb: *mut f32, per_dev: *mut RtStats) -> c_int;
    // the same with the per-device scene copies kept across renders (an animation, progressive passes)
    pub fn rt_multi_create(n_dev: c_int, devices: *const c_int, desc: *const RtSceneDesc, out: *mut *mut RtMulti) -> c_int;
    pub fn rt_multi_render(multi: *mut RtMulti, cam: *const RtCamera, p: *const RtParams, accum_rgb: *mut f32,
                           per_dev: *mut RtStats) -> c_int;
    // ... in the reference's own arithmetic (alias.rs:4-7): the f64 kernels on every shard, f64 sums
    pub fn rt_multi_render_f64(multi: *mut RtMulti, cam: *const RtCamera, p: *const RtParams, accum_rgb: *mut f64,
                               per_dev: *mut RtStats) -> c_int;
    pub fn rt_multi_destroy(multi: *mut RtMulti);
    pub fn rt_multi_shard(spp: u64, g: u32, n_dev: u32, first: *mut u64, count: *mut u64);
    pub fn rt_primary_hits(scene: *mut RtScene, cam: *const RtCamera, width: u32, height: u32, seed: u64, time: f64,
                           prim_id: *mut u32, t: *mut f32, normal: *mut f32, uv: *mut f32, front_face: *mut u8) -> c_int;
    pub fn rt_tonemap(accum_rgb: *const f32, n_pixels: u64, spp: u64, rgb8: *mut u8);
    // the same closure in the reference's own arithmetic (f64 throughout, alias.rs:4-7) and its tone map
    pub fn rt_render_f64(scene: *mut RtScene, cam: *const RtCamera, p: *const RtParams, accum_rgb: *mut f64,
                         stats: *mut RtStats) -> c_int;
    pub fn rt_render_f64_device(scene: *mut RtScene, cam: *const RtCamera, p: *const RtParams, accum_rgb_device: *mut f64,
                                accumulate: c_int, stream: *mut c_void, stats: *mut RtStats) -> c_int;
    pub fn rt_tonemap_f64(accum_rgb: *const f64, n_pixels: u64, spp: u64, rgb8: *mut u8);
    pub fn rt_device_count() -> c_int;
    pub fn rt_last_error() -> *const c_char;
}

/// Owner of every array a `RtSceneDesc` points into (the library copies them during `rt_scene_create`).
#[derive(Default)]
pub struct FlatScene {
    pub bvh_height: u32,
    pub root: u32,
    pub node_box: Vec<f64>,
    pub node_left: Vec<u32>,
    pub node_right: Vec<u32>,
    pub sphere: Vec<f64>,
    pub sphere_material: Vec<u32>,
    pub sphere_object: Vec<u32>,
    pub msphere: Vec<f64>,
    pub msphere_material: Vec<u32>,
    pub msphere_object: Vec<u32>,
    pub cube: Vec<f64>,
    pub cube_material: Vec<u32>,
    pub cube_object: Vec<u32>,
    pub complex_body: Vec<u32>,
    pub complex_op_first: Vec<u32>,
    pub complex_op_count: Vec<u32>,
    pub complex_material: Vec<u32>,
    pub complex_object: Vec<u32>,
    pub op_tag: Vec<u8>,
    pub op_data: Vec<f64>,
    pub mat_tag: Vec<u8>,
    pub mat_texture: Vec<u32>,
    pub mat_params: Vec<f64>,
    pub tex_tag: Vec<u8>,
    pub tex_color: Vec<f64>,
    pub tex_odd: Vec<u32>,
    pub tex_even: Vec<u32>,
    pub tex_texel_offset: Vec<u64>,
    pub tex_width: Vec<u32>,
    pub tex_height: Vec<u32>,
    pub texels: Vec<u8>,
    pub lights_kind: u32,
    pub light_tag: Vec<u8>,
    pub light_sphere: Vec<f64>,
    pub background: [f64; 3],
    pub next_object: u32,
    // instances (ABI version 3): see Hittable::flatten_world in flatten_hittable.rs
    pub instance_owner: Option<u32>,
    pub pending_instances: Vec<(u32, u32, *const Hittable)>, // (complex index, owner object, sub-tree)
}

impl FlatScene {
    /// main.rs:125-139 in flat form: `world` is the value `Hittable::new_bvh` returned, `lights` the
    /// `Option<Hittable>` after the fold at main.rs:126-131.
    pub fn new(world: &Hittable, lights: &Option<Hittable>, background: Color) -> Result<Self, String> {
        let mut fs = FlatScene::default();
        fs.background = [background[0], background[1], background[2]];
        let (root, height) = world.flatten_world(&mut fs)?; // flatten_hittable.rs
        fs.root = root;
        fs.bvh_height = height;
        match lights {
            None => fs.lights_kind = LIGHTS_NONE,
            Some(Hittable::List(items)) => {
                fs.lights_kind = LIGHTS_LIST;
                for l in items {
                    l.push_light(&mut fs);
                }
            }
            Some(single) => {
                fs.lights_kind = LIGHTS_SINGLE;
                single.push_light(&mut fs);
            }
        }
        Ok(fs)
    }

    pub fn desc(&self) -> RtSceneDesc {
        RtSceneDesc {
            abi_version: RT_ABI_VERSION,
            bvh_height: self.bvh_height,
            n_nodes: self.node_left.len() as u32,
            root: self.root,
            node_box: self.node_box.as_ptr(),
            node_left: self.node_left.as_ptr(),
            node_right: self.node_right.as_ptr(),
            n_spheres: self.sphere_material.len() as u32,
            sphere: self.sphere.as_ptr(),
            sphere_material: self.sphere_material.as_ptr(),
            sphere_object: self.sphere_object.as_ptr(),
            n_mspheres: self.msphere_material.len() as u32,
            msphere: self.msphere.as_ptr(),
            msphere_material: self.msphere_material.as_ptr(),
            msphere_object: self.msphere_object.as_ptr(),
            n_cubes: self.cube_material.len() as u32,
            cube: self.cube.as_ptr(),
            cube_material: self.cube_material.as_ptr(),
            cube_object: self.cube_object.as_ptr(),
            n_complex: self.complex_body.len() as u32,
            complex_body: self.complex_body.as_ptr(),
            complex_op_first: self.complex_op_first.as_ptr(),
            complex_op_count: self.complex_op_count.as_ptr(),
            complex_material: self.complex_material.as_ptr(),
            complex_object: self.complex_object.as_ptr(),
            n_ops: self.op_tag.len() as u32,
            op_tag: self.op_tag.as_ptr(),
            op_data: self.op_data.as_ptr(),
            n_materials: self.mat_tag.len() as u32,
            mat_tag: self.mat_tag.as_ptr(),
            mat_texture: self.mat_texture.as_ptr(),
            mat_params: self.mat_params.as_ptr(),
            n_textures: self.tex_tag.len() as u32,
            tex_tag: self.tex_tag.as_ptr(),
            tex_color: self.tex_color.as_ptr(),
            tex_odd: self.tex_odd.as_ptr(),
            tex_even: self.tex_even.as_ptr(),
            tex_texel_offset: self.tex_texel_offset.as_ptr(),
            tex_width: self.tex_width.as_ptr(),
            tex_height: self.tex_height.as_ptr(),
            n_texel_bytes: self.texels.len() as u64,
            texels: self.texels.as_ptr(),
            lights_kind: self.lights_kind,
            n_lights: self.light_tag.len() as u32,
            light_tag: self.light_tag.as_ptr(),
            light_sphere: self.light_sphere.as_ptr(),
            light_box: std::ptr::null(),
            background: self.background,
        }
    }
}

use crate::material::{Dielectric, Material, Metal};
use crate::texture::Texture;

impl FlatScene {
    /// texture.rs:7-14 -> tex_* arrays; a Checker's children get larger indices than their parent.
    pub fn add_texture(&mut self, t: &Texture) -> u32 {
        let me = self.tex_tag.len();
        self.tex_tag.push(TEX_SOLID);
        self.tex_color.extend_from_slice(&[0.0, 0.0, 0.0]);
        self.tex_odd.push(RT_REF_NONE);
        self.tex_even.push(RT_REF_NONE);
        self.tex_texel_offset.push(0);
        self.tex_width.push(0);
        self.tex_height.push(0);
        match t {
            Texture::Solid(c) => {
                self.tex_color[3 * me] = c[0];
                self.tex_color[3 * me + 1] = c[1];
                self.tex_color[3 * me + 2] = c[2];
            }
            Texture::Checker((odd, even)) => {
                self.tex_tag[me] = TEX_CHECKER;
                let o = self.add_texture(odd);
                let e = self.add_texture(even);
                self.tex_odd[me] = o;
                self.tex_even[me] = e;
            }
            Texture::Image(img) => {
                self.tex_tag[me] = TEX_IMAGE;
                self.tex_width[me] = img.width();
                self.tex_height[me] = img.height();
                self.tex_texel_offset[me] = self.texels.len() as u64;
                self.texels.extend_from_slice(img.as_raw()); // RGB8 row-major
            }
            Texture::Normal => self.tex_tag[me] = TEX_NORMAL,
            Texture::UV => self.tex_tag[me] = TEX_UV,
        }
        me as u32
    }

    /// material.rs:23-30 -> mat_* arrays.
    pub fn add_material(&mut self, m: &Material) -> u32 {
        let me = self.mat_tag.len();
        self.mat_tag.push(MAT_LAMBERTIAN);
        self.mat_texture.push(RT_REF_NONE);
        self.mat_params.extend_from_slice(&[0.0; 4]);
        match m {
            Material::Lambertian(t) => self.mat_texture[me] = self.add_texture(t),
            Material::Dielectric(Dielectric { ir }) => {
                self.mat_tag[me] = MAT_DIELECTRIC;
                self.mat_params[4 * me] = *ir;
            }
            Material::Metal(Metal { albedo, fuzz }) => {
                self.mat_tag[me] = MAT_METAL;
                self.mat_params[4 * me] = albedo[0];
                self.mat_params[4 * me + 1] = albedo[1];
                self.mat_params[4 * me + 2] = albedo[2];
                self.mat_params[4 * me + 3] = *fuzz;
            }
            Material::DiffuseLight(t) => {
                self.mat_tag[me] = MAT_DIFFUSE_LIGHT;
                self.mat_texture[me] = self.add_texture(t);
            }
            Material::Isotropic(t) => {
                self.mat_tag[me] = MAT_ISOTROPIC;
                self.mat_texture[me] = self.add_texture(t);
            }
        }
        me as u32
    }
}

fn check(code: c_int) -> Result<(), String> {
    if code == 0 {
        Ok(())
    } else {
        Err(unsafe { CStr::from_ptr(rt_last_error()) }.to_string_lossy().into_owned())
    }
}

/// Replaces main.rs:152-174 (the rayon closure, `/= spp`, `pixel_from_color`, `put_pixel`).
/// Uses every GPU of the node (`rt_render_multi` shards the samples); `seed` keys the Philox streams.
pub fn render(world: &Hittable, lights: &Option<Hittable>, background: Color, cam: &Camera, image_width: u32,
              image_height: u32, samples_per_pixel: u32, max_depth: u32, seed: u64) -> Result<RgbImage, String> {
    let flat = FlatScene::new(world, lights, background)?;
    let desc = flat.desc();
    let cam_c = cam.to_gpu(); // flatten_camera.rs
    let params = RtParams { width: image_width, height: image_height, spp: samples_per_pixel, max_depth, seed,
                            sample_offset: 0, spp_per_launch: 0, flags: 0 };
    let n_px = image_width as usize * image_height as usize;
    let mut accum = vec![0f32; n_px * 3];
    let devices: Vec<c_int> = (0..unsafe { rt_device_count() }.max(1)).collect();
    check(unsafe { rt_render_multi(devices.len() as c_int, devices.as_ptr(), &desc, &cam_c, &params,
                                   accum.as_mut_ptr(), std::ptr::null_mut()) })?;
    let mut rgb8 = vec![0u8; n_px * 3]; // row-major [y][x][rgb], the layout of image::RgbImage
    unsafe { rt_tonemap(accum.as_ptr(), n_px as u64, samples_per_pixel as u64, rgb8.as_mut_ptr()) };
    RgbImage::from_raw(image_width, image_height, rgb8).ok_or_else(|| "image size mismatch".to_string())
}
