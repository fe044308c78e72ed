/*
 * rt_host.h — C ABI of librt_host.so: the host layer ABOVE include/rt.h.
 *
 * The reference is a Rust binary; rustc/cargo are absent from this image, so the host side
 * is C++17 (raytracer_amd/csrc/host/) and mirrors the reference's own construction API
 * (hittable.rs:113-157, material.rs:32-55, texture.rs:16-31, cam.rs:26-61, scenes.rs,
 * main.rs:107-177).  These entry points are what Python (ctypes) and the bench call; a Rust
 * host would bind include/rt.h directly and keep its own scenes.rs.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene; /* a flattened scene + its camera, host memory only */

typedef struct rth_scene_opts {
  int32_t scene_id;        /* main.rs:117-124: 0 random, 1 two_spheres, 2 earth, 3 simple_light, 4 cube_scene,
                              other small_scene; plus 6 cornell_box, 7 bvh_stress (17 bvh_stress_sparse: the same spheres at a
                              tenth of their radii), 8 final_scene, 16 final_scene_instanced, 9 mixed_scene (parity-test
                              generator), and the extensions 10 perlin_spheres, 11 final_scene with the Perlin sphere,
                              12 cornell_box_area (the ceiling light a box: RT_LIGHT_CUBE + RT_MAT_AREA_LIGHT), 13 area_light_plate, 15 area_light_list (the box light in a List of three lights);
                              14 random_scene with its geometry drawn from StdRng::seed_from_u64(scene_seed) like
                              scenes.rs:64-98 (parity unpinned, see rth_kat_stdrng) */
  int32_t rotate_bbox;     /* -1 scene default, 0 reference (Q3), 1 correct */
  double aspect_ratio;     /* main.rs:41-42 */
  uint64_t scene_seed;     /* scenes.rs:64 uses 5 */
  uint64_t bvh_seed;       /* split-axis stream (hittable.rs:161 is unseeded in the reference) */
  uint32_t n;              /* scene 7: number of random spheres */
  uint32_t tex_width, tex_height;
  const uint8_t* texels;   /* decoded assets/earthmap.jpg, RGB8 row-major (scenes 2, 4, 8, 11; 9 optionally) */
} rth_scene_opts;

/* Replaces main.rs:117-150: scene function, Hittable::new_bvh, lights fold, Camera::new. */
int rth_scene_build(const rth_scene_opts* opts, rth_scene** out);
void rth_scene_free(rth_scene* scene);
const rt_scene_desc* rth_scene_desc(const rth_scene* scene);
const rt_camera* rth_scene_camera(const rth_scene* scene);
uint32_t rth_scene_object_count(const rth_scene* scene);
uint32_t rth_scene_bvh_calls(const rth_scene* scene);

/* The texels of Texture::new_image (texture.rs:25-31: open + decode + to_rgb8) WITHOUT a decode: RGB8 in memory
 * (rth_scene_opts.texels; Python ships raytracer_amd/assets/earthmap_rgb8.npz, a PIL decode of the reference's
 * assets/earthmap.jpg) or a binary PPM (P6, maxval 255) read by this function.  For the reference's own file format see
 * rth_texture_from_image_file below.  *texels is malloc'ed RGB8 row-major [h][w][3]; release it with
 * rth_texture_free.  A missing or malformed file is RT_ERR_BAD_ARG (the reference panics, texture.rs:29).  */
int rth_texture_from_rgb8_file(const char* path, uint32_t* width, uint32_t* height, uint8_t** texels);
void rth_texture_free(uint8_t* texels);
/* Replaces Texture::new_image (texture.rs:25-31) for the file the reference itself opens (scenes.rs:126, 178:
 * assets/earthmap.jpg): a BASELINE JPEG (SOF0 / SOF1 Huffman, 8 bit, grey or YCbCr; csrc/host/jpeg_baseline.hpp) or a
 * binary PPM, told apart by their first two bytes.  The reference decodes through image 0.23.13 -> jpeg-decoder 0.1.22
 * (Cargo.lock), which are not in /root/reference: this decoder restates the published algorithm (T.81 Huffman decoding,
 * the 12-bit fixed-point LL&M inverse DCT, JFIF colour conversion) and is pinned against an independent decode of
 * the same file (tests/test_jpeg.py: 99.76 % of the texel bytes identical, 99.98 % within 1 LSB, none more than 3 off); against the crate itself parity is
 * unpinned (+-1 LSB possible).  Progressive, arithmetic-coded, 12-bit and CMYK files are refused with
 * RT_ERR_BAD_ARG and a reason.  Same ownership as above (rth_texture_free).                                        */
int rth_texture_from_image_file(const char* path, uint32_t* width, uint32_t* height, uint8_t** texels);

/* Replaces Camera::new (cam.rs:26-61). */
int rth_camera_new(const double lookfrom[3], const double lookat[3], const double vup[3], double vfov,
                   double aspect_ratio, double aperture, double focus_dist, double time_open, double time_close,
                   rt_camera* out);

typedef struct rth_render_opts {   /* = struct Opts, main.rs:28-49, plus what main() hard-codes */
  rth_scene_opts scene;
  uint32_t image_width;            /* -i, default 1200 */
  uint32_t samples_per_pixel;      /* --samples-per-pixel, default 500 */
  uint32_t max_depth;              /* --max-depth, default 5 */
  int32_t device;                  /* GPU index; negative = every GPU of the node (rt_render_multi), as rayon
                                      takes every core (main.rs:157) */
  uint64_t seed;
  uint32_t precision;              /* 0 = the f32 product kernels; 1 = the f64 audit kernels and an f64 tone map: the
                                      reference's own arithmetic (alias.rs:4-7), pixel for pixel the CPU oracle's image up to
                                      libm's last bit (one device only) */
  uint32_t reserved;
} rth_render_opts;

/* Replaces main() from "World" to the put_pixel loop (main.rs:116-174): builds the scene,
 * renders on the GPU through include/rt.h, tonemaps.  rgb8 is [h][w][3] with
 * h = trunc(image_width / aspect_ratio) (main.rs:112); accum (may be null) receives the
 * per-pixel radiance SUM.  *height_out receives h.                                        */
int rth_render(const rth_render_opts* opts, uint8_t* rgb8, float* accum, uint32_t* height_out, rt_stats* stats);
uint32_t rth_image_height(uint32_t image_width, double aspect_ratio); /* main.rs:112 */

/* Replaces image.save(opts.output) (main.rs:176) for the reference's default format: an 8-bit RGB PNG
 * (row filters + one fixed-Huffman deflate block over LZ77 matches, csrc/host/png_write.hpp; decodes to exactly rgb8).  rgb8 is [height][width][3].                      */
int rth_write_png(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);

/* Host-side known-answer hooks for the two functions the reference's tests pin. */
void rth_rotate(double angle_rad, const double axis[3], const double v[3], double out[3]);
int rth_bvh_axis(uint64_t bvh_seed, uint32_t call);
/* ... and for the generator of scene 14 (StdRng::seed_from_u64, scenes.rs:64; csrc/host/std_rng.hpp): the ChaCha block
 * function on a caller-made 16-word state (out = permuted(state) + state; pinned by RFC 7539 2.3.2 at 20 rounds, StdRng
 * uses 12), one quarter round in place (RFC 7539 2.1.1), and the first n draws of a seeded generator as raw u64 words,
 * as gen::<f64>() and as gen_range(lo..hi), each from its own copy of the generator.                                  */
void rth_kat_chacha(const uint32_t state[16], uint32_t rounds, uint32_t out[16]);
void rth_kat_chacha_quarter(uint32_t abcd[4]);
void rth_kat_stdrng(uint64_t seed, uint32_t n, uint64_t* words, double* unit, double* ranged, double lo, double hi);

const char* rth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
