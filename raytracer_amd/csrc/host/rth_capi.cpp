// C ABI of librt_host.so (include/rt_host.h).
#include "jpeg_baseline.hpp"
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <stdexcept>
#include <string>

#include "../../../include/rt_host.h"
#include "rth.hpp"
#include "std_rng.hpp"
#include "png_write.hpp"

using namespace rth;

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
}  // namespace

struct rth_scene {
  FlatScene flat;
  rt_scene_desc desc;
  rt_camera camera;
  uint32_t n_objects;
};

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

// image.save(path) of main.rs:176 for an RGB8 image (PNG only: the reference's default output is a .png).
int rth_write_png(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
  if (!path || !rgb8 || width == 0 || height == 0) return fail(RT_ERR_BAD_ARG, "rth_write_png: bad argument");
  if (!rth::write_file(path, rth::encode_png_rgb8(rgb8, width, height)))
    return fail(RT_ERR_BAD_ARG, std::string("rth_write_png: cannot write ") + path);
  return RT_OK;
}

// Texture::new_image (texture.rs:25-31) for the one format this layer decodes itself: a binary PPM
// ("P6 <w> <h> 255\n" + w*h*3 bytes, '#' comments allowed in the header).  The reference decodes
// assets/earthmap.jpg with the `image` crate (jpeg-decoder 0.1.22); no JPEG decoder is linked here, so the file is
// decoded once, offline, to PPM / NPZ (raytracer_amd.write_earthmap_ppm; +-1 LSB against the crate is possible).
int rth_texture_from_rgb8_file(const char* path, uint32_t* width, uint32_t* height, uint8_t** texels) {
  if (!path || !width || !height || !texels) return fail(RT_ERR_BAD_ARG, "rth_texture_from_rgb8_file: null argument");
  *texels = nullptr;
  *width = *height = 0;
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(RT_ERR_BAD_ARG, std::string("rth_texture_from_rgb8_file: cannot open '") + path + "' (texture.rs:29 would panic)");
  auto token = [&](std::string& t) {
    t.clear();
    int c;
    for (;;) {
      c = std::fgetc(f);
      if (c == '#') {
        while (c != '\n' && c != EOF) c = std::fgetc(f);
      } else if (c == EOF || !std::isspace(c)) {
        break;
      }
    }
    while (c != EOF && !std::isspace(c)) {
      t.push_back((char)c);
      c = std::fgetc(f);
    }
    return !t.empty();
  };
  auto number = [](const std::string& t, unsigned long limit, unsigned long& out) {
    if (t.empty() || t.size() > 9) return false;
    for (char ch : t)
      if (ch < '0' || ch > '9') return false;
    out = std::strtoul(t.c_str(), nullptr, 10);
    return out <= limit;
  };
  std::string magic, sw, sh, smax;
  unsigned long vw = 0, vh = 0, vmax = 0;
  bool ok = token(magic) && magic == "P6" && token(sw) && token(sh) && token(smax) && number(sw, 1ul << 16, vw) &&
            number(sh, 1ul << 16, vh) && number(smax, 255, vmax) && vmax == 255 && vw > 0 && vh > 0;
  uint8_t* buf = nullptr;
  if (ok) {
    const size_t n = (size_t)vw * vh * 3;
    buf = (uint8_t*)std::malloc(n);
    ok = buf && std::fread(buf, 1, n, f) == n;
  }
  std::fclose(f);
  if (!ok) {
    std::free(buf);
    return fail(RT_ERR_BAD_ARG, std::string("rth_texture_from_rgb8_file: '") + path +
                                    "' is not a complete binary PPM (P6, maxval 255, at most 65536 x 65536)");
  }
  *texels = buf;
  *width = (uint32_t)vw;
  *height = (uint32_t)vh;
  return RT_OK;
}
void rth_texture_free(uint8_t* texels) { std::free(texels); }

int rth_texture_from_image_file(const char* path, uint32_t* width, uint32_t* height, uint8_t** texels) {
  if (!path || !width || !height || !texels) return fail(RT_ERR_BAD_ARG, "rth_texture_from_image_file: null argument");
  *texels = nullptr;
  *width = *height = 0;
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(RT_ERR_BAD_ARG, std::string("rth_texture_from_image_file: cannot open '") + path + "' (texture.rs:29 would panic)");
  unsigned char magic[2] = {0, 0};
  const size_t got = std::fread(magic, 1, 2, f);
  if (got == 2 && magic[0] == 'P' && magic[1] == '6') {
    std::fclose(f);
    return rth_texture_from_rgb8_file(path, width, height, texels);
  }
  if (got != 2 || magic[0] != 0xFF || magic[1] != 0xD8) {
    std::fclose(f);
    return fail(RT_ERR_BAD_ARG, std::string("rth_texture_from_image_file: '") + path + "' is neither a JPEG nor a binary PPM (P6)");
  }
  std::vector<uint8_t> bytes;
  std::fseek(f, 0, SEEK_END);
  const long size = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  if (size > 0 && size < (1l << 30)) {
    bytes.resize((size_t)size);
    if (std::fread(bytes.data(), 1, bytes.size(), f) != bytes.size()) bytes.clear();
  }
  std::fclose(f);
  if (bytes.empty()) return fail(RT_ERR_BAD_ARG, std::string("rth_texture_from_image_file: cannot read '") + path + "'");
  try {
    uint32_t w = 0, h = 0;
    const std::vector<uint8_t> rgb = rth::jpeg::Decoder(bytes.data(), bytes.size()).decode(w, h);
    uint8_t* buf = (uint8_t*)std::malloc(rgb.size());
    if (!buf) return fail(RT_ERR_BAD_ARG, "rth_texture_from_image_file: out of memory");
    std::memcpy(buf, rgb.data(), rgb.size());
    *texels = buf;
    *width = w;
    *height = h;
    return RT_OK;
  } catch (const std::exception& e) {
    return fail(RT_ERR_BAD_ARG, std::string("rth_texture_from_image_file: '") + path + "': " + e.what());
  }
}

uint32_t rth_image_height(uint32_t image_width, double aspect_ratio) {
  double h = (double)image_width / aspect_ratio;  // `as u32` truncates and saturates
  if (!(h == h) || h <= 0.0) return 0;
  if (h >= 4294967295.0) return 4294967295u;
  return (uint32_t)h;
}

int rth_scene_build(const rth_scene_opts* o, rth_scene** out) {
  if (!o || !out) return fail(RT_ERR_BAD_ARG, "rth_scene_build: null argument");
  *out = nullptr;
  try {
    SceneOpts so;
    so.scene_id = o->scene_id;
    so.scene_seed = o->scene_seed;
    so.n = o->n;
    if (o->texels && o->tex_width && o->tex_height) {
      so.texels = std::make_shared<const std::vector<uint8_t>>(o->texels,
                                                               o->texels + (size_t)3 * o->tex_width * o->tex_height);
      so.tex_w = o->tex_width;
      so.tex_h = o->tex_height;
    }
    SceneDef def = scene_by_id(so);
    RotateBBox mode = def.rotate_bbox;
    if (o->rotate_bbox == 0) mode = RotateBBox::Reference;
    if (o->rotate_bbox == 1) mode = RotateBBox::Correct;
    auto* s = new rth_scene;
    s->flat = flatten(def, o->bvh_seed, mode);
    s->desc = s->flat.desc();
    s->n_objects = s->flat.n_objects;
    s->camera = camera_new(def.cam.lookfrom, def.cam.lookat, def.cam.vup, def.cam.vfov, o->aspect_ratio,
                           def.cam.aperture, def.cam.focus_dist, def.cam.time_open, def.cam.time_close);
    *out = s;
    return RT_OK;
  } catch (const std::exception& e) {
    return fail(RT_ERR_BAD_ARG, std::string("rth_scene_build: ") + e.what());
  }
}

void rth_scene_free(rth_scene* s) { delete s; }
const rt_scene_desc* rth_scene_desc(const rth_scene* s) { return s ? &s->desc : nullptr; }
const rt_camera* rth_scene_camera(const rth_scene* s) { return s ? &s->camera : nullptr; }
uint32_t rth_scene_object_count(const rth_scene* s) { return s ? s->n_objects : 0; }
uint32_t rth_scene_bvh_calls(const rth_scene* s) { return s ? s->flat.bvh.calls : 0; }

int rth_camera_new(const double lookfrom[3], const double lookat[3], const double vup[3], double vfov,
                   double aspect_ratio, double aperture, double focus_dist, double time_open, double time_close,
                   rt_camera* out) {
  if (!lookfrom || !lookat || !vup || !out) return fail(RT_ERR_BAD_ARG, "rth_camera_new: null argument");
  *out = camera_new(Point(lookfrom[0], lookfrom[1], lookfrom[2]), Point(lookat[0], lookat[1], lookat[2]),
                    Vector(vup[0], vup[1], vup[2]), vfov, aspect_ratio, aperture, focus_dist, time_open, time_close);
  return RT_OK;
}

int rth_render(const rth_render_opts* o, uint8_t* rgb8, float* accum, uint32_t* height_out, rt_stats* stats) {
  if (!o || !rgb8) return fail(RT_ERR_BAD_ARG, "rth_render: null argument");
  const uint32_t w = o->image_width;
  const uint32_t h = rth_image_height(w, o->scene.aspect_ratio);
  if (height_out) *height_out = h;
  rth_scene* hs = nullptr;
  int rc = rth_scene_build(&o->scene, &hs);
  if (rc != RT_OK) return rc;
  rt_params p;
  std::memset(&p, 0, sizeof(p));
  p.width = w;
  p.height = h;
  p.spp = o->samples_per_pixel;
  p.max_depth = o->max_depth;
  p.seed = o->seed;
  std::vector<float> local;
  float* acc = accum;
  if (!acc) {
    local.resize((size_t)w * h * 3);
    acc = local.data();
  }
  if (o->precision > 1) {
    rth_scene_free(hs);
    return fail(RT_ERR_BAD_ARG, "rth_render: precision must be 0 (f32) or 1 (f64)");
  }
  if (o->precision == 1) {
    // the audit kernels: sums in f64, divided and tone-mapped in f64 like main.rs:167, 51-63
    if (o->device < 0) {
      rth_scene_free(hs);
      return fail(RT_ERR_BAD_ARG, "rth_render: the f64 kernels render on one device (device >= 0)");
    }
    std::vector<double> acc64((size_t)w * h * 3);
    rt_scene* ds = nullptr;
    rc = rt_scene_create(o->device, &hs->desc, &ds);
    if (rc == RT_OK) {
      rc = rt_render_f64(ds, &hs->camera, &p, acc64.data(), stats);
      rt_scene_destroy(ds);
    }
    if (rc != RT_OK) {
      g_err = std::string("rth_render: ") + rt_last_error();
    } else {
      for (size_t i = 0; i < acc64.size(); ++i) acc[i] = (float)acc64[i];
      rt_tonemap_f64(acc64.data(), (uint64_t)w * h, p.spp, rgb8);
    }
    rth_scene_free(hs);
    return rc;
  }
  if (o->device < 0) {
    // every GPU of the node, as rayon takes every core (main.rs:157): the samples are sharded by rt_render_multi;
    // *stats receives the sum of the shards' counters and the LONGEST shard's kernel time
    const int n_dev = rt_device_count();
    if (n_dev <= 0) {
      rth_scene_free(hs);
      return fail(RT_ERR_HIP, "rth_render: no GPU found");
    }
    std::vector<int> devices((size_t)n_dev);
    for (int g = 0; g < n_dev; ++g) devices[(size_t)g] = g;
    std::vector<rt_stats> per((size_t)n_dev);
    rc = rt_render_multi(n_dev, devices.data(), &hs->desc, &hs->camera, &p, acc, per.data());
    if (rc == RT_OK && stats) {
      std::memset(stats, 0, sizeof(*stats));
      for (const rt_stats& s : per) {
        stats->rays += s.rays;
        stats->light_probes += s.light_probes;
        stats->medium_probes += s.medium_probes;
        stats->nonfinite_samples += s.nonfinite_samples;
        stats->launches += s.launches;
        stats->kernel_ms = std::max(stats->kernel_ms, s.kernel_ms);
      }
    }
  } else {
    rt_scene* ds = nullptr;
    rc = rt_scene_create(o->device, &hs->desc, &ds);
    if (rc == RT_OK) {
      rc = rt_render(ds, &hs->camera, &p, acc, stats);
      rt_scene_destroy(ds);
    }
  }
  if (rc != RT_OK)
    g_err = std::string("rth_render: ") + rt_last_error();
  else
    rt_tonemap(acc, (uint64_t)w * h, p.spp, rgb8);
  rth_scene_free(hs);
  return rc;
}

void rth_rotate(double angle_rad, const double axis[3], const double v[3], double out[3]) {
  d3 r = rotate(d3(v[0], v[1], v[2]), new_quaternion(angle_rad, Vector(axis[0], axis[1], axis[2])));
  out[0] = r[0];
  out[1] = r[1];
  out[2] = r[2];
}
int rth_bvh_axis(uint64_t bvh_seed, uint32_t call) { return bvh_axis(bvh_seed, call); }
void rth_kat_chacha(const uint32_t* state, uint32_t rounds, uint32_t* out) {
  ChaCha::Words in;
  std::memcpy(in.data(), state, sizeof(in));
  const ChaCha::Words r = ChaCha::block(in, rounds);
  std::memcpy(out, r.data(), sizeof(r));
}
void rth_kat_chacha_quarter(uint32_t* abcd) { ChaCha::quarter(abcd[0], abcd[1], abcd[2], abcd[3]); }
void rth_kat_stdrng(uint64_t seed, uint32_t n, uint64_t* words, double* unit, double* ranged, double lo, double hi) {
  StdRng a(seed), b(seed), c(seed);
  for (uint32_t i = 0; i < n; ++i) {
    words[i] = a.next_u64();
    unit[i] = b.next_f64();
    ranged[i] = c.range(lo, hi);
  }
}

}  // extern "C"
