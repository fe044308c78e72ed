// `raytracer` — the reference's command line (struct Opts + main(), main.rs:28-49, 107-178) over
// librt_host.so / librt_hip.so: same options, defaults, stdout lines and output file; the render runs on
// the GPU through include/rt.h.  Options after the reference's own are extensions and say so in --help.
//
// Differences a user can see: the PNG is compressed by the host layer's own small deflate (same pixels, other bytes than the image crate's);
// the earth texture is opened where the reference opens it (assets/earthmap.jpg, scenes.rs:126) and decoded by the
// host layer's baseline-JPEG decoder (rth_texture_from_image_file; texels within +-1 LSB of other decoders), or from
// `--earthmap <file>` (a baseline JPEG or a binary PPM); the reference's panics (bad numbers, unwritable output,
// missing texture) are error messages with exit status 2 / 1.
#include <cctype>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/rt_host.h"

namespace {

struct Opts {  // main.rs:28-49
  std::string output = "renders/image.png";
  uint32_t scene = 3;
  uint32_t image_width = 1200;
  double aspect_ratio = 1.5;
  uint32_t samples_per_pixel = 500;
  uint32_t max_depth = 5;
  // extensions
  int device = 0;
  unsigned precision = 0;  // --precision f32 | f64
  uint64_t seed = 1, scene_seed = 5, bvh_seed = 7;
  uint32_t n = 0;
  std::string earthmap = "assets/earthmap.jpg";  // scenes.rs:126
};

const char* USAGE =
    "raytracer 1.0\nBEat\nA cool raytracer!\n\n"
    "USAGE:\n    raytracer [OPTIONS]\n\n"
    "FLAGS:\n"
    "    -h, --help       Prints help information\n"
    "    -V, --version    Prints version information\n\n"
    "OPTIONS:\n"
    "    -a, --aspect-ratio <aspect-ratio>              [default: 1.5]\n"
    "    -i, --image-width <image-width>                [default: 1200]\n"
    "        --max-depth <max-depth>                    [default: 5]\n"
    "    -o, --output <output>                          Sets an output file to render too [default: renders/image.png]\n"
    "        --samples-per-pixel <samples-per-pixel>    [default: 500]\n"
    "    -s, --scene <scene>                            [default: 3]\n"
    "\nEXTENSIONS (not in the reference):\n"
    "        --device <n|all>      GPU to render on, or every GPU of the node (samples sharded) [default: 0]\n"
    "        --precision <f32|f64> f32: the product kernels; f64: the audit kernels, the reference's own arithmetic\n"
    "                              (slower; one device) [default: f32]\n"
    "        --seed <n>            Philox key of the per-path draws [default: 1]\n"
    "        --scene-seed <n>      scene generator seed (scenes.rs:64 uses 5) [default: 5]\n"
    "        --bvh-seed <n>        BVH split-axis stream [default: 7]\n"
    "        --n <n>               scene 7 (bvh_stress): number of spheres\n"
    "        --earthmap <file>     the earth texture of scenes 2, 4, 8, 11, 16: a baseline JPEG or a binary PPM (P6)\n"
    "                              [default: assets/earthmap.jpg]\n"
    "    scenes: 0 random_scene, 1 two_spheres, 2 earth, 3 simple_light, 4 cube_scene, 6 cornell_box,\n"
    "            7 bvh_stress, 8 final_scene, 9 mixed_scene (test generator), 10 perlin_spheres and\n"
    "            11 final_scene with the Perlin sphere (extensions), anything else small_scene (main.rs:117-124)\n";

[[noreturn]] void usage_error(const std::string& msg) {
  std::fprintf(stderr, "error: %s\n\nUSAGE:\n    raytracer [OPTIONS]\n\nFor more information try --help\n", msg.c_str());
  std::exit(2);
}

bool parse_u64(const std::string& s, uint64_t max, uint64_t& out) {
  if (s.empty() || s[0] == '-' || s[0] == '+') return false;
  char* end = nullptr;
  errno = 0;
  unsigned long long v = std::strtoull(s.c_str(), &end, 10);
  if (errno || *end || v > max) return false;
  out = v;
  return true;
}
bool parse_f64(const std::string& s, double& out) {
  if (s.empty()) return false;
  char* end = nullptr;
  out = std::strtod(s.c_str(), &end);
  return *end == 0;
}

}  // namespace

int main(int argc, char** argv) {
  Opts o;
  for (int i = 1; i < argc; ++i) {
    std::string arg = argv[i], name = arg, value;
    bool have_value = false;
    if (arg.rfind("--", 0) == 0) {
      const size_t eq = arg.find('=');
      if (eq != std::string::npos) {
        name = arg.substr(0, eq);
        value = arg.substr(eq + 1);
        have_value = true;
      }
    } else if (arg.size() > 2 && arg[0] == '-') {  // -i800
      name = arg.substr(0, 2);
      value = arg.substr(2);
      if (!value.empty() && value[0] == '=') value.erase(0, 1);
      have_value = true;
    }
    if (name == "-h" || name == "--help") {
      std::fputs(USAGE, stdout);
      return 0;
    }
    if (name == "-V" || name == "--version") {
      std::puts("raytracer 1.0");
      return 0;
    }
    auto need = [&]() -> const std::string& {
      if (!have_value) {
        if (i + 1 >= argc) usage_error("The argument '" + name + " <value>' requires a value but none was supplied");
        value = argv[++i];
        have_value = true;
      }
      return value;
    };
    auto u32 = [&](uint32_t& dst) {
      uint64_t v;
      if (!parse_u64(need(), 0xFFFFFFFFull, v)) usage_error("Invalid value for '" + name + "': '" + value + "' is not a u32");
      dst = (uint32_t)v;
    };
    auto u64 = [&](uint64_t& dst) {
      if (!parse_u64(need(), ~0ull, dst)) usage_error("Invalid value for '" + name + "': '" + value + "' is not a u64");
    };
    if (name == "-o" || name == "--output") {
      o.output = need();
    } else if (name == "-s" || name == "--scene") {
      u32(o.scene);
    } else if (name == "-i" || name == "--image-width") {
      u32(o.image_width);
    } else if (name == "-a" || name == "--aspect-ratio") {
      if (!parse_f64(need(), o.aspect_ratio)) usage_error("Invalid value for '" + name + "': '" + value + "' is not a float");
    } else if (name == "--samples-per-pixel") {
      u32(o.samples_per_pixel);
    } else if (name == "--max-depth") {
      u32(o.max_depth);
    } else if (name == "--device") {
      if (need() == "all") {
        o.device = -1;
      } else {
        uint32_t d = 0;
        u32(d);
        o.device = (int)d;
      }
    } else if (name == "--precision") {
      const std::string v = need();
      if (v == "f32")
        o.precision = 0;
      else if (v == "f64")
        o.precision = 1;
      else
        usage_error("Invalid value for '" + name + "': '" + v + "' is not f32 or f64");
    } else if (name == "--seed") {
      u64(o.seed);
    } else if (name == "--scene-seed") {
      u64(o.scene_seed);
    } else if (name == "--bvh-seed") {
      u64(o.bvh_seed);
    } else if (name == "--n") {
      u32(o.n);
    } else if (name == "--earthmap") {
      o.earthmap = need();
    } else {
      usage_error("Found argument '" + arg + "' which wasn't expected, or isn't valid in this context");
    }
  }

  const uint32_t image_width = o.image_width;
  const uint32_t image_height = rth_image_height(image_width, o.aspect_ratio);  // main.rs:112
  std::printf("%u %u\n", image_width, image_height);                            // main.rs:114
  std::fflush(stdout);
  if (image_width < 2 || image_height < 2) {  // (image_width - 1) as f64 divides by zero / ImageBuffer of nothing
    std::fprintf(stderr, "error: the image must be at least 2x2 pixels (got %ux%u)\n", image_width, image_height);
    return 1;
  }

  rth_render_opts ro;
  std::memset(&ro, 0, sizeof(ro));
  ro.scene.scene_id = (int32_t)(o.scene > 0x7FFFFFFFu ? 0x7FFFFFFF : o.scene);
  ro.scene.rotate_bbox = -1;
  ro.scene.aspect_ratio = o.aspect_ratio;
  ro.scene.scene_seed = o.scene_seed;
  ro.scene.bvh_seed = o.bvh_seed;
  ro.scene.n = o.n;
  uint8_t* texels = nullptr;
  struct TexelGuard {
    uint8_t*& p;
    ~TexelGuard() { rth_texture_free(p); }
  } texel_guard{texels};
  if (o.scene == 2 || o.scene == 4 || o.scene == 8 || o.scene == 11 || o.scene == 16) {  // Texture::new_image (texture.rs:25-31, scenes.rs:126)
    if (rth_texture_from_image_file(o.earthmap.c_str(), &ro.scene.tex_width, &ro.scene.tex_height, &texels) != RT_OK) {
      std::fprintf(stderr, "error: scene %u needs the earth texture (scenes.rs:126): %s\n", o.scene, rth_last_error());
      return 1;
    }
    ro.scene.texels = texels;
  }
  ro.image_width = image_width;
  ro.samples_per_pixel = o.samples_per_pixel;
  ro.max_depth = o.max_depth;
  ro.device = o.device;
  ro.seed = o.seed;
  ro.precision = o.precision;

  std::vector<uint8_t> rgb8((size_t)image_width * image_height * 3);
  rt_stats st;
  std::memset(&st, 0, sizeof(st));
  uint32_t h_out = 0;
  const auto now = std::chrono::steady_clock::now();  // main.rs:153
  if (rth_render(&ro, rgb8.data(), nullptr, &h_out, &st) != RT_OK) {
    std::fprintf(stderr, "error: %s\n", rth_last_error());
    return 1;
  }
  const float secs = std::chrono::duration<float>(std::chrono::steady_clock::now() - now).count();
  std::printf("Rendered in %g seconds\n", (double)secs);  // main.rs:175
  std::fprintf(stderr, "(%llu rays, %.1f ms in kernels, %.1f Mrays/s)\n", (unsigned long long)st.rays, st.kernel_ms,
               st.kernel_ms > 0 ? (double)st.rays / st.kernel_ms * 1e-3 : 0.0);
  if (rth_write_png(o.output.c_str(), rgb8.data(), image_width, image_height) != RT_OK) {  // main.rs:176
    std::fprintf(stderr, "error: %s\n", rth_last_error());
    return 1;
  }
  std::puts("Created image!");  // main.rs:177
  return 0;
}
