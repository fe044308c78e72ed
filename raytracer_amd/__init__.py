"""raytracer_amd — the per-pixel path-tracing hot path of dr-BEat/raytracer as gfx950 HIP
kernels behind a C ABI (include/rt.h), with a C++17 host layer (include/rt_host.h).

Importing this package loads the in-tree native libraries and raises if they are missing:
there is no CPU fallback."""
from . import _ffi
from .api import (CLI, SCENE_IDS, DeviceScene, HostScene, MultiScene, image_height, load_earthmap, render, render_multi, multi_shard, texture_from_file, tonemap,
                  write_earthmap_ppm, write_png)

__all__ = ["CLI", "write_earthmap_ppm", "write_png", "SCENE_IDS", "DeviceScene", "HostScene", "MultiScene", "multi_shard", "image_height", "load_earthmap", "render", "render_multi", "texture_from_file", "tonemap", "_ffi"]
