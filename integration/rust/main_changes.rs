// Changes to src/main.rs (line numbers of the reference at the commit this repository was built against).
//
// 1. next to the other `mod` lines (main.rs:11-23):
mod gpu;

// 2. delete main.rs:152 (`let mut image ... ImageBuffer::new`) and main.rs:154-174 (the `pixels` pipeline:
//    flat_map / par_iter / the per-pixel sample loop / `/= spp` / pixel_from_color, and the put_pixel loop),
//    keep `let now = Instant::now();` (main.rs:153), and put this in their place:
let image: RgbImage = gpu::render(
    &world, &lights, background, &cam,
    image_width, image_height, opts.samples_per_pixel, opts.max_depth,
    1, // Philox key of the per-path draws: renders are reproducible, unlike thread_rng
)
.expect("GPU render failed");

// 3. main.rs:175-177 (`Rendered in`, `image.save`, `Created image!`) stay as they are.
//    ray_color, pixel_from_color and the rayon / rand imports become unused (keep them for a CPU fallback
//    or delete them with their `use` lines).
