"""The reference's command line (struct Opts + main(), main.rs:28-49, 107-178) and its image.save."""
import os
import shutil
import subprocess

import numpy as np
import pytest
from PIL import Image

import raytracer_amd as R


def run(*args, cwd=None):
    return subprocess.run([R.CLI, *args], capture_output=True, text=True, cwd=cwd, timeout=300)


def test_help_lists_the_reference_options_and_defaults():
    p = run("--help")
    assert p.returncode == 0
    for flag, default in (("-o, --output <output>", "renders/image.png"), ("-s, --scene <scene>", "3"),
                          ("-i, --image-width <image-width>", "1200"), ("-a, --aspect-ratio <aspect-ratio>", "1.5"),
                          ("--samples-per-pixel <samples-per-pixel>", "500"), ("--max-depth <max-depth>", "5")):
        line = [l for l in p.stdout.splitlines() if flag in l]
        assert line and f"[default: {default}]" in line[0], flag
    assert run("-V").stdout.strip() == "raytracer 1.0"  # #[clap(version = "1.0")]


@pytest.mark.parametrize("args", [["--bogus"], ["-i", "abc"], ["--scene"], ["-a", "x"], ["--max-depth", "-1"]])
def test_bad_arguments_exit_2_like_clap(args):
    p = run(*args)
    assert p.returncode == 2 and p.stderr.startswith("error:") and p.stdout == ""


def test_degenerate_image_is_an_error_not_a_crash(tmp_path):
    p = run("-i", "1", "-o", str(tmp_path / "x.png"))
    assert p.returncode == 1 and p.stdout.splitlines()[0] == "1 0"  # main.rs:114 prints before anything else


@pytest.mark.parametrize("shape", [(1, 1), (23, 37), (200, 300)])  # (decoded back with PIL)
def test_png_roundtrip(tmp_path, shape):
    rgb8 = np.random.default_rng(shape[0]).integers(0, 256, size=shape + (3,), dtype=np.uint8)
    path = str(tmp_path / "img.png")
    R.write_png(path, rgb8)
    with Image.open(path) as im:
        assert im.mode == "RGB" and im.size == (shape[1], shape[0])
        assert np.array_equal(np.asarray(im), rgb8)


def test_png_compresses_images_and_decodes_to_the_same_pixels(tmp_path):
    """The writer's own deflate (row filters, LZ77 + fixed Huffman codes): smooth pictures shrink, every picture decodes
    (with PIL's zlib) to exactly the bytes written; noise falls back to stored blocks (test_png_roundtrip)."""
    yy, xx = np.mgrid[0:200, 0:300]
    cases = {"gradient": np.stack([xx % 256, yy % 256, (xx + yy) % 256], -1).astype(np.uint8), "flat": np.full((64, 64, 3), 77, np.uint8),
             "earth": R.load_earthmap(), "stripes": np.tile(np.array([[0, 0, 0], [255, 255, 255]], np.uint8), (50, 75, 1)).reshape(50, 150, 3)}
    for name, rgb8 in cases.items():
        path = str(tmp_path / (name + ".png"))
        R.write_png(path, np.ascontiguousarray(rgb8))
        with Image.open(path) as im:
            assert np.array_equal(np.asarray(im.convert("RGB")), rgb8), name
        assert os.path.getsize(path) < 0.6 * rgb8.size, (name, os.path.getsize(path), rgb8.size)


def test_png_errors(tmp_path):
    with pytest.raises(Exception):
        R.write_png(str(tmp_path / "no_such_dir" / "img.png"), np.zeros((2, 2, 3), np.uint8))  # image.save(..).unwrap()
    with pytest.raises(ValueError):
        R.write_png(str(tmp_path / "img.png"), np.zeros((2, 2), np.uint8))


def test_earthmap_ppm(tmp_path):
    path = str(tmp_path / "assets" / "earthmap.ppm")
    R.write_earthmap_ppm(path)
    with Image.open(path) as im:
        assert np.array_equal(np.asarray(im), R.load_earthmap())


@pytest.mark.gpu
@pytest.mark.parametrize("scene,texture", [(5, None), (2, "ppm"), (2, "jpg")])
def test_cli_renders_what_the_api_renders(tmp_path, scene, texture):
    args = ["-s", str(scene), "-i", "96", "--samples-per-pixel=16", "--max-depth", "5"]
    if texture == "ppm":  # the API's texels, byte for byte
        R.write_earthmap_ppm(str(tmp_path / "assets" / "earthmap.ppm"))
        args += ["--earthmap", "assets/earthmap.ppm"]
    if texture == "jpg":  # what the reference does (scenes.rs:126): assets/earthmap.jpg, decoded by the host layer
        os.makedirs(tmp_path / "assets")
        shutil.copy(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "earthmap.jpg"),
                    tmp_path / "assets" / "earthmap.jpg")
    os.makedirs(tmp_path / "renders")
    p = run(*args, cwd=str(tmp_path))  # default output renders/image.png, default aspect 1.5
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert lines[0] == "96 64" and lines[1].startswith("Rendered in ") and lines[1].endswith(" seconds")
    assert lines[2] == "Created image!"
    with Image.open(tmp_path / "renders" / "image.png") as im:
        got = np.asarray(im).astype(np.int32)
    want, _, _ = R.render(scene=scene, image_width=96, aspect_ratio=1.5, samples_per_pixel=16, max_depth=5)
    # same draws; the leaf-list kernel sums a pixel's units with float atomics, so a channel may land on
    # the other side of a quantisation step
    # (jpg: 0.25 % of the texels differ by 1-3 LSB from the decode the API ships)
    assert got.shape == want.shape and np.abs(got - want).max() <= (3 if texture == "jpg" else 1) and (got != want).mean() < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("scene", [6, 0, 3])
def test_cli_in_the_references_precision_writes_the_oracles_image(tmp_path, scene):
    """`--precision f64` (the audit kernels + an f64 tone map: main.rs:160-168, 51-63 in the reference's own arithmetic):
    the PNG the command line writes is the CPU oracle's image of the same scene, seeds and sample count — pixel for
    pixel, up to a quantisation step where libm's last bit differs."""
    import oracle_ffi as O
    os.makedirs(tmp_path / "renders")
    w, asp, spp, depth = 72, 1.5, 8, 6
    p = run("-s", str(scene), "-i", str(w), "-a", str(asp), f"--samples-per-pixel={spp}", "--max-depth", str(depth),
            "--precision", "f64", "--seed", "3", cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr
    with Image.open(tmp_path / "renders" / "image.png") as im:
        got = np.asarray(im).astype(np.int32)
    h = R.image_height(w, asp)
    o, _, _ = O.OracleScene(scene, asp).render(w, h, spp, depth, seed=3)
    g = np.sqrt(o / spp)
    want = np.where(np.isnan(g), 0, np.clip(g, 0.0, 0.9999) * 256.0).astype(np.uint8).astype(np.int32)  # main.rs:51-63
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1 and (got != want).mean() <= 2e-3, ((got != want).mean(), np.abs(got - want).max())
    # and through the Python form of main()
    rgb8, _, _ = R.render(scene=scene, image_width=w, aspect_ratio=asp, samples_per_pixel=spp, max_depth=depth, seed=3, f64=True)
    # (two f64 renders: a pixel's units are summed by atomics in the order they finish, so a channel may land on the
    # other side of a quantisation step)
    d = np.abs(rgb8.astype(np.int32) - got)
    assert d.max() <= 1 and (d != 0).mean() <= 1e-3


@pytest.mark.gpu
def test_cli_missing_texture_and_missing_directory(tmp_path):
    p = run("-s", "2", "-i", "32", "--samples-per-pixel", "1", cwd=str(tmp_path))
    assert p.returncode == 1 and "earth texture" in p.stderr
    p = run("-s", "5", "-i", "32", "--samples-per-pixel", "1", cwd=str(tmp_path))  # no renders/ directory
    assert p.returncode == 1 and "cannot write" in p.stderr and "Created image!" not in p.stdout


@pytest.mark.gpu
def test_cli_device_all_equals_device_0_on_a_one_gpu_box(tmp_path):
    """--device all shards the samples over every GPU of the node (rt_render_multi); with one GPU that is the
    same render (same Philox streams), reduced through the multi-device path."""
    os.makedirs(tmp_path / "renders")
    imgs = []
    for dev in ("0", "all"):
        p = run("-s", "6", "-i", "80", "-a", "1", "--samples-per-pixel", "32", "--max-depth", "20", "--device", dev,
                "-o", f"renders/{dev}.png", cwd=str(tmp_path))
        assert p.returncode == 0, p.stderr
        with Image.open(tmp_path / "renders" / f"{dev}.png") as im:
            imgs.append(np.asarray(im).astype(np.int32))
    assert np.abs(imgs[0] - imgs[1]).max() <= 1 and (imgs[0] != imgs[1]).mean() < 0.01
