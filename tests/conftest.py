import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# (scene id, aspect ratio, n, max_depth) — every scene of scenes.rs plus the three config scenes
SCENES = {
    "random_scene": (0, 16.0 / 9.0, 0, 50),
    "two_spheres": (1, 1.5, 0, 5),
    "earth": (2, 1.5, 0, 5),
    "simple_light": (3, 1.5, 0, 5),
    "cube_scene": (4, 1.5, 0, 5),
    "small_scene": (5, 1.5, 0, 5),
    "cornell_box": (6, 1.0, 0, 50),
    "bvh_stress": (7, 16.0 / 9.0, 20000, 50),
    "final_scene": (8, 1.0, 0, 50),
    # extensions: Perlin noise texture
    "perlin_spheres": (10, 1.5, 0, 5),
    "final_scene_perlin": (11, 1.0, 0, 50),
    # extensions: a Cube as an importance-sampled area light (RT_LIGHT_CUBE + RT_MAT_AREA_LIGHT)
    "cornell_box_area": (12, 1.0, 0, 50),
    "area_light_plate": (13, 1.0, 0, 5),
    # random_scene with its geometry from StdRng::seed_from_u64(5) (parity unpinned: tests/test_stdrng.py)
    "random_scene_stdrng": (14, 16.0 / 9.0, 0, 50),
    "area_light_list": (15, 1.0, 0, 8),  # the box light inside a lights List (with a sphere lamp and a non-emitting bead)
    # final_scene with its 1000 small spheres as ONE Translate(Rotate(Bvh)) instance (hittable.rs:81-91 over a Bvh)
    "final_scene_instanced": (16, 1.0, 0, 50),
    # bvh_stress with every radius x 0.1: rays cross the whole volume (the bench's HBM-streaming configuration)
    "bvh_stress_sparse": (17, 16.0 / 9.0, 20000, 50),
}


@pytest.fixture(scope="session")
def earthmap():
    import raytracer_amd as R
    return R.load_earthmap()
