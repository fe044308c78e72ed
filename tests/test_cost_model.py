"""bench.py's cost table is the frozen one in oracle/cost_model.h."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_cost_table_matches_frozen_header():
    hdr = open(os.path.join(ROOT, "oracle", "cost_model.h")).read()
    frozen = {k.lower(): int(v) for k, v in re.findall(r"#define RT_COST_([A-Z_]+) (\d+)", hdr)}
    sys.path.insert(0, ROOT)
    import bench
    assert frozen == bench.COST
    assert len(frozen) == 13


def test_algorithmic_per_ray():
    sys.path.insert(0, ROOT)
    import bench
    st = dict(rays=10, nodes_visited=70, sphere_tests=10, sphere_hits=5, cube_tests=40, cube_hits=20, rotate_visits=20,
              translate_visits=20, diffuse_vertices=8, specular_vertices=0, closest_hits=9)
    flops, byts = bench.algorithmic_per_ray(st, n_pixels=4, launches=1)
    assert flops == (18 * 70 + 25 * 10 + 17 * 5 + 18 * 40 + 30 * 20 + 112 * 20 + 6 * 20 + 60 * 8) / 10
    assert byts == (32 * 70 + 16 * 50 + 32 * 9 + 12 * 4) / 10
