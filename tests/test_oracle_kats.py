"""The oracle (and the product's host layer) against the reference's own known-answer vectors
(tests/golden/reference_kats.json: hittable.rs:547-552, alias.rs:188-214) and Random123 Philox KATs."""
import json
import math
import os

import numpy as np

import oracle_ffi as O

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
TOL = KATS["tolerance"]


def test_oracle_get_sphere_uv():
    for k in KATS["get_sphere_uv"]:
        assert np.abs(O.get_sphere_uv(k["p"]) - np.array(k["uv"])).max() < TOL


def test_oracle_rotate():
    for k in KATS["rotate"]:
        got = O.rotate(math.radians(k["angle_deg"]), k["axis"], k["p"])
        assert np.abs(got - np.array(k["result"])).max() < TOL


def test_oracle_rotate_identity_property():
    ident = KATS["rotate_identity"]
    rng = np.random.default_rng(0)
    for angle in rng.uniform(*ident["angle_range_rad"], size=256):
        assert np.abs(O.rotate(angle, ident["axis"], ident["p"]) - np.array(ident["p"])).max() < TOL


def test_host_rotate_matches_kats_and_oracle():
    import ctypes as C
    from raytracer_amd import _ffi as F
    rng = np.random.default_rng(1)
    cases = [(math.radians(k["angle_deg"]), k["axis"], k["p"], k["result"]) for k in KATS["rotate"]]
    for angle, axis, p, want in cases:
        axis, p = np.array(axis, dtype=np.float64), np.array(p, dtype=np.float64)
        out = np.zeros(3)
        F.host.rth_rotate(angle, axis.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p),
                          out.ctypes.data_as(C.c_void_p))
        assert np.abs(out - np.array(want)).max() < TOL
    for _ in range(64):
        angle, axis, p = rng.uniform(-6, 6), rng.normal(size=3), rng.normal(size=3)
        out = np.zeros(3)
        F.host.rth_rotate(angle, axis.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p),
                          out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out, O.rotate(angle, axis, p))  # same arithmetic, two implementations


def test_oracle_philox_kats():
    for rounds in (7, 10):  # 7: per-path draw streams, 10: scene and BVH-axis streams
        for k in KATS["philox4x32_%d" % rounds]:
            assert list(O.philox(k["ctr"], k["key"], rounds)) == k["out"]


def test_streams_are_uniform_and_keyed():
    a = O.path_draws(1, 10, 3, 1, 4096)
    b = O.path_draws(1, 10, 3, 2, 4096)
    c = O.path_draws(2, 10, 3, 1, 4096)
    assert 0.0 <= a.min() and a.max() < 1.0
    assert abs(a.mean() - 0.5) < 0.02 and abs(a.var() - 1 / 12) < 0.01
    assert not np.array_equal(a, b) and not np.array_equal(a, c)
    # draws are words of consecutive Philox blocks of the (pixel, sample, slot) stream
    blk = O.philox([10, 3, 1, 0], [1, 0], rounds=7)
    assert np.array_equal(a[:4], blk.astype(np.float64) / 2**32)


def test_bvh_axis_stream_matches_host():
    from raytracer_amd import _ffi as F
    for seed in (0, 7, 2**40 + 3):
        for call in range(200):
            assert F.host.rth_bvh_axis(seed, call) == O.lib().orc_bvh_axis(seed, call)
    axes = [O.lib().orc_bvh_axis(7, c) for c in range(3000)]
    assert set(axes) == {0, 1, 2}
    assert all(abs(axes.count(a) / 3000 - 1 / 3) < 0.05 for a in (0, 1, 2))
