"""The reference's own known-answer vectors (tests/golden/reference_kats.json) evaluated by the
DEVICE functions through the C ABI, in both precisions, plus Philox KATs on the device."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("use_f32", [0, 1])
def test_device_get_sphere_uv(use_f32):
    from raytracer_amd import _ffi as F
    p = np.array([k["p"] for k in KATS["get_sphere_uv"]], dtype=np.float64)
    want = np.array([k["uv"] for k in KATS["get_sphere_uv"]], dtype=np.float64)
    uv = np.zeros_like(want)
    F.check_hip(F.hip.rt_kat_sphere_uv(0, len(p), _ptr(p), _ptr(uv), use_f32))
    tol = 1e-6 if use_f32 else KATS["tolerance"]  # f32 cannot reach the reference's 1e-8
    assert np.abs(uv - want).max() < tol


@pytest.mark.parametrize("use_f32", [0, 1])
def test_device_rotate(use_f32):
    from raytracer_amd import _ffi as F
    ks = KATS["rotate"]
    rng = np.random.default_rng(0)
    ident = rng.uniform(*KATS["rotate_identity"]["angle_range_rad"], size=64)
    angle = np.array([math.radians(k["angle_deg"]) for k in ks] + list(ident))
    axis = np.array([k["axis"] for k in ks] + [KATS["rotate_identity"]["axis"]] * 64, dtype=np.float64)
    v = np.array([k["p"] for k in ks] + [KATS["rotate_identity"]["p"]] * 64, dtype=np.float64)
    want = np.array([k["result"] for k in ks] + [KATS["rotate_identity"]["p"]] * 64, dtype=np.float64)
    out = np.zeros_like(want)
    F.check_hip(F.hip.rt_kat_rotate(0, len(angle), _ptr(angle), _ptr(axis), _ptr(v), _ptr(out), use_f32))
    tol = 1e-6 if use_f32 else KATS["tolerance"]
    assert np.abs(out - want).max() < tol


@pytest.mark.parametrize("rounds", [7, 10])  # 7: per-path draw streams, 10: scene and BVH-axis streams
def test_device_philox(rounds):
    from raytracer_amd import _ffi as F
    import oracle_ffi as O
    ks = KATS["philox4x32_%d" % rounds]
    rng = np.random.default_rng(1)
    extra = 1000
    ctr = np.concatenate([np.array([k["ctr"] for k in ks], dtype=np.uint32),
                          rng.integers(0, 2**32, size=(extra, 4), dtype=np.uint64).astype(np.uint32)])
    key = np.concatenate([np.array([k["key"] for k in ks], dtype=np.uint32),
                          rng.integers(0, 2**32, size=(extra, 2), dtype=np.uint64).astype(np.uint32)])
    out = np.zeros_like(ctr)
    F.check_hip(F.hip.rt_kat_philox(0, len(ctr), _ptr(ctr), _ptr(key), rounds, _ptr(out)))
    for i, k in enumerate(ks):
        assert list(out[i]) == k["out"]
    for i in range(len(ks), len(ctr)):
        assert np.array_equal(out[i], O.philox(ctr[i], key[i], rounds))


def test_device_libm_against_the_hosts_measures_the_last_bit():
    """The f64 audit kernel follows the oracle path for path except where a transcendental function of the device's
    libm and the host's differ in the last bit (the sweeps' "libm class": a direction one ulp off flips a grazing hit
    many vertices later).  This measures that class instead of assuming it: sin / cos of 2 pi r (alias.rs:78-90), log
    (hittable.rs:342), atan2 / acos (get_sphere_uv, hittable.rs:530-540) on 2^20 arguments each — never more than
    2 ulp apart, and (for the record, printed) how many results differ at all."""
    import ctypes as C
    from raytracer_amd import _ffi as F
    rng = np.random.default_rng(11)
    n = 1 << 20
    r = rng.random(n)
    cases = [(0, "sin", 2.0 * np.pi * r, r, np.sin), (1, "cos", 2.0 * np.pi * r, r, np.cos),
             (2, "log", np.maximum(r, 1e-300), r, np.log), (4, "acos", 2.0 * r - 1.0, r, np.arccos)]
    y = rng.standard_normal(n)
    x = rng.standard_normal(n)
    cases.append((3, "atan2", y, x, None))
    report = []
    for op, name, a, b, fn in cases:
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.zeros(n, dtype=np.float64)
        F.check_hip(F.hip.rt_kat_libm(0, op, n, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                      out.ctypes.data_as(C.c_void_p)))
        want = fn(a) if fn else np.arctan2(a, b)
        ulp = np.abs(out.view(np.int64) - want.view(np.int64))
        ok = np.isfinite(want)
        report.append(f"{name}: {(ulp[ok] != 0).mean():.2e} differ, max {int(ulp[ok].max())} ulp")
        assert ulp[ok].max() <= 2, (name, int(ulp[ok].max()))
    print("device libm vs host libm, 2^20 arguments each: " + "; ".join(report))
