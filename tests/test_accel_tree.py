"""The fast path's own binned-SAH tree (raytracer_amd/csrc/accel_build.hpp) checked on the CPU through
rt_debug_build_tree: structure (every leaf exactly once, pre-order, children after parents), boxes
(each child box bounds everything below it), the depth cap, and degenerate inputs."""
import ctypes as C

import numpy as np
import pytest

LEAF = 0x80000000


def _build(boxes, max_depth=64):
    from raytracer_amd import _ffi as F
    n = len(boxes)
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    refs = (LEAF | np.arange(n, dtype=np.uint32)).astype(np.uint32)
    nodes = np.zeros((n - 1, 16), dtype=np.uint32)
    h = C.c_uint32()
    F.check_hip(F.hip.rt_debug_build_tree(n, boxes.ctypes.data, refs.ctypes.data, max_depth, nodes.ctypes.data, C.byref(h)))
    return nodes, int(h.value)


def _check(boxes, nodes):
    n = len(boxes)
    fb = nodes.view(np.float32)
    seen = np.zeros(n, dtype=np.int64)

    def walk(ref, depth):
        """returns (bounds lo, hi, height) of the subtree"""
        if ref & LEAF:
            i = int(ref & 0x0FFFFFFF)
            seen[i] += 1
            return boxes[i, :3], boxes[i, 3:], 0
        assert 0 <= ref < n - 1
        lo = np.full(3, np.inf)
        hi = np.full(3, -np.inf)
        height = 0
        for side in (0, 1):
            child = int(nodes[ref, 12 + side])
            if not child & LEAF:
                assert child > ref, "children come after parents (pre-order)"
            clo, chi, ch = walk(child, depth + 1)
            blo, bhi = fb[ref, 6 * side:6 * side + 3], fb[ref, 6 * side + 3:6 * side + 6]
            assert (blo <= clo).all() and (chi <= bhi).all(), "a child box must bound its subtree"
            lo, hi, height = np.minimum(lo, blo), np.maximum(hi, bhi), max(height, ch)
        return lo, hi, height + 1

    import sys
    sys.setrecursionlimit(10000)
    _, _, height = walk(0, 0)
    assert (seen == 1).all(), "every leaf exactly once"
    return height


def _random_boxes(rng, n, giant=True):
    c = rng.uniform([-40, 0.1, -40], [40, 20, 40], size=(n, 3))
    r = rng.uniform(0.05, 0.15, size=(n, 1))
    b = np.hstack([c - r, c + r]).astype(np.float32)
    if giant:  # the r = 1000 ground sphere of random_scene / bvh_stress
        b[0] = [-1000, -2000, -1000, 1000, 0, 1000]
    return b


@pytest.mark.parametrize("n", [2, 3, 7, 64, 1000, 20000])
def test_tree_is_valid_and_bounded(n):
    rng = np.random.default_rng(n)
    boxes = _random_boxes(rng, n)
    nodes, h = _build(boxes)
    assert _check(boxes, nodes) == h
    assert h <= max(1, int(np.ceil(np.log2(n)))) + 6 or h <= 64


def test_depth_cap_forces_balanced_splits():
    rng = np.random.default_rng(1)
    # exponentially spaced boxes: SAH alone would peel one leaf per level
    x = 2.0 ** np.arange(200, dtype=np.float64) % 1e30
    boxes = np.stack([x, x * 0, x * 0, x * 1.01 + 1e-3, x * 0 + 1, x * 0 + 1], axis=1).astype(np.float32)
    cap = int(np.ceil(np.log2(len(boxes)))) + 2
    nodes, h = _build(boxes, max_depth=cap)
    assert _check(boxes, nodes) == h
    assert h <= cap
    nodes2, h2 = _build(boxes, max_depth=3)  # below ceil(log2 n): raised to it
    assert h2 <= int(np.ceil(np.log2(len(boxes))))


def test_coincident_boxes():
    boxes = np.tile(np.array([[0, 0, 0, 1, 1, 1]], dtype=np.float32), (33, 1))
    nodes, h = _build(boxes)
    assert _check(boxes, nodes) == h
    assert h <= 6 + 6


def test_rejects_bad_arguments():
    from raytracer_amd import _ffi as F
    h = C.c_uint32()
    one = np.zeros((1, 6), dtype=np.float32)
    refs = np.array([LEAF], dtype=np.uint32)
    out = np.zeros((1, 16), dtype=np.uint32)
    assert F.hip.rt_debug_build_tree(1, one.ctypes.data, refs.ctypes.data, 8, out.ctypes.data, C.byref(h)) != 0
    two = np.zeros((2, 6), dtype=np.float32)
    bad = np.array([LEAF, 5], dtype=np.uint32)
    assert F.hip.rt_debug_build_tree(2, two.ctypes.data, bad.ctypes.data, 8, out.ctypes.data, C.byref(h)) != 0
