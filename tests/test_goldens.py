"""Committed golden vectors (tests/golden/oracle_goldens.npz, written by make_oracle_goldens.py from
the CPU oracle: SURVEY.md section 8c) against (a) the oracle as built now, (b) the product's host
BVH builder, and (c) on the GPU, the f64 audit kernels and the f32 production kernels."""
import os

import numpy as np
import pytest

import oracle_ffi as O
from conftest import SCENES
from golden.make_oracle_goldens import OFFSET, SEED, SPP, TIME, W, scene_args

LEAF = 0x80000000
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_goldens.npz"))


@pytest.mark.parametrize("name", list(SCENES))
def test_oracle_reproduces_goldens(name, earthmap):
    import raytracer_amd as R
    sid, asp, n, depth = scene_args(name)
    orc = O.OracleScene(sid, asp, n=n, texels=earthmap)
    h = R.image_height(W, asp)
    root, boxes, left, right = orc.bvh_preorder()
    assert root == int(G[f"{name}/root"])
    assert np.array_equal(left, G[f"{name}/node_left"]) and np.array_equal(right, G[f"{name}/node_right"])
    assert np.array_equal(boxes, G[f"{name}/node_box"])
    ids, t, nrm, uv, ff = orc.primary_hits(W, h, seed=3, time=TIME)
    assert np.array_equal(ids, G[f"{name}/primary_id"])
    hit = ids != O.NO_OBJECT
    np.testing.assert_allclose(t[hit], G[f"{name}/primary_t"][hit], rtol=1e-13)
    np.testing.assert_allclose(nrm[hit], G[f"{name}/primary_normal"][hit], atol=1e-13)
    img, cnt, _ = orc.render(W, h, SPP, depth, seed=SEED, sample_offset=OFFSET, threads=4)
    assert np.array_equal(np.array([cnt[k] for k in O.COUNTER_FIELDS], dtype=np.uint64), G[f"{name}/counters"])
    gold = G[f"{name}/image_sum"]
    fin = np.isfinite(gold)
    assert np.array_equal(fin, np.isfinite(img))
    np.testing.assert_allclose(img[fin], gold[fin], rtol=1e-12)


@pytest.mark.parametrize("name", list(SCENES))
def test_host_bvh_node_order_equals_golden(name):
    """Parity level 1: bit-exact flattened BVH node order (children as object ids) and f64 boxes."""
    import raytracer_amd as R
    sid, asp, n, _ = scene_args(name)
    hs = R.HostScene(sid, asp, n=n)
    box, left, right, root = hs.nodes()

    def conv(ref):
        ref = int(ref)
        return ref if ref == 0xFFFFFFFF or not (ref & LEAF) else LEAF | hs.leaf_object(ref)

    assert conv(root) == int(G[f"{name}/root"])
    assert np.array_equal(np.array([conv(x) for x in left], dtype=np.uint32), G[f"{name}/node_left"])
    assert np.array_equal(np.array([conv(x) for x in right], dtype=np.uint32), G[f"{name}/node_right"])
    assert np.array_equal(box, G[f"{name}/node_box"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENES))
def test_hip_primary_hits_equal_golden(name):
    """Levels 1-2 against the committed vectors: f64 audit ids bit-exact, t 1e-12; f32 production ids
    bit-exact too (audited walk), t within 1e-5 relative on every pixel."""
    import raytracer_amd as R
    sid, asp, n, _ = scene_args(name)
    ds = R.DeviceScene(R.HostScene(sid, asp, n=n), 0)
    h = R.image_height(W, asp)
    gid, gt, gn = G[f"{name}/primary_id"], G[f"{name}/primary_t"], G[f"{name}/primary_normal"]
    hit = gid != O.NO_OBJECT
    ids, t, nrm, uv, ff = ds.primary_hits(W, h, seed=3, time=TIME, f64=True)
    assert np.array_equal(ids, gid)
    np.testing.assert_allclose(t[hit], gt[hit], rtol=1e-12)
    np.testing.assert_allclose(nrm[hit], gn[hit], atol=1e-12)
    assert np.array_equal(ff[hit], G[f"{name}/primary_front"][hit])
    ids, t, nrm, uv, ff = ds.primary_hits(W, h, seed=3, time=TIME, f64=False)
    assert np.array_equal(ids, gid)  # the audited f32 walk (device/audit.hpp): exact ids, t within 1e-5 everywhere
    assert np.array_equal(ff[hit], G[f"{name}/primary_front"][hit])
    if hit.any():
        rel = np.abs(t[hit] - gt[hit]) / np.abs(gt[hit])
        assert rel.max() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENES))
def test_hip_f64_image_equals_golden(name):
    """Level 3 smoke: the f64 audit kernel consumes the same Philox words on the same branches as the
    oracle, so the low-spp radiance sums agree to rounding (a rare last-ulp branch flip aside)."""
    import raytracer_amd as R
    sid, asp, n, depth = scene_args(name)
    ds = R.DeviceScene(R.HostScene(sid, asp, n=n), 0)
    h = R.image_height(W, asp)
    # (RT_FLAG_FORCE_BVH: the reference-order walk, whose traversal counters are the oracle's; the default f64 walk of
    # leaf-list scenes is checked against it in test_gpu_parity.py)
    img, st = ds.render(W, h, SPP, depth, seed=SEED, sample_offset=OFFSET, stats=True, f64=True, force_bvh=True)
    gold = G[f"{name}/image_sum"]
    fin = np.isfinite(gold) & np.isfinite(img)
    assert (np.isfinite(gold) == np.isfinite(img)).mean() >= 0.999
    err = np.abs(img - gold)[fin] / (1.0 + np.abs(gold[fin]))
    assert (err > 1e-9).mean() <= 1e-3, f"{name}: {(err > 1e-9).mean()}"
    cnt = dict(zip(O.COUNTER_FIELDS, G[f"{name}/counters"].tolist()))
    for k in ("rays", "nodes_visited", "sphere_tests", "cube_tests", "closest_hits", "misses"):
        assert abs(st[k] - cnt[k]) <= 1e-4 * max(1, cnt[k]), (name, k, st[k], cnt[k])
