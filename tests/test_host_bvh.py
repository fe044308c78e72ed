"""Parity level 1 on the CPU: the product's flat pre-order BVH (raytracer_amd/csrc/host, built
straight into arrays) against the oracle's pointer tree (oracle/, recursive like hittable.rs:159-206)
dumped in pre-order — two independent implementations, same axis stream: node order, child
references and f64 boxes must be bit-identical; so must Camera::new."""
import numpy as np
import pytest

import oracle_ffi as O
from conftest import SCENES

LEAF = 0x80000000


def _compare(hs, orc):
    box, left, right, root = hs.nodes()
    oroot, obox, oleft, oright = orc.bvh_preorder()

    def conv(ref):
        ref = int(ref)
        if ref == 0xFFFFFFFF or not (ref & LEAF):
            return ref
        return LEAF | hs.leaf_object(ref)

    assert conv(root) == oroot
    assert len(left) == orc.info.n_nodes
    assert np.array_equal(np.array([conv(x) for x in left], dtype=np.uint32), oleft)
    assert np.array_equal(np.array([conv(x) for x in right], dtype=np.uint32), oright)
    assert np.array_equal(box, obox)  # bit-exact f64
    assert np.array_equal(hs.camera_array(), orc.camera_array())
    assert hs.n_objects == orc.info.n_objects
    assert hs.desc.n_lights == orc.info.n_lights


@pytest.mark.parametrize("name", list(SCENES))
def test_flat_bvh_equals_oracle_tree(name, earthmap):
    import raytracer_amd as R
    sid, asp, n, _ = SCENES[name]
    n = min(n, 5000)
    for bvh_seed in (7, 12345):
        hs = R.HostScene(sid, asp, n=n, bvh_seed=bvh_seed)
        orc = O.OracleScene(sid, asp, n=n, bvh_seed=bvh_seed, texels=earthmap)
        _compare(hs, orc)


@pytest.mark.parametrize("mode", [0, 1])
def test_rotate_bbox_modes(mode):
    """Q3: `reference` rotates two corners by q (may clip the object); `correct` bounds it."""
    import raytracer_amd as R
    for sid in (3, 6):
        hs = R.HostScene(sid, 1.0, rotate_bbox=mode)
        orc = O.OracleScene(sid, 1.0, rotate_bbox=mode)
        _compare(hs, orc)


def test_bvh_shape_rules():
    """hittable.rs:170-188: [] -> Empty, [x] -> bare, [l, r] -> one node, else split at len/2; pre-order."""
    import raytracer_amd as R
    one = R.HostScene(2, 1.5)       # earth: 1 object
    assert one.desc.n_nodes == 0 and one.desc.root & LEAF and one.bvh_calls == 0
    two = R.HostScene(1, 1.5)       # two_spheres
    assert two.desc.n_nodes == 1 and two.desc.root == 0 and two.bvh_calls == 1
    big = R.HostScene(7, 1.5, n=999)  # 1000 objects -> 999 nodes, height ceil(log2 1000) = 10
    box, left, right, root = big.nodes()
    assert len(left) == 999 and big.bvh_calls == 999 and big.desc.bvh_height == 10
    # pre-order: a node child always has a larger index; the left child node (if any) is index + 1
    for i in range(len(left)):
        if not left[i] & LEAF:
            assert left[i] == i + 1
        if not right[i] & LEAF:
            assert right[i] > i
    # every world object appears exactly once as a leaf
    leaves = [big.leaf_object(int(r)) for r in list(left) + list(right) if r & LEAF]
    assert sorted(leaves) == list(range(1000))
    # parent boxes bound their children's boxes
    for i in range(len(left)):
        for c in (left[i], right[i]):
            if not c & LEAF:
                assert (box[i, :3] <= box[c, :3]).all() and (box[c, 3:] <= box[i, 3:]).all()


def test_image_height_truncates():
    import raytracer_amd as R
    assert R.image_height(400, 16.0 / 9.0) == 225
    assert R.image_height(400, 1.7778) == 224       # why 16/9 must be passed exactly
    assert R.image_height(1200, 1.5) == 800
