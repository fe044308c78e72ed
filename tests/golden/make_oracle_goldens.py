"""Writes tests/golden/oracle_goldens.npz: what the CPU oracle (oracle/, the C++17 f64 restatement
of the reference — "parity unpinned" beyond the reference's 11 KATs, see oracle/README.md) computes
for every scene at 64 px wide with fixed seeds (SURVEY.md section 8c "golden data the build commits"):

  (i)  the BVH in pre-order (child references with leaves as object ids, f64 boxes) and per-pixel
       primary-ray object id / t / normal / front_face (parity levels 1-2);
  (ii) the radiance SUM image at SPP samples per pixel (level 3 smoke).

These are DATA produced by this repository's own oracle (the reference is Rust and cannot run
here); they freeze the oracle, so an unintended change to it — or to the draw-stream conventions
the HIP kernels share with it — fails tests/test_goldens.py on the CPU and on the GPU.
Add the vectors of scenes that have none yet with:  python tests/golden/make_oracle_goldens.py
(the scenes already in the file are kept byte for byte); regenerate everything deliberately with --all.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_ffi as O  # noqa: E402
from conftest import SCENES  # noqa: E402

W, SPP, SEED, OFFSET, TIME = 64, 16, 11, 5, 0.625
BVH_STRESS_N = 2000  # keep the fixture small


def scene_args(name):
    sid, asp, n, depth = SCENES[name]
    return sid, asp, (min(n, BVH_STRESS_N) if n else 0), depth


def main():
    import raytracer_amd as R
    earth = R.load_earthmap()
    path = os.path.join(HERE, "oracle_goldens.npz")
    out = {}
    if os.path.exists(path) and "--all" not in sys.argv:  # additive by default: scenes already frozen stay as they are
        with np.load(path) as old:
            out = {k: old[k] for k in old.files}
    for name in SCENES:
        if f"{name}/root" in out:
            continue
        sid, asp, n, depth = scene_args(name)
        orc = O.OracleScene(sid, asp, n=n, texels=earth)
        h = R.image_height(W, asp)
        root, boxes, left, right = orc.bvh_preorder()
        ids, t, nrm, uv, ff = orc.primary_hits(W, h, seed=3, time=TIME)
        img, cnt, _ = orc.render(W, h, SPP, depth, seed=SEED, sample_offset=OFFSET, threads=8)
        out[f"{name}/root"] = np.uint32(root)
        out[f"{name}/node_box"] = boxes
        out[f"{name}/node_left"] = left
        out[f"{name}/node_right"] = right
        out[f"{name}/primary_id"] = ids
        out[f"{name}/primary_t"] = t
        out[f"{name}/primary_normal"] = nrm
        out[f"{name}/primary_front"] = ff
        out[f"{name}/image_sum"] = img
        out[f"{name}/counters"] = np.array([cnt[k] for k in O.COUNTER_FIELDS], dtype=np.uint64)
        print(name, "nodes", len(left), "rays", cnt["rays"])
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
