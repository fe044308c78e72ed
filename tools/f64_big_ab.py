"""f64 on the BVH-bound scenes: the ordered walk (default) against the reference-order walk (RT_FLAG_FORCE_BVH) —
same image required (bit for bit up to the order of the per-pixel atomic sums), kernel_ms of each.
usage: f64_big_ab.py [scene ...]   (default: C1 C3 C4 as in bench.py's f64 sublines)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raytracer_amd as R

CASES = {
    "C1": ("random_scene", 400, 16.0 / 9.0, 0, 100),
    "C3": ("final_scene", 1024, 1.0, 0, 16),
    "C3i": ("final_scene_instanced", 512, 1.0, 0, 8),
    "C4": ("bvh_stress", 640, 16.0 / 9.0, 1000000, 1),
    "two": ("two_spheres", 400, 16.0 / 9.0, 0, 32),
    "earth": ("earth", 400, 16.0 / 9.0, 0, 32),
    "light": ("simple_light", 400, 1.5, 0, 64),
    "mixed": ("mixed_scene", 256, 1.0, 0, 32),
    "cube": ("cube_scene", 256, 1.0, 0, 32),
    "rstd": ("random_scene_stdrng", 400, 16.0 / 9.0, 0, 16),
}


def main():
    names = sys.argv[1:] or ["C1", "C3", "C4"]
    earth = R.load_earthmap()
    for name in names:
        scene, w, asp, n, spp = CASES[name]
        h = R.image_height(w, asp)
        hs = R.HostScene(scene, asp, n=n, texels=earth)
        ds = R.DeviceScene(hs, 0)
        out = {}
        if os.environ.get("ONLY_ORDERED"):  # tuning runs: time the default walk only
            acc = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
            ds.render_device(acc.data_ptr(), w, h, 1, 50, f64=True)
            ms = min(ds.render_device(acc.data_ptr(), w, h, spp, 50, seed=5, f64=True)["kernel_ms"] for _ in range(3))
            print(f"{name} {ms:.2f} ms", flush=True)
            continue
        for label, force in (("ordered", False), ("reference", True)):
            acc = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
            ds.render_device(acc.data_ptr(), w, h, 1, 50, f64=True, force_bvh=force)
            acc.zero_()
            st = ds.render_device(acc.data_ptr(), w, h, spp, 50, seed=5, f64=True, force_bvh=force, stats=True)
            acc2 = torch.zeros_like(acc)
            ms = ds.render_device(acc2.data_ptr(), w, h, spp, 50, seed=5, f64=True, force_bvh=force)["kernel_ms"]
            out[label] = (acc.cpu().numpy(), st, ms)
        a, sa, ma = out["ordered"]
        b, sb, mb = out["reference"]
        err = np.abs(a - b) / (1 + np.abs(b))
        same = {k: sa[k] == sb[k] for k in ("rays", "closest_hits", "misses", "diffuse_vertices", "specular_vertices", "emitter_hits", "light_probes")}
        print(f"{name} {scene} {w}x{h}x{spp}: ordered {ma:.2f} ms  reference {mb:.2f} ms  x{mb / ma:.2f}  "
              f"max rel err {err.max():.2e}  nodes {sa['nodes_visited']} vs {sb['nodes_visited']}  outcome counters equal: "
              f"{all(same.values())}", flush=True)


if __name__ == "__main__":
    main()
