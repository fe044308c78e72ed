"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs (sizes the oracle finishes in seconds).

Three levels (BASELINE.json north_star):
  1. primary-ray hit primitive ids: bit-exact for the f64 audit kernels; the f32 production
     kernels may differ only on silhouette pixels (bounded fraction).
  2. primary-ray t and normals within 1e-5 relative (f32), 1e-12 (f64 audit).
  3. images: the f64 audit kernel follows the oracle path-for-path (same Philox words, same
     branches), so low-spp images agree to rounding; the f32 production kernel agrees within
     Monte-Carlo noise (PSNR / RMSE bounds written at each test).
"""
import numpy as np
import pytest

from conftest import SCENES

pytestmark = pytest.mark.gpu

W = 64


def _pair(name, earthmap):
    import oracle_ffi as O
    import raytracer_amd as R
    sid, asp, n, depth = SCENES[name]
    hs = R.HostScene(sid, asp, n=n)
    ds = R.DeviceScene(hs, 0)
    os_ = O.OracleScene(sid, asp, n=n, texels=earthmap)
    h = R.image_height(W, asp)
    return hs, ds, os_, h, depth


@pytest.mark.parametrize("name", list(SCENES))
def test_primary_hits_f64_exact(name, earthmap):
    hs, ds, os_, h, _ = _pair(name, earthmap)
    for time in (0.0, 0.625):
        gid, gt, gn, guv, gff = ds.primary_hits(W, h, seed=3, time=time, f64=True)
        oid, ot, on, ouv, off = os_.primary_hits(W, h, seed=3, time=time)
        assert np.array_equal(gid, oid), f"{name}: {(gid != oid).sum()} primary ids differ"
        hit = oid != 0xFFFFFFFF
        assert np.array_equal(gff[hit], off[hit])
        # same operations in the same order with contraction off: only libm (sin/cos/acos/atan2/log)
        # may differ in the last ulp
        np.testing.assert_allclose(gt[hit], ot[hit], rtol=1e-12, atol=0)
        np.testing.assert_allclose(gn[hit], on[hit], rtol=0, atol=1e-12)
        np.testing.assert_allclose(guv[hit], ouv[hit], rtol=0, atol=1e-12)


@pytest.mark.parametrize("w", [64, 150])
@pytest.mark.parametrize("name", list(SCENES))
def test_primary_hits_f32(name, w, earthmap):
    """north_star levels 1-2 for the f32 product walk (rt_primary_hits: the leaf arithmetic rt_render runs, audited by
    device/audit.hpp; pixels whose decisions sit inside their f32 rounding bounds are re-decided in f64):
    primitive ids and front_face EQUAL to the oracle's on every pixel, t within 1e-5 relative on every pixel.
    The audit must stay a filter, not a detour: it may hand at most 8 % of the pixels to the f64 kernel.
    Normals: 1e-5 is met wherever f32 STORAGE allows it — a sphere of radius r whose centre c is stored in f32 has
    its normal (p - c) / r known to 2^-24 |c| / r at best (bvh_stress: r = 0.05 at |c| = 40 gives 5e-5) — so the
    bound is 1e-5 on 95 % of the pixels and 3e-4 on all (bvh_stress_sparse, radii ten times smaller: 3e-3)."""
    hs, ds, os_, _, _ = _pair(name, earthmap)
    h = max(2, int(w / SCENES[name][1]))
    for fast in (False, True):
        gid, gt, gn, guv, gff = ds.primary_hits(w, h, seed=3, time=0.25, f64=False, fast=fast)
        redecided = ds.redecided
        oid, ot, on, ouv, off = os_.primary_hits(w, h, seed=3, time=0.25)
        assert np.array_equal(gid, oid), f"{name}: {(gid != oid).sum()} of {gid.size} primary ids differ"
        hit = oid != 0xFFFFFFFF
        assert np.array_equal(gff[hit], off[hit])
        rel_t = np.abs(gt[hit] - ot[hit]) / np.abs(ot[hit])
        dn = np.abs(gn[hit] - on[hit]).max(axis=-1)
        print(f"{name} w={w} fast={fast}: re-decided {redecided}/{gid.size}; t rel err max {rel_t.max() if rel_t.size else 0:.3g}; "
              f"normal abs err max {dn.max() if dn.size else 0:.3g}, p95 {np.quantile(dn, 0.95) if dn.size else 0:.3g}")
        # (an instance is not audited: every ray whose reference-order walk reaches it is re-decided in f64)
        bound = 0.5 if name == "final_scene_instanced" else 0.08
        assert redecided <= bound * gid.size, f"{name}: the audit sent {redecided} of {gid.size} pixels to f64"
        if rel_t.size:
            assert rel_t.max() <= 1e-5
            assert np.quantile(dn, 0.95) <= 1e-5
            assert dn.max() <= (3e-3 if name == "bvh_stress_sparse" else 3e-4)


@pytest.mark.parametrize("name", ["random_scene", "two_spheres", "earth", "simple_light", "cube_scene",
                                  "small_scene", "cornell_box", "final_scene", "perlin_spheres", "final_scene_perlin",
                                  "cornell_box_area", "area_light_plate", "random_scene_stdrng", "area_light_list",
                                  "final_scene_instanced"])
def test_render_f64_follows_oracle_path_for_path(name, earthmap):
    """Both f64 walks against the oracle (main.rs:65-105 + hittable.rs:208-415), same seed: the reference-order BVH
    walk (RT_FLAG_FORCE_BVH: every work counter identical, nodes and primitive tests included) and the walk
    rt_render_f64 takes by default — the leaf list on leaf-list scenes (no node tests, every leaf tested: only the
    traversal counters differ), the reference-order walk elsewhere."""
    hs, ds, os_, h, depth = _pair(name, earthmap)
    spp = 8
    o, ocnt, _ = os_.render(W, h, spp, depth, seed=11, sample_offset=5)
    leaf_list_scene = None
    for force_bvh in (True, False):
        g, gst = ds.render(W, h, spp, depth, seed=11, sample_offset=5, stats=True, f64=True, force_bvh=force_bvh)
        fin = np.isfinite(o) & np.isfinite(g)
        assert (np.isfinite(o) == np.isfinite(g)).mean() >= 0.999
        err = np.abs(g - o)[fin] / (1.0 + np.abs(o[fin]))
        # same draws, same branches; differences come from last-ulp libm results and the iterative vs
        # recursive product order — except for the rare path where such an ulp flips a branch
        assert np.quantile(err, 0.999) <= 1e-9, f"{name}: err q99.9 {np.quantile(err, 0.999)}"
        assert (err > 1e-9).mean() <= 1e-3
        # identical work counters (the f64 kernels do not early-exit)
        keys = ("rays", "closest_hits", "diffuse_vertices", "specular_vertices", "emitter_hits", "misses", "light_probes")
        if force_bvh:
            keys += ("nodes_visited", "sphere_tests", "cube_tests", "medium_probes")
        for k in keys:
            assert abs(gst[k] - ocnt[k]) <= 1e-4 * max(1, ocnt[k]), (name, force_bvh, k, gst[k], ocnt[k])
        if not force_bvh:
            leaf_list_scene = hs.desc.n_nodes > 0 and gst["nodes_visited"] == 0
    if name in ("cornell_box", "small_scene", "two_spheres"):
        assert leaf_list_scene, f"{name}: rt_render_f64 did not take the leaf list"


def _psnr(a, b):
    mse = np.mean((a - b) ** 2)
    return 10 * np.log10(1.0 / mse) if mse > 0 else np.inf


def _chunk_means(render_chunk, spp, chunks, clip):
    """Per-pixel mean and standard error of the mean from `chunks` independent renders of spp/chunks samples."""
    per = spp // chunks
    ms = []
    for c in range(chunks):
        m = render_chunk(c * per, per).astype(np.float64) / per
        m = np.where(np.isfinite(m), m, 0.0)
        ms.append(np.clip(m, 0, clip) if clip else m)
    ms = np.stack(ms)
    return ms.mean(axis=0), ms.std(axis=0, ddof=1) / np.sqrt(chunks)


@pytest.mark.parametrize("name,wavefront", [(n, False) for n in (
    "random_scene", "simple_light", "cornell_box", "earth", "small_scene", "final_scene", "bvh_stress", "perlin_spheres",
    "cornell_box_area", "area_light_plate", "area_light_list", "final_scene_instanced")] + [
    # the wavefront-queue form (RT_FLAG_WAVEFRONT) against the ORACLE (main.rs:65-105), not only against
    # its sibling kernel (tests/test_gpu_wavefront.py)
    ("random_scene", True), ("final_scene", True), ("bvh_stress", True)])
def test_render_f32_within_monte_carlo_noise(name, wavefront, earthmap):
    """f32 production kernel vs oracle, different seeds, matched spp (main.rs:160-167's estimator): the
    difference of two independent estimates must look like noise of the MEASURED per-pixel variance —
    per pixel channel, z = |gpu - oracle| / sqrt(se_gpu^2 + se_oracle^2) with the standard errors taken from
    the spread of 16 chunk means: >= 98.5 % inside 3 sigma (a t-distribution with 15 degrees of freedom has
    99.1 % there), no pixel beyond 8 sigma except in the heavy-tailed scenes, and the image means within
    4 standard errors of the difference of the means."""
    hs, ds, os_, h, depth = _pair(name, earthmap)
    spp, chunks = 512, 16
    # final_scene / bvh_stress: the reference's estimator has fireflies of 1e20 and more (fog gains energy per
    # scatter; glass next to the sky): compare chunk means clipped to the displayable range
    clip = 4.0 if name in ("final_scene", "bvh_stress", "final_scene_instanced") else 0.0
    launches = []

    def gpu_chunk(off, n):
        img, st = ds.render(W, h, n, depth, seed=21, sample_offset=off, wavefront=wavefront, stats=wavefront)
        launches.append(st["launches"])
        return img
    g, sg = _chunk_means(gpu_chunk, spp, chunks, clip)
    if wavefront:
        assert min(launches) >= 3, "RT_FLAG_WAVEFRONT did not take the wavefront form (1 + 2 launches per bounce)"
    o, so = _chunk_means(lambda off, n: os_.render(W, h, n, depth, seed=22, sample_offset=off)[0], spp, chunks, clip)
    sig = np.sqrt(sg * sg + so * so)
    # pixels without real spread in either estimate (background, saturated emitters: every sample the same number
    # up to rounding) must agree outright; the z-score is for the others
    lit = sig > 1e-6 * (np.abs(g) + np.abs(o)) + 1e-9
    z = np.abs(g - o)[lit] / sig[lit]
    flat = ~lit
    assert np.allclose(g[flat], o[flat], rtol=1e-4, atol=1e-6), f"{name}: flat pixels differ by {np.abs(g - o)[flat].max()}"
    mean_g, mean_o = g.mean(), o.mean()
    se_mean = np.sqrt((sg ** 2).sum() + (so ** 2).sum()) / g.size
    print(f"{name}: within 3 sigma {np.mean(z <= 3):.4f}, max z {z.max():.2f}, mean ratio {mean_g / mean_o:.5f} "
          f"(+- {se_mean / mean_o:.5f})")
    assert np.mean(z <= 3.0) >= 0.985, f"{name}: only {np.mean(z <= 3):.4f} of pixel channels inside 3 sigma"
    if not clip:
        assert z.max() <= 8.0, f"{name}: a pixel channel is {z.max():.1f} sigma off"
    assert abs(mean_g - mean_o) <= 4.0 * se_mean, f"{name}: means {mean_g} vs {mean_o}, se {se_mean}"


@pytest.mark.parametrize("name", ["cornell_box", "small_scene", "two_spheres", "cube_scene"])
def test_leaf_list_path_equals_bvh_walk(name, earthmap):
    """Small scenes take the leaf-list traversal in f32; RT_FLAG_FORCE_BVH walks the BVH instead.
    Same leaves in the same order with the same bounds and the same draws.  The two kernels are
    separate instantiations of the same source, built with -ffp-contract=on (fused per source expression,
    not per inlined copy): they compute the same bits and follow the same paths."""
    hs, ds, os_, h, depth = _pair(name, earthmap)
    a, sa = ds.render(W, h, 32, depth, seed=5, stats=True)
    b, sb = ds.render(W, h, 32, depth, seed=5, stats=True, force_bvh=True)
    for k in ("rays", "closest_hits", "diffuse_vertices", "specular_vertices", "emitter_hits", "misses"):
        assert abs(sa[k] - sb[k]) <= 0.005 * max(sa[k], 1) + 2, (k, sa[k], sb[k])
    if hs.desc.n_nodes:
        assert sa["nodes_visited"] == 0 and sb["nodes_visited"] > 0
    same = np.isclose(a, b, rtol=1e-4, atol=1e-5).all(axis=-1)
    # the helpers both kernels share (dot products, rigid maps, sphere core, medium exit) use one explicit
    # fused form, so the two walks compute the same bits and follow the same paths: observed 100 %
    assert same.mean() >= 0.999, f"{name}: only {same.mean():.4f} of pixels agree"
    assert abs(a.sum() / b.sum() - 1.0) <= 1e-4


def test_leaf_list_path_refused_when_boxes_clip(earthmap):
    """simple_light is built with the reference's Rotate::bounding_box (Q3), whose box does not bound
    the rotated cube: BVH pruning is then part of the image, so the library must walk the BVH."""
    hs, ds, os_, h, depth = _pair("simple_light", earthmap)
    a, sa = ds.render(W, h, 4, depth, seed=5, stats=True)
    assert sa["nodes_visited"] > 0


@pytest.mark.parametrize("name", ["random_scene", "final_scene", "bvh_stress", "earth", "final_scene_instanced"])
def test_ordered_walk_equals_reference_walk(name, earthmap):
    """Scenes too large for the leaf list take the near-child-first walk in f32 (verified-conservative
    boxes only); RT_FLAG_FORCE_BVH takes the reference's left-first walk (hittable.rs:400-412).  Both run
    the same leaf tests in the same f32 kernels; the winner is the smallest t with ties to the smaller
    DFS rank, and medium draws are keyed by object, so the images agree except where a cull decision
    hangs on the last ulp.  The ordered walk must also visit fewer nodes."""
    hs, ds, os_, h, depth = _pair(name, earthmap)
    a, sa = ds.render(W, h, 16, depth, seed=5, stats=True)
    b, sb = ds.render(W, h, 16, depth, seed=5, stats=True, force_bvh=True)
    for k in ("rays", "closest_hits", "diffuse_vertices", "specular_vertices", "emitter_hits", "misses"):
        assert abs(sa[k] - sb[k]) <= 0.002 * max(sa[k], 1) + 2, (k, sa[k], sb[k])
    print(f"{name}: nodes/ray ordered {sa['nodes_visited'] / sa['rays']:.1f}, reference {sb['nodes_visited'] / sb['rays']:.1f}")
    if name == "random_scene":  # dense enough for near-first culling to pay at this size
        assert sa["nodes_visited"] < sb["nodes_visited"]
    # The two walks are separate kernels, but every helper they share is written in one explicit fused form
    # (vec.hpp mul_add3 and friends), so they compute the same bits, take the same paths and trace the same
    # number of rays; only the f32 order in which a pixel's samples are summed differs.
    same = np.isclose(a, b, rtol=1e-4, atol=1e-5).all(axis=-1)
    assert same.mean() >= 0.999, f"{name}: only {same.mean():.4f} of pixels agree"
    assert abs(np.clip(a, 0, 16).sum() / np.clip(b, 0, 16).sum() - 1.0) <= 1e-4
    assert sa["rays"] == sb["rays"]


def test_render_multi_equals_single_render(earthmap):
    """rt_render_multi with the one GPU listed three times: three host threads, three scene copies,
    three sample shards (sizes 10/11/11 of 32: uneven on purpose), summed on devices[0].  The union of
    the shards' Philox streams is the single render's, so only the f32 summation order differs."""
    import raytracer_amd as R
    hs = R.HostScene(6, 1.0)
    ds = R.DeviceScene(hs, 0)
    one, s1 = ds.render(W, W, 32, 50, seed=9, sample_offset=3, stats=True)
    multi, sm = R.render_multi(hs, [0, 0, 0], W, W, 32, 50, seed=9, sample_offset=3, stats=True)
    assert sum(s["rays"] for s in sm) == s1["rays"]
    np.testing.assert_allclose(multi, one, rtol=2e-5, atol=1e-5)
    solo, _ = R.render_multi(hs, [0], W, W, 32, 50, seed=9, sample_offset=3)
    np.testing.assert_allclose(solo, one, rtol=2e-5, atol=1e-5)
    with pytest.raises(RuntimeError):
        R.render_multi(hs, [99], W, W, 4, 5)


def test_multi_scene_is_reused_across_renders():
    """rt_multi_create / rt_multi_render / rt_multi_destroy: the per-device scene copies, accumulators and the
    staging area outlive a render.  Four entries (the one GPU four times) so that three peer copies land in three
    staging slots before the one add; renders of growing and shrinking size, with and without the per-shard
    counters (without them the whole call is asynchronous up to its final wait), a sample count smaller than the
    number of shards (empty shards), and shard-wise equality with plain renders of rt_multi_shard's ranges."""
    import raytracer_amd as R
    hs = R.HostScene(6, 1.0)
    ds = R.DeviceScene(hs, 0)
    ms = R.MultiScene(hs, [0, 0, 0, 0])
    for (w, spp, per_device) in ((32, 9, True), (96, 30, False), (48, 3, True), (96, 30, True)):
        one, s1 = ds.render(w, w, spp, 50, seed=5, sample_offset=7, stats=True)
        multi, sm = ms.render(w, w, spp, 50, seed=5, sample_offset=7, stats=True, per_device=per_device)
        np.testing.assert_allclose(multi, one, rtol=2e-5, atol=1e-5)
        if per_device:
            assert sum(s["rays"] for s in sm) == s1["rays"]
            for g, s in enumerate(sm):
                first, count = R.multi_shard(spp, g, 4)
                _, sg = ds.render(w, w, count, 50, seed=5, sample_offset=7 + first, stats=True)
                assert s["rays"] == sg["rays"]
    # the sum of the shards' own renders (a shard's pixel sums come from float atomics of several sample units, so
    # equality is to the last bits, not bitwise)
    parts = [ds.render(64, 64, R.multi_shard(10, g, 4)[1], 50, seed=2, sample_offset=R.multi_shard(10, g, 4)[0])[0]
             for g in range(4)]
    want = ((parts[0] + parts[1]) + parts[2]) + parts[3]
    got, _ = ms.render(64, 64, 10, 50, seed=2)
    np.testing.assert_allclose(got, want, rtol=3e-6, atol=1e-6)
    ms.close()


def test_multi_scene_in_the_references_arithmetic():
    """rt_multi_render_f64: the same sharding with every shard on the f64 kernels, summed in f64 in shard order.  The
    shards' paths are the single render's path for path (equal work counters), the image equal to the order of the sums;
    f32 and f64 renders may alternate on one handle (the accumulators are sized in bytes)."""
    import raytracer_amd as R
    hs = R.HostScene(6, 1.0)
    ds = R.DeviceScene(hs, 0)
    ms = R.MultiScene(hs, [0, 0, 0])
    for (w, spp) in ((40, 7), (72, 20)):
        one, s1 = ds.render(w, w, spp, 50, seed=5, sample_offset=2, stats=True, f64=True)
        multi, sm = ms.render(w, w, spp, 50, seed=5, sample_offset=2, stats=True, f64=True)
        assert multi.dtype == np.float64
        for k in ("rays", "closest_hits", "misses", "diffuse_vertices"):
            assert sum(s[k] for s in sm) == s1[k], k
        assert np.array_equal(np.isnan(multi), np.isnan(one))
        ok = np.isfinite(one)
        np.testing.assert_allclose(multi[ok], one[ok], rtol=1e-12, atol=1e-12)
        f32, _ = ms.render(w, w, spp, 50, seed=5, sample_offset=2)  # the same handle, the f32 kernels
        one32, _ = ds.render(w, w, spp, 50, seed=5, sample_offset=2)
        np.testing.assert_allclose(f32, one32, rtol=2e-5, atol=1e-5)
    ms.close()


@pytest.mark.parametrize("name", list(SCENES))
def test_f32_tracks_f64_audit_at_equal_seed(name, earthmap):
    """Same seed, same Philox words: the f32 production kernels and the f64 audit kernel follow the same
    paths except where a branch hangs on rounding, so depth-limited radiance totals agree closely — a
    systematic f32 defect (this test's origin: fog that never scattered on rays deep inside a large
    medium) shows up as a percent-level gap.  Depth-limited because the reference's fog estimator gains
    energy per scatter and its full-depth totals are dominated by 1e20-sized samples."""
    hs, ds, os_, h, depth = _pair(name, earthmap)
    for d in (2, 3, min(depth, 6)):
        a, sa = ds.render(W, h, 64, d, seed=21, stats=True)
        b, sb = ds.render(W, h, 64, d, seed=21, stats=True, f64=True)
        ta, tb = float(a[np.isfinite(a)].sum()), float(b[np.isfinite(b)].sum())
        assert abs(ta / tb - 1.0) <= 2e-3, (name, d, ta, tb)
        # (work counters are not compared: the f32 kernel stops a path at zero throughput, the audit kernel
        # walks it to the end like the reference)


def test_sorted_kernel_experiment_renders_the_same_paths(earthmap):
    """RT_LINEAR_SORTED=1 (read once per process, so a child process) selects the LDS-sorted variant of
    the leaf-list kernel: same (pixel, sample) Philox streams, different summation order."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import raytracer_amd as R\n"
        "hs = R.HostScene(6, 1.0); ds = R.DeviceScene(hs, 0)\n"
        "a, st = ds.render(72, 72, 48, 50, seed=5, sample_offset=7, stats=True)\n"
        "np.save(sys.argv[1], a); print(st['rays'])\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = {}
    for tag, env in (("default", {}), ("sorted", {"RT_LINEAR_SORTED": "1"})):
        path = f"/tmp/rt_sorted_{tag}_{os.getpid()}.npy"
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code, path], env=e, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = (np.load(path), int(r.stdout.strip().splitlines()[-1]))
        os.remove(path)
    (a, ra), (b, rb) = out["default"], out["sorted"]
    assert abs(ra - rb) <= 0.002 * ra  # a path may differ where a branch hangs on the last ulp
    same = np.isclose(a, b, rtol=1e-4, atol=1e-5).all(axis=-1)
    assert same.mean() >= 0.97, same.mean()
    assert abs(a.sum() / b.sum() - 1.0) <= 0.005


@pytest.mark.parametrize("name", list(SCENES))
def test_production_walk_picks_the_reference_walks_primary_hit(name, earthmap):
    """rt_primary_hits_fast answers the primary-ray query with the walk rt_render uses (leaf list, or the
    ordered walk of the library's own SAH tree), rt_primary_hits with the reference-order walk of the
    reference's tree.  Same f32 leaf tests, so the winner (object id), t, normal and front_face must be
    IDENTICAL — the claim of DESIGN.md section 3.2 ("why the winner is the same"), including exact ties."""
    hs, ds, os_, h, _ = _pair(name, earthmap)
    for time in (0.0, 0.625):
        for w in (W, 97):  # 97: ragged tiles
            hh = max(2, int(w / SCENES[name][1]))
            a = ds.primary_hits(w, hh, seed=3, time=time, fast=True)
            b = ds.primary_hits(w, hh, seed=3, time=time)
            assert np.array_equal(a[0], b[0]), f"{name}: {(a[0] != b[0]).sum()} ids differ"
            hit = b[0] != 0xFFFFFFFF
            assert np.array_equal(a[1][hit], b[1][hit])
            assert np.array_equal(a[2][hit], b[2][hit]) and np.array_equal(a[4][hit], b[4][hit])


def test_production_walk_primary_hits_at_one_million_spheres():
    """The same identity at BASELINE config 4's scene size (10^6 spheres, SAH tree of 999 999 nodes,
    1280x720 primary rays): every pixel's winner, t and normal equal the reference-order walk's."""
    import raytracer_amd as R
    hs = R.HostScene(7, 16.0 / 9.0, n=1000000)
    ds = R.DeviceScene(hs, 0)
    a = ds.primary_hits(1280, 720, seed=3, time=0.25, fast=True)
    b = ds.primary_hits(1280, 720, seed=3, time=0.25)
    assert np.array_equal(a[0], b[0]), f"{(a[0] != b[0]).sum()} ids differ"
    hit = b[0] != 0xFFFFFFFF
    assert hit.mean() > 0.5
    assert np.array_equal(a[1][hit], b[1][hit]) and np.array_equal(a[2][hit], b[2][hit])


def test_an_instance_renders_what_its_flattened_scene_renders(earthmap):
    """Scene 16 holds final_scene's 1000 small spheres as ONE Translate(Rotate(Bvh)) instance (hittable.rs:81-91 over a
    sub-tree), scene 8 wraps each sphere by itself: the same primitives in the same places.  hit() returns the smallest
    candidate t whatever the tree, draws are keyed, so both scenes trace the same paths: in f64 (the
    instance arm of the reference-order walk against 1000 wrapper chains) and in f32 (both de-instanced)."""
    import raytracer_amd as R
    h = W
    inst = R.DeviceScene(R.HostScene("final_scene_instanced", 1.0), 0)
    flat = R.DeviceScene(R.HostScene("final_scene", 1.0), 0)
    for kw, tol in (({"f64": True}, 1e-9), ({"f64": True, "force_bvh": True}, 1e-9), ({}, 1e-4), ({"force_bvh": True}, 1e-4)):
        a, sa = inst.render(W, h, 16, 50, seed=5, stats=True, **kw)
        b, sb = flat.render(W, h, 16, 50, seed=5, stats=True, **kw)
        for k in ("rays", "closest_hits", "diffuse_vertices", "specular_vertices", "emitter_hits", "misses"):
            assert abs(sa[k] - sb[k]) <= 2e-3 * sb[k] + 2, (kw, k, sa[k], sb[k])
        fin = np.isfinite(a).all(axis=-1) & np.isfinite(b).all(axis=-1)
        same = np.isclose(a[fin], b[fin], rtol=tol, atol=tol).all(axis=-1)
        assert same.mean() >= 0.995, (kw, same.mean())
    # the instance arm really ran: the reference-order walk counts one Rotate / Translate visit per ENTRY of the
    # instance, the flattened scene one per wrapped sphere it tests
    _, si = inst.render(W, h, 4, 50, seed=5, stats=True, f64=True, force_bvh=True)
    _, sf = flat.render(W, h, 4, 50, seed=5, stats=True, f64=True, force_bvh=True)
    assert 0 < si["rotate_visits"] < sf["rotate_visits"] / 4
