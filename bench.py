#!/usr/bin/env python3
"""Headline benchmark: Mrays/s, Cornell box 1024x1024 at 4096 spp, depth 50 (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one full pass of the hot path over the workload: every rank renders `spp` samples of every
pixel of the Cornell box into an HBM-resident accumulator (rt_render_device), then (N > 1) one RCCL
sum all-reduce merges the accumulators, then the image is tonemapped on the device.  Weak scaling:
per-GPU work is fixed (4096 spp per GPU), the merged image has N x 4096 spp.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# frozen cost table: oracle/cost_model.h (tests/test_cost_model.py keeps the two in sync)
COST = {
    "bytes_per_node": 32, "bytes_per_prim_test": 16, "bytes_per_closest_hit": 32, "bytes_per_pixel_write": 12,
    "flops_per_node": 18, "flops_per_sphere_test": 25, "flops_per_sphere_hit": 17, "flops_per_cube_test": 18,
    "flops_per_cube_hit": 30, "flops_per_rotate_visit": 112, "flops_per_translate_visit": 6,
    "flops_per_diffuse_vertex": 60, "flops_per_specular_vertex": 40,
}
PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md: vector FP32 peak (= FP32 MFMA peak)
PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec


def algorithmic_per_ray(st, n_pixels, launches):
    """ALGORITHMIC flops and bytes per ray from the work counters."""
    rays = max(1, st["rays"])
    flops = (COST["flops_per_node"] * st["nodes_visited"] + COST["flops_per_sphere_test"] * st["sphere_tests"]
             + COST["flops_per_sphere_hit"] * st["sphere_hits"] + COST["flops_per_cube_test"] * st["cube_tests"]
             + COST["flops_per_cube_hit"] * st["cube_hits"] + COST["flops_per_rotate_visit"] * st["rotate_visits"]
             + COST["flops_per_translate_visit"] * st["translate_visits"]
             + COST["flops_per_diffuse_vertex"] * st["diffuse_vertices"]
             + COST["flops_per_specular_vertex"] * st["specular_vertices"])
    byts = (COST["bytes_per_node"] * st["nodes_visited"]
            + COST["bytes_per_prim_test"] * (st["sphere_tests"] + st["cube_tests"])
            + COST["bytes_per_closest_hit"] * st["closest_hits"]
            + COST["bytes_per_pixel_write"] * n_pixels * launches)
    return flops / rays, byts / rays


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=4096, help="samples per pixel PER GPU per step")
    ap.add_argument("--max-depth", type=int, default=50)
    ap.add_argument("--spp-per-launch", type=int, default=0)
    ap.add_argument("--scene", default="cornell_box")
    ap.add_argument("--aspect", type=float, default=1.0)
    ap.add_argument("--n", type=int, default=0, help="scene 7 (bvh_stress): number of spheres")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target duration of the CPU baseline sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import raytracer_amd as R
    from raytracer_amd.distributed import allreduce_accum

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # RT_BENCH_REHEARSAL=1: rehearse the multi-rank path on a ONE-GPU box (every rank on device 0, gloo
    # instead of RCCL, which refuses two ranks on one device).  Not a measurement.
    rehearsal = os.environ.get("RT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    width = args.width
    height = R.image_height(width, args.aspect)
    hs = R.HostScene(args.scene, args.aspect, n=args.n)
    ds = R.DeviceScene(hs, local_rank)
    dev = torch.device("cuda", local_rank)
    accum = torch.zeros((height, width, 3), dtype=torch.float32, device=dev)
    rgb8 = torch.zeros((height, width, 3), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    from raytracer_amd import _ffi as F

    def step(i):
        # every step renders fresh global sample indices, so no step can reuse another's result
        offset = (i * world + rank) * args.spp
        st = ds.render_device(accum.data_ptr(), width, height, args.spp, args.max_depth, seed=1,
                              sample_offset=offset, spp_per_launch=args.spp_per_launch, stream=stream)
        allreduce_accum(accum)
        F.check_hip(F.hip.rt_tonemap_device(local_rank, accum.data_ptr(), width * height, args.spp * world,
                                            rgb8.data_ptr(), stream))
        return st

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    rays = 0
    kernel_ms = 0.0
    launches = 0
    nonfinite = 0
    for i in range(args.steps):
        st = step(args.warmup + i)
        rays += st["rays"]
        kernel_ms += st["kernel_ms"]
        launches += st["launches"]
        nonfinite += st["nonfinite_samples"]
    sync()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([elapsed, float(rays)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        total_rays = float(tot[1])
    else:
        total_rays = float(rays)

    if rank == 0:
        samples = float(width) * height * args.spp * world * args.steps
        out = {
            "metric": "Mrays/s (whole node) 1/2/4/8 MI355X, Cornell box 1024² 4096spp",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo)" if rehearsal else ""),
            "config": {"workload": f"{args.scene} {width}x{height}, {args.spp} spp per GPU, max depth {args.max_depth}"
                                   + (f", n={args.n}" if args.n else ""),
                       "scene_seed": 5, "bvh_seed": 7, "render_seed": 1,
                       "parallelism": f"sample-sharded x{world}, RCCL sum all-reduce of the accumulator"},
            "msamples_per_s": samples / elapsed / 1e6,
            "rays_per_sample": total_rays / samples,
            "nonfinite_samples_dropped": nonfinite,
        }
        # roofline of the dominant kernel (render_kernel<float>): live HIP-event duration on the launch
        # stream, algorithmic work from a counted pass of the same launch shape at 16 spp
        # (small scenes run the leaf-list walk, which visits no nodes: their algorithmic work is counted on
        # the reference-order BVH walk, RT_FLAG_FORCE_BVH; larger scenes are counted on the walk they run)
        c_spp = 16 if hs.desc.n_nodes < 4096 else 2
        st_c = ds.render_device(accum.data_ptr(), width, height, c_spp, args.max_depth, seed=1, sample_offset=0,
                                spp_per_launch=c_spp, stream=stream, stats=True)
        counted_on = "executed ordered BVH walk (box tests count as nodes)"
        if st_c["nodes_visited"] == 0 and hs.desc.n_nodes > 0:
            st_c = ds.render_device(accum.data_ptr(), width, height, c_spp, args.max_depth, seed=1, sample_offset=0,
                                    spp_per_launch=c_spp, stream=stream, stats=True, force_bvh=True)
            counted_on = "reference-order BVH walk of the same scene (RT_FLAG_FORCE_BVH)"
        flops_per_ray, bytes_per_ray = algorithmic_per_ray(st_c, width * height, st_c["launches"])
        avg_launch_ms = kernel_ms / max(1, launches)
        rays_per_launch = rays / max(1, launches)
        tflops = flops_per_ray * rays_per_launch / (avg_launch_ms * 1e-3) / 1e12
        gbps = bytes_per_ray * rays_per_launch / (avg_launch_ms * 1e-3) / 1e9
        valu_frac, hbm_frac = tflops / PEAK_FP32_VALU_TFLOPS, gbps / PEAK_HBM_GBPS
        # Which roof binds: a scene whose nodes + primitives fit in one XCD's 4 MiB L2 (here:
        # in LDS) never streams from HBM — its algorithmic bytes are served on-chip and the vector-FP32
        # issue rate is the roof; a scene larger than that is priced against HBM.
        d = hs.desc
        scene_bytes = 32 * d.n_nodes + 16 * d.n_spheres + 48 * d.n_mspheres + 32 * d.n_cubes + 128 * d.n_complex
        # measured once with PMC counters for the default workload (profiles/r01j_traffic.txt); null otherwise
        default_workload = (args.scene == "cornell_box" and width == 1024 and args.spp == 4096 and launches == args.steps)
        traffic = 3.2e9 if default_workload else None
        if scene_bytes <= 4 << 20:
            out["roofline"] = {"bound": "valu_fp32", "achieved": tflops, "peak": PEAK_FP32_VALU_TFLOPS,
                               "unit": "TFLOP/s", "frac": valu_frac, "traffic": traffic,
                               "traffic_note": "bytes per launch at the L2<->fabric boundary (upper bound: 5.03e7 float-atomic "
                                               "requests x 64 B + 4.5 MB reads; profiles/r01j_traffic.txt); the algorithmic "
                                               "bytes are served from LDS" if traffic else None}
        else:
            out["roofline"] = {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                               "frac": hbm_frac, "traffic": None}
        out["roofline"].update({
            "kernel": "render_refill_kernel" if "ordered" in counted_on else "render_persist_kernel",
            "algorithmic_work_counted_on": counted_on, "avg_launch_ms": avg_launch_ms, "launches": launches,
            "algorithmic_flops_per_ray": flops_per_ray, "algorithmic_bytes_per_ray": bytes_per_ray,
            "algorithmic_tflops": tflops, "algorithmic_gbps": gbps,
            "frac_valu_fp32": valu_frac, "frac_hbm": hbm_frac, "scene_bytes": scene_bytes,
            "note": "no MFMA on this path; the binding peak is vector FP32 (157.3 TFLOP/s), not a matrix peak",
        })
        if out["roofline"]["bound"] == "hbm" and scene_bytes <= 200 << 20:
            out["roofline"]["note"] += ("; this scene fits the 256 MiB Infinity Cache, so its algorithmic bytes are largely "
                                        "served on-die (L1/L2/MALL): frac prices them against HBM as the contract asks "
                                        "and can exceed 1 -- it is not an HBM utilisation (see profiles/ for EA counters)")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, width, height)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, width, height):
    """The CPU oracle (oracle/, a line-faithful f64 restatement of the reference: the Rust binary cannot
    be built here) timed on this box's host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    import raytracer_amd as R
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    threads = max(1, min(threads, os.cpu_count() or 1))
    sid = R.SCENE_IDS[args.scene]
    tex = R.load_earthmap() if sid in (2, 4, 8) else None
    sc = O.OracleScene(sid, args.aspect, n=args.n, texels=tex)
    # calibrate on a quarter-resolution 1-spp pass, then size the sample for ~cpu_seconds
    cw, ch = max(2, width // 4), max(2, height // 4)
    _, cnt, secs = sc.render(cw, ch, 1, args.max_depth, seed=1, threads=threads)
    rate = cnt["rays"] / max(secs, 1e-6)
    rays_per_spp_full = cnt["rays"] * (width * height) / float(cw * ch)
    spp = int(max(1, min(64, args.cpu_seconds * rate / max(1.0, rays_per_spp_full))))
    # rays of the sample: counted by the checker build (liboracle.so); time: the same sources built without the
    # work counters (liboracle_nocount.so, ~1.3x faster), same seed, so the same paths
    _, cnt, secs_counted = sc.render(width, height, spp, args.max_depth, seed=1, threads=threads)
    fast = O.OracleScene(sid, args.aspect, n=args.n, texels=tex, library="liboracle_nocount.so")
    _, _, secs = fast.render(width, height, spp, args.max_depth, seed=1, threads=threads)
    return {"value": cnt["rays"] / secs / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{args.scene} {width}x{height} at {spp} spp, depth {args.max_depth} ({secs:.1f} s, "
                      f"{cnt['rays']} rays; {secs_counted:.1f} s with the checker's work counters on); "
                      "rates are spp-independent",
            "msamples_per_s": width * height * spp / secs / 1e6}


if __name__ == "__main__":
    main()
