#!/usr/bin/env python3
"""Headline benchmark: Mrays/s, Cornell box 1024x1024 at 4096 spp, depth 50 (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W [--precision f64|f32] [--scaling weak|strong] [--config C5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one full pass of the hot path over the workload: every rank renders its samples of every pixel of the
Cornell box into an HBM-resident accumulator, then (N > 1) one RCCL sum all-reduce merges the accumulators, then the
image is tonemapped on the device.
  * --precision f64 (default): rt_render_f64_device — the reference computes in f64 throughout (alias.rs:4-7), so
    this is the line the metric is quoted on (`dtype` "f64"); the kernel follows the CPU oracle path for path.
    --precision f32: rt_render_device, the f32 product kernel (narrower than the reference's arithmetic).
  * --scaling weak (default): every GPU renders --spp samples per step (the merged image has N x spp);
    --scaling strong: the step's --spp samples are split over the ranks by shard_samples (north_star's
    "a 1024^2 render at 4096 spp on 8 GPUs"; --config C5 = 3840x2160, 65536 spp, strong: BASELINE configs[4]).
Rank 0 prints ONE JSON line.  At N = 1 the line also carries
  * `f32_leg`: the same workload on the f32 product kernel, with its own roofline (a peer line, not the metric:
    its arithmetic is narrower than the reference's);
  * `cpu_baseline`: the CPU oracle timed on this box's host cores on a bounded sample of the workload;
  * `other_configs`: one-GPU lines for BASELINE.json configs C1 / C3 / C4 (+ a scene that outgrows the Infinity
    Cache) at bounded spp.
`--gpus N` with N > 1 from a bare shell (no WORLD_SIZE) starts the N ranks itself, before any GPU call.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
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
PEAK_FP64_VALU_TFLOPS = 78.6    # vector FP64: half the FP32 rate
PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec
MALL_GATHER_GBPS = 8600.0       # MI355X_MICROARCH.md (L2 section): measured random-row gather from a 38 MB table that
                                # sits in the 256 MiB Infinity Cache (151 MB table: 7400-7900)
INFINITY_CACHE_BYTES = 256 << 20
METRIC = "Mrays/s (whole node) 1/2/4/8 MI355X, Cornell box 1024² 4096spp"


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


def measured_traffic(workload_key):
    """HBM-side bytes per launch of the dominant kernel from the newest committed PMC summary
    (profiles/*_traffic.json, written by tools/pmc_traffic.sh from separate --pmc passes; corrections per
    MI355X_MICROARCH.md section HBM).  None when no summary covers this workload: never a pasted constant."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get("workload") == workload_key:
            best = (rec, os.path.relpath(path, ROOT))
    return best


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` from a bare shell: start N fresh rank processes (torch.distributed.run) BEFORE this
    process has touched a GPU, relay rank 0's JSON line and the children's exit status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line:
        print(line, flush=True)
    return rc if rc else (0 if line else 1)


class Workload:
    """One scene resident on one GPU plus its output buffers."""

    def __init__(self, R, torch, scene, width, aspect, n, device):
        self.R, self.torch = R, torch
        self.scene, self.width, self.aspect, self.n = scene, width, aspect, n
        self.height = R.image_height(width, aspect)
        self.hs = R.HostScene(scene, aspect, n=n)
        self.ds = R.DeviceScene(self.hs, device)
        self.device = device
        self.dev = torch.device("cuda", device)
        self.accum = torch.zeros((self.height, width, 3), dtype=torch.float32, device=self.dev)
        self._accum64 = None
        self.rgb8 = torch.zeros((self.height, width, 3), dtype=torch.uint8, device=self.dev)
        self.stream = torch.cuda.current_stream().cuda_stream

    def accum_for(self, f64):
        if not f64:
            return self.accum
        if self._accum64 is None:
            self._accum64 = self.torch.zeros((self.height, self.width, 3), dtype=self.torch.float64, device=self.dev)
        return self._accum64

    def scene_bytes(self):
        """what the f32 product walk of this scene touches: the reference's tree (32-B nodes) for leaf-list scenes,
        the library's own SAH tree (64-B two-child nodes, one per leaf) for the ordered walk, plus the primitives"""
        d = self.hs.desc
        leaves = d.n_spheres + d.n_mspheres + d.n_cubes + d.n_complex
        node_bytes = 32 * d.n_nodes if d.n_nodes < 64 else 64 * max(0, leaves - 1)
        return node_bytes + 16 * d.n_spheres + 48 * d.n_mspheres + 32 * d.n_cubes + 128 * d.n_complex

    def counted(self, depth, f64=False):
        """Algorithmic work per ray from a counted pass of the same launch shape at low spp.  Small scenes run the
        leaf list, which visits no nodes: their work is counted on the reference-order BVH walk of the same
        precision (RT_FLAG_FORCE_BVH: the reference algorithm's own node and primitive tests); larger scenes are
        counted on the walk they run."""
        c_spp = 16 if self.hs.desc.n_nodes < 4096 else 2
        ptr = self.accum_for(f64).data_ptr()
        st = self.ds.render_device(ptr, self.width, self.height, c_spp, depth, seed=1, sample_offset=0,
                                   spp_per_launch=c_spp, stream=self.stream, stats=True, f64=f64)
        on = "the executed walk (box tests count as nodes)"
        if st["nodes_visited"] == 0 and self.hs.desc.n_nodes > 0:
            st = self.ds.render_device(ptr, self.width, self.height, c_spp, depth, seed=1, sample_offset=0,
                                       spp_per_launch=c_spp, stream=self.stream, stats=True, force_bvh=True, f64=f64)
            on = "reference-order BVH walk of the same scene and precision (RT_FLAG_FORCE_BVH)"
        flops, byts = algorithmic_per_ray(st, self.width * self.height, st["launches"])
        return flops, byts, on

    def kernel_name(self, counted_on, f64):
        d = self.hs.desc
        small = "RT_FLAG_FORCE_BVH" in counted_on or d.n_nodes == 0
        if f64:
            return "render_queue_kernel<double, ..., TRAV_LEAVES>" if small else "render_queue_kernel<double, ..., TRAV_ORDERED>"
        return "render_persist_kernel<float, ..., TRAV_LINEAR>" if small else "render_units_kernel"


def roofline(wl, depth, rays, kernel_ms, launches, workload_key, f64=False):
    flops_per_ray, bytes_per_ray, counted_on = wl.counted(depth, f64=f64)
    avg_launch_ms = kernel_ms / max(1, launches)
    rays_per_launch = rays / max(1, launches)
    tflops = flops_per_ray * rays_per_launch / (avg_launch_ms * 1e-3) / 1e12
    gbps = bytes_per_ray * rays_per_launch / (avg_launch_ms * 1e-3) / 1e9
    valu_peak = PEAK_FP64_VALU_TFLOPS if f64 else PEAK_FP32_VALU_TFLOPS
    valu_frac, hbm_frac = tflops / valu_peak, gbps / PEAK_HBM_GBPS
    # Which roof binds: a scene whose nodes + primitives fit in one XCD's 4 MiB L2 (here: in LDS)
    # never streams from HBM — its algorithmic bytes are served on-chip and the vector issue rate of the kernel's
    # arithmetic type is the roof; a larger scene is priced against the memory system.
    scene_bytes = wl.scene_bytes()
    tr = measured_traffic(workload_key)
    traffic = tr[0].get("bytes_per_launch") if tr else None
    if scene_bytes <= 4 << 20:
        r = {"bound": "valu_fp64" if f64 else "valu_fp32", "achieved": tflops, "peak": valu_peak, "unit": "TFLOP/s",
             "frac": valu_frac}
    else:
        # Memory-bound scenes.  Algorithmic bytes per second can exceed what HBM delivers when the scene sits in the
        # 256 MiB Infinity Cache (the cost model prices cache-served bytes): such a figure is not a utilisation.  With
        # a committed PMC summary, `achieved` is the MEASURED fabric-side traffic of a launch / its duration (what left
        # the L2s: Infinity-Cache hits included, so an upper bound on the HBM share) and `frac` = that / the HBM peak,
        # never above 1 by construction of the peak; the algorithmic rate stays beside it.  Without a summary,
        # `achieved` is the algorithmic rate, capped for `frac` at the on-die gather rate the guide measures.
        if traffic is not None:
            measured = traffic / (avg_launch_ms * 1e-3) / 1e9
            r = {"bound": "hbm", "achieved": measured, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": measured / PEAK_HBM_GBPS,
                 "achieved_basis": "measured fabric-side traffic per launch (FETCH_SIZE + WRITE_SIZE) / launch duration"}
        else:
            r = {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": min(1.0, hbm_frac),
                 "achieved_basis": "algorithmic bytes (no PMC summary for this workload: frac capped at 1)"}
        r["cache_resident"] = scene_bytes <= INFINITY_CACHE_BYTES
        r["frac_of_on_die_gather_rate"] = gbps / MALL_GATHER_GBPS
        r["on_die_gather_rate_gbps"] = MALL_GATHER_GBPS
    r["traffic"] = traffic
    if tr:
        r["traffic_source"] = tr[1]
        r["traffic_note"] = tr[0].get("note")
        for k in ("l2_hit_rate", "fetch_bytes_per_launch", "write_bytes_per_launch"):
            if k in tr[0]:
                r[k] = tr[0][k]
    r.update({
        "kernel": wl.kernel_name(counted_on, f64), "algorithmic_work_counted_on": counted_on,
        "avg_launch_ms": avg_launch_ms, "launches": launches,
        "algorithmic_flops_per_ray": flops_per_ray, "algorithmic_bytes_per_ray": bytes_per_ray,
        "algorithmic_tflops": tflops, "algorithmic_gbps": gbps,
        "frac_valu": valu_frac, "algorithmic_gbps_over_hbm_peak": hbm_frac, "scene_bytes": scene_bytes,
        "note": "no MFMA on this path; the compute peak is the vector rate of the kernel's arithmetic type "
                f"({valu_peak} TFLOP/s), not a matrix peak",
    })
    if r["bound"] == "hbm" and r.get("cache_resident"):
        r["note"] += ("; this scene fits the 256 MiB Infinity Cache: its algorithmic bytes are largely served on-die "
                      "(L1/L2/MALL), see frac_of_on_die_gather_rate; frac is the measured fabric-side share of the HBM peak")
    return r


def time_renders(wl, spp, depth, steps, warmup, spp_per_launch=0, f64=False, base_step=0):
    """`warmup` untimed + `steps` timed renders on one GPU; returns (seconds, rays, kernel_ms, launches, nonfinite)."""
    torch = wl.torch
    ptr = wl.accum_for(f64).data_ptr()
    for i in range(warmup):
        wl.ds.render_device(ptr, wl.width, wl.height, spp, depth, seed=1, sample_offset=(base_step + i) * spp,
                            spp_per_launch=spp_per_launch, stream=wl.stream, f64=f64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rays = launches = nonfinite = 0
    kernel_ms = 0.0
    for i in range(steps):
        st = wl.ds.render_device(ptr, wl.width, wl.height, spp, depth, seed=1,
                                 sample_offset=(base_step + warmup + i) * spp, spp_per_launch=spp_per_launch,
                                 stream=wl.stream, f64=f64)
        rays += st["rays"]
        kernel_ms += st["kernel_ms"]
        launches += st["launches"]
        nonfinite += st["nonfinite_samples"]
    torch.cuda.synchronize()
    return time.perf_counter() - t0, rays, kernel_ms, launches, nonfinite


def workload_key(scene, width, height, spp, depth, n=0, f64=False):
    """The key a profiles/*_traffic.json names its workload by (f32 keys are round 2's; f64 ones end in ', f64')."""
    return f"{scene} {width}x{height}, {spp} spp, max depth {depth}" + (f", n={n}" if n else "") + (", f64" if f64 else "")


def peer_leg(wl, args, f64):
    """The same workload, same spp, on the kernel of the OTHER precision, with its own roofline."""
    spp, steps = args.spp, args.leg_steps
    secs, rays, kernel_ms, launches, nonfinite = time_renders(wl, spp, args.max_depth, steps, 1, f64=f64, base_step=1000)
    samples = float(wl.width) * wl.height * spp * steps
    rl = roofline(wl, args.max_depth, rays, kernel_ms, launches,
                  workload_key(args.scene, wl.width, wl.height, spp, args.max_depth, args.n, f64), f64=f64)
    leg = {
        "dtype": "f64" if f64 else "f32", "entry": "rt_render_f64_device" if f64 else "rt_render_device",
        "workload": f"{args.scene} {wl.width}x{wl.height}, {spp} spp, max depth {args.max_depth}",
        "steps": steps, "warmup": 1, "ms_per_step": 1e3 * secs / steps,
        "value": rays / secs / 1e6, "unit": "Mrays/s", "msamples_per_s": samples / secs / 1e6,
        "rays_per_sample": rays / samples, "nonfinite_samples_dropped": nonfinite, "roofline": rl,
    }
    if f64:
        leg["note"] = ("walks every path to its end like the reference (no zero-throughput early exit, NaN samples poison "
                       "the pixel): compare with the f32 line in samples/s, not rays/s")
    else:
        leg["note"] = ("the f32 product kernel: narrower arithmetic than the reference's f64 (alias.rs:4-7), so this is a peer "
                       "line, not the metric; it stops zero-throughput paths and drops non-finite samples (rt.h), hence "
                       "fewer rays per sample: compare with the f64 line in samples/s")
    return leg


OTHER_CONFIGS = [
    # (tag, scene, width, aspect, n, full spp, bench spp, depth, f64 leg: (width, spp) or None)
    ("C1", "random_scene", 400, 16.0 / 9.0, 0, 100, 100, 50, (400, 100)),
    ("C3", "final_scene", 1024, 1.0, 0, 10000, 512, 50, (1024, 64)),
    ("C4", "bvh_stress", 3840, 16.0 / 9.0, 1000000, 256, 4, 50, (1920, 2)),
    # beyond BASELINE.json: the C4 scene grown until its tree (8 M leaves x 64-B nodes = 512 MB + 128 MB of spheres)
    # no longer fits the 256 MiB Infinity Cache.  Dense (scene 7: the spheres fill a quarter of the volume) a path
    # stays within a few units of the camera's visible shell and the touched set still caches (measured: 0.2 GB fetched
    # per launch, profiles/r03_c4hbm_traffic.json); sparse (scene 17: every radius x 0.1) a ray crosses the whole volume
    # and the walk streams the tree (3.3 TB/s fabric-side, L2 hit 0.49, profiles/r03_c4sparse_traffic.json) — the
    # configuration that is priced against HBM
    ("C4-hbm", "bvh_stress_sparse", 1920, 16.0 / 9.0, 8000000, 256, 8, 50, None),
]


def other_configs(R, torch, device, skip_big=False):
    """BASELINE.json configs C1, C3, C4 on one GPU at bounded spp (the rates are spp-independent once a launch is
    long enough; C1 is run at its full size), f32 product walk; `f64` = the same scene on the f64 kernel
    (the ordered walk in f64: same paths as the oracle's reference-order walk, fewer node visits) on a bounded sample."""
    lines = []
    for tag, scene, width, aspect, n, full_spp, spp, depth, leg64 in OTHER_CONFIGS:
        if skip_big and n > 1000000:
            continue
        t_build = time.perf_counter()
        wl = Workload(R, torch, scene, width, aspect, n, device)
        t_build = time.perf_counter() - t_build
        steps = 5 if tag == "C1" else 2
        secs, rays, kernel_ms, launches, nonfinite = time_renders(wl, spp, depth, steps, 1)
        key = workload_key(scene, width, wl.height, spp, depth, n)
        rl = roofline(wl, depth, rays, kernel_ms, launches, key)
        samples = float(width) * wl.height * spp * steps
        line = {"config": tag, "workload": key, "full_config_spp": full_spp, "steps": steps,
                "ms_per_step": 1e3 * secs / steps, "value": rays / secs / 1e6, "unit": "Mrays/s",
                "msamples_per_s": samples / secs / 1e6, "rays_per_sample": rays / samples,
                "nonfinite_samples_dropped": nonfinite, "dtype": "f32", "scene_build_s": t_build,
                "roofline": {k: rl[k] for k in rl if k not in ("traffic_note", "algorithmic_work_counted_on", "launches")}}
        del wl
        torch.cuda.empty_cache()
        if leg64:
            w64, spp64 = leg64
            wl = Workload(R, torch, scene, w64, aspect, n, device)
            secs, rays, kernel_ms, launches, _ = time_renders(wl, spp64, depth, 1, 1, f64=True)
            s64 = float(w64) * wl.height * spp64
            line["f64"] = {"workload": workload_key(scene, w64, wl.height, spp64, depth, n, True), "ms_per_step": 1e3 * secs,
                           "value": rays / secs / 1e6, "unit": "Mrays/s", "msamples_per_s": s64 / secs / 1e6,
                           "rays_per_sample": rays / s64, "kernel": "render_queue_kernel<double, ..., TRAV_ORDERED> (deep trees: render_persist_kernel)"}
            del wl
            torch.cuda.empty_cache()
        lines.append(line)
    return lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", choices=("f64", "f32"), default="f64",
                    help="arithmetic of the timed kernel; f64 = the reference's (alias.rs:4-7)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --spp per GPU per step; strong: --spp per step, split over the GPUs by shard_samples")
    ap.add_argument("--config", choices=("C2", "C5"), default="C2",
                    help="C2 = Cornell 1024^2 x 4096 spp (the metric's); C5 = Cornell 3840x2160 x 65536 spp, strong scaling")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel per step (per GPU when --scaling weak)")
    ap.add_argument("--max-depth", type=int, default=50)
    ap.add_argument("--spp-per-launch", type=int, default=0)
    ap.add_argument("--scene", default="cornell_box")
    ap.add_argument("--aspect", type=float, default=None)
    ap.add_argument("--n", type=int, default=0, help="scene 7 (bvh_stress): number of spheres")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--no-peer-leg", action="store_true", help="skip the other precision's line")
    ap.add_argument("--leg-steps", type=int, default=3)
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-big-scene", action="store_true", help="other_configs without the 8 M-sphere scene (15 s of host build)")
    args = ap.parse_args()
    if args.config == "C5":
        args.width = args.width or 3840
        args.aspect = args.aspect or 16.0 / 9.0
        args.spp = args.spp or 65536
        args.scaling = "strong"
    args.width = args.width or 1024
    args.aspect = args.aspect or 1.0
    args.spp = args.spp or 4096

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)  # nothing above has touched a GPU (torch is not even imported yet)

    import torch
    import torch.distributed as dist

    import raytracer_amd as R
    from raytracer_amd.distributed import allreduce_accum, shard_samples

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # RT_BENCH_REHEARSAL=1: rehearse the multi-rank path on a ONE-GPU box (every rank on device 0, gloo
    # instead of RCCL, which refuses two ranks on one device).  Not a measurement.
    rehearsal = os.environ.get("RT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # under a launcher (WORLD_SIZE set) the process group is set up even for one rank, so that
    # `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` runs the RCCL path on a one-GPU box
    distributed = world > 1 or "WORLD_SIZE" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            # set the communicator up now (RCCL does it at the first collective), so that a run with --warmup 0
            # does not time it
            dist.all_reduce(torch.zeros(1, device=torch.device("cuda", local_rank)))

    f64 = args.precision == "f64"
    strong = args.scaling == "strong"
    wl = Workload(R, torch, args.scene, args.width, args.aspect, args.n, local_rank)
    width, height = wl.width, wl.height
    dev = torch.device("cuda", local_rank)
    accum = wl.accum_for(f64)
    from raytracer_amd import _ffi as F
    tonemap = F.hip.rt_tonemap_f64_device if f64 else F.hip.rt_tonemap_device
    spp_image = args.spp if strong else args.spp * world  # samples per pixel of the merged image of one step

    def step(i):
        # every step renders fresh global sample indices, so no step can reuse another's result
        if strong:   # this rank's contiguous share of the step's samples (the rule of rt_multi_shard)
            offset, count = shard_samples(args.spp, rank, world, base_offset=i * args.spp)
        else:
            offset, count = (i * world + rank) * args.spp, args.spp
        st = wl.ds.render_device(accum.data_ptr(), width, height, count, args.max_depth, seed=1,
                                 sample_offset=offset, spp_per_launch=args.spp_per_launch, stream=wl.stream, f64=f64)
        if rehearsal and distributed:  # gloo reduces host tensors
            host = accum.cpu()
            allreduce_accum(host)
            accum.copy_(host)
        else:
            allreduce_accum(accum)
        F.check_hip(tonemap(local_rank, accum.data_ptr(), width * height, spp_image, wl.rgb8.data_ptr(), wl.stream))
        return st

    def sync():
        if distributed:
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

    tot = torch.tensor([elapsed, float(rays)], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if distributed:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        total_rays = float(tot[1])
    else:
        total_rays = float(rays)

    if rank == 0:
        samples = float(width) * height * spp_image * args.steps
        per = "per step, split over the GPUs" if strong else "per GPU"
        workload = (f"{args.scene} {width}x{height}, {args.spp} spp {per}, max depth {args.max_depth}"
                    + (f", n={args.n}" if args.n else ""))
        out = {
            "metric": METRIC,
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo)" if rehearsal else ""),
            "config": {"workload": workload, "scene_seed": 5, "bvh_seed": 7, "render_seed": 1,
                       "entry": "rt_render_f64_device" if f64 else "rt_render_device",
                       "parallelism": f"sample-sharded x{world} ({args.scaling} scaling), RCCL sum all-reduce of the "
                                      f"{args.precision} accumulator"},
            "msamples_per_s": samples / elapsed / 1e6,
            "rays_per_sample": total_rays / samples,
            "nonfinite_samples": nonfinite,
        }
        if f64:
            out["note"] = ("the reference's own arithmetic (f64, alias.rs:4-7); the kernel walks every path to its end like the "
                           "reference (5.1 rays per sample in this scene) and follows the CPU oracle path for path "
                           "(tests/test_gpu_parity.py); the f32 product kernel is the f32_leg line")
        # roofline of the dominant kernel: live HIP-event duration on the launch stream (rt_stats.kernel_ms, events
        # recorded on the stream the kernels are launched on), algorithmic work from a counted pass.  (Rank 0's own
        # launches: with strong scaling a launch renders this rank's share of the samples.)
        spp_launch = shard_samples(args.spp, 0, world)[1] if strong else args.spp
        key = workload_key(args.scene, width, height, spp_launch, args.max_depth, args.n, f64)
        out["roofline"] = roofline(wl, args.max_depth, rays, kernel_ms, launches, key, f64=f64)
        if world == 1 and not args.no_peer_leg:
            out["f32_leg" if f64 else "f64_leg"] = peer_leg(wl, args, not f64)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args, width, height)
            out["cpu_baseline"] = cb
            # like for like: the oracle walks every path to its end like the reference and like the f64 kernel; the
            # f32 kernel stops zero-throughput paths (fewer rays per sample), so compare in SAMPLES per second
            out["gpu_vs_cpu_msamples"] = out["msamples_per_s"] / cb["msamples_per_s"]
            for leg in ("f32_leg", "f64_leg"):
                if leg in out:
                    out[leg]["gpu_vs_cpu_msamples"] = out[leg]["msamples_per_s"] / cb["msamples_per_s"]
        if world == 1 and not args.no_other_configs:
            del wl
            torch.cuda.empty_cache()
            out["other_configs"] = other_configs(R, torch, local_rank, skip_big=args.no_big_scene)
        print(json.dumps(out), flush=True)

    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def native_oracle():
    """the timed CPU baseline is built -O3 -march=native ON THE BOX THAT TIMES IT (the shipped
    liboracle_nocount.so is built in another container, without -march=native).  Same sources, same flags otherwise
    (contraction off); written to the temporary directory, never into the repository.  Returns (library, build note);
    falls back to the shipped library when no compiler is at hand."""
    import shutil
    import tempfile
    cxx = shutil.which(os.environ.get("CXX", "g++"))
    src = os.path.join(ROOT, "oracle", "oracle_capi.cpp")
    if cxx and os.path.exists(src):
        out = os.path.join(tempfile.gettempdir(), f"liboracle_nocount_native_{os.getuid()}.so")
        cmd = [cxx, "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-DORC_NO_COUNT", "-shared",
               "-o", out, src]
        try:
            subprocess.run(cmd, check=True, timeout=300, capture_output=True, cwd=os.path.join(ROOT, "oracle"))
            return out, "g++ -O3 -march=native -ffp-contract=off, built on this box"
        except (subprocess.SubprocessError, OSError):
            pass
    return "liboracle_nocount.so", "g++ -O3 -ffp-contract=off (shipped build, no -march=native: no compiler on this box)"


def cpu_baseline(args, width, height):
    """The CPU oracle (oracle/, a line-faithful f64 restatement of the reference: the Rust binary cannot
    be built here) timed on this box's host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    import raytracer_amd as R
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    threads = max(1, min(threads, os.cpu_count() or 1))
    sid = R.SCENE_IDS[args.scene]
    tex = R.load_earthmap() if sid in (2, 4, 8, 11, 16) else None
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
    library, build_note = native_oracle()
    fast = O.OracleScene(sid, args.aspect, n=args.n, texels=tex, library=library)
    _, _, secs = fast.render(width, height, spp, args.max_depth, seed=1, threads=threads)
    return {"value": cnt["rays"] / secs / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "build": build_note,
            "sample": f"{args.scene} {width}x{height} at {spp} spp, depth {args.max_depth} ({secs:.1f} s, "
                      f"{cnt['rays']} rays; {secs_counted:.1f} s with the checker's work counters on); "
                      "rates are spp-independent",
            "msamples_per_s": width * height * spp / secs / 1e6,
            "rays_per_sample": cnt["rays"] / float(width * height * spp)}


if __name__ == "__main__":
    sys.exit(main())
