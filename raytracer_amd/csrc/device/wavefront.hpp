// WAVEFRONT form of the render closure (main.rs:154-170) for BVH-bound scenes (f32): a pool of P paths in HBM, and per
// iteration two kernels — TRACE (world.hit of every live path's current ray, hittable.rs:400-412 as the ordered walk of
// intersect.hpp) and SHADE (one vertex of ray_color, main.rs:77-104, then the next camera sample in place when the path
// has ended).
//
// Why, next to the megakernel (render_refill_kernel): in a BVH-bound scene the megakernel is bound by VALU issue at a
// lane utilisation of ~0.4 — a lane whose walk has ended idles until enough of its wave's lanes have ended theirs to
// make the (long, divergent) shading pass worth running — and it holds the whole path state and the shading code's
// registers through the walk (96 VGPRs, 5 waves per SIMD).  Here a lane whose walk has ended stores 8 bytes and
// fetches the next ray from the pool (32 bytes) at once, the trace kernel needs only the walk's registers, and the
// shade kernel runs one thread per path.  The price is the path state's round trip through memory: ~190 B per ray,
// served from L2 / Infinity Cache at the sizes used (72 B per pool slot).
//
// Same paths and draws as every other f32 kernel: a sample's Philox streams are keyed by (pixel, sample, slot), the
// leaf tests, hit records and shade() are the shared code, a finished sample is added to the accumulator with float
// atomics (only the f32 order of a pixel's sample sum differs).
#pragma once
#include "intersect.hpp"
#include "wf_types.hpp"

namespace rt {

#ifndef RT_WF_TRACE_MIN_WAVES
#define RT_WF_TRACE_MIN_WAVES 8
#endif
#ifndef RT_WF_SHADE_MIN_WAVES
#define RT_WF_SHADE_MIN_WAVES 5
#endif

// The camera ray of job j into slot i (cam.rs:63-71, main.rs:161-163).  Returns false if the job's pixel lies outside
// the image (a tile that straddles the edge).
RT_DEV bool wf_camera_job(const CameraView<float>& cam, const WfArgs& a, uint64_t j, Rng& rng, RayT<float>& ray) {
  const uint64_t q = j >> 6;
  const uint32_t lp = (uint32_t)j & 63u, smp = (uint32_t)(q / a.n_tiles), tile = (uint32_t)(q - (uint64_t)smp * a.n_tiles);
  const uint32_t px = (tile % a.tiles_x) * 8u + (lp & 7u), py = (tile / a.tiles_x) * 8u + (lp >> 3);
  if (px >= a.width || py >= a.height) return false;
  rng.pixel = py * a.width + px;
  rng.sample = a.sample_begin + smp;
  rng.begin(0);
  const float ju = rng.next<float>();
  const float jv = rng.next<float>();
  const float u = rdiv((float)px + ju, (float)(a.width - 1));                  // main.rs:161
  const float v = rdiv((float)(a.height - py - 1) + jv, (float)(a.height - 1));  // main.rs:162-163
  ray = camera_ray(cam, u, v, rng);
  return true;
}

// Next job for the lanes that ask for one: one atomic per wave and round on a shard's cursor (see WF_SHARDS).  Jobs
// outside the image are skipped.  A lane that finds every shard spent returns false.
RT_DEV bool wf_take_job(const CameraView<float>& cam, const WfArgs& a, WfCounters* ctr, bool want, Rng& rng, RayT<float>& ray) {
  bool got = false;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t home = (blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) % WF_SHARDS;  // wave-uniform
  uint32_t tried = 0;                                                                  // wave-uniform
  if (__ballot(want) == 0) return false;
  static_assert(WF_SHARDS == 64, "the spent mask is one 64-bit word");
  unsigned long long spent = __hip_atomic_load(&ctr->spent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  spent = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(spent >> 32)) << 32) |
          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)spent);
  while (tried < WF_SHARDS) {
    const uint64_t m = __ballot(want && !got);
    if (m == 0) break;
    const uint32_t s = (home + tried) % WF_SHARDS;
    if ((spent >> s) & 1ull) {
      ++tried;
      continue;
    }
    // shard s owns the blocks s, s + WF_SHARDS, ...: n_s of them, 64 jobs each
    const uint64_t n_s = a.n_blocks > s ? (a.n_blocks - s + WF_SHARDS - 1) / WF_SHARDS : 0;
    const uint64_t jobs_s = n_s * 64ull;
    uint32_t base = 0;
    const uint32_t leader = (uint32_t)__builtin_ctzll(m);
    if (lane == leader) base = atomicAdd(&ctr->job_cursor[s * WF_SHARD_STRIDE], (unsigned int)__popcll(m));
    base = (uint32_t)__shfl((int)base, (int)leader, 64);
    if ((uint64_t)base + (uint64_t)__popcll(m) > jobs_s) {  // this shard ends inside (or before) this request
      ++tried;
      if (lane == leader) atomicOr(&ctr->spent, 1ull << s);
    }
    if (want && !got) {
      const uint64_t q = (uint64_t)base + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
      if (q < jobs_s) {
        const uint64_t block = (uint64_t)s + (q >> 6) * WF_SHARDS;
        got = wf_camera_job(cam, a, block * 64ull + (q & 63ull), rng, ray);
      }
    }
  }
  return got;
}

RT_DEV void wf_store_path(const WfPool& p, uint32_t i, const RayT<float>& ray, Vec3<float> thr, Vec3<float> L, uint32_t pixel,
                          uint32_t sample, uint32_t slot) {
  p.ray_o[i] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.time);
  p.ray_d[i] = make_float4(ray.d.x, ray.d.y, ray.d.z, 0.0f);
  p.thr_l[i] = make_float4(thr.x, thr.y, thr.z, L.x);
  p.l_ids[i] = make_float4(L.y, L.z, __uint_as_float(pixel), __uint_as_float(sample));
  p.slot[i] = slot;
}

// Append slot i to a list for the lanes with `yes` (one atomic per wave; the wave's entries stay in lane order).
RT_DEV void wf_list_append(uint32_t* list, unsigned int* count, bool yes, uint32_t i) {
  const uint64_t m = __ballot(yes);
  if (m == 0) return;
  const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__builtin_ctzll(m);
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(count, (unsigned int)__popcll(m));
  base = __shfl(base, (int)leader, 64);
  if (yes) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
}

// Fill the pool: slot i takes a job (or stays empty when there are fewer jobs than slots).
__global__ void __launch_bounds__(BLOCK) wf_init_kernel(CameraView<float> cam, WfArgs a, WfPool pool, WfCounters* ctr) {
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  const bool mine = i < pool.n_slots;
  Rng rng;
  rng.init(a.seed, 0, 0);
  RayT<float> ray;
  const bool got = wf_take_job(cam, a, ctr, mine && a.max_depth > 0, rng, ray);
  const uint64_t rm = __ballot(mine && !got);
  if (rm && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(rm)) atomicAdd(&ctr->retired, (unsigned int)__popcll(rm));
  if (!mine) return;
  if (got)
    wf_store_path(pool, i, ray, Vec3<float>(1, 1, 1), Vec3<float>(0, 0, 0), rng.pixel, rng.sample, 1u);
  else
    pool.slot[i] = 0u;
}

// TRACE: a wave owns WF_TRACE_ENTRIES consecutive entries (all slots, or in tail mode the list of live slots); a lane
// whose walk has ended stores its hit and takes the wave's next entry, so the lanes stay busy until the wave's range is
// spent (no atomics: the hardware's workgroup dispatch balances the waves).
template <bool STATS>
__global__ void __launch_bounds__(BLOCK, RT_WF_TRACE_MIN_WAVES) wf_trace_kernel(SceneView<float> sc, WfArgs a, WfPool pool, WfCounters* ctr,
                                                                                uint32_t parity, DeviceStats* stats) {
  using R = float;
  extern __shared__ uint32_t lds_stack[];
  uint32_t* stack = lds_stack + threadIdx.x;
  uint16_t* stack_t = reinterpret_cast<uint16_t*>(lds_stack + a.stack_depth * BLOCK) + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t* list = pool.list[parity];
  const uint32_t n_entries = a.in_list ? ctr->count[parity] : pool.n_slots;
  // the other list's count must be zero before this iteration's shade kernel appends to it; the previous shade kernel
  // was its last reader, and nothing in this kernel reads it
  if (blockIdx.x == 0 && threadIdx.x == 0) ctr->count[parity ^ 1u] = 0u;
  const uint64_t first = ((uint64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * WF_TRACE_ENTRIES;
  if (first >= n_entries) return;  // (whole waves leave together; nothing below is workgroup-wide)
  uint32_t pos = (uint32_t)first;                                                              // wave-uniform
  const uint32_t end = first + WF_TRACE_ENTRIES < n_entries ? (uint32_t)first + WF_TRACE_ENTRIES : n_entries;
  Cnt<STATS> cnt;
  cnt.clear();
  bool walking = false;
  uint32_t idx = 0;
  RayT<R> ray;
  Rng rng;
  rng.init(a.seed, 0, 0);
  OrderedWalk walk;
  walk.cur = REF_NONE;
  walk.sp = 0;
  walk.best = REF_NONE;
  walk.closest = 0.0f;
  walk.a = 0.0f;
  const uint32_t refill_q = (64u * a.refill_eighths) >> 3;
  const uint32_t refill_at = refill_q > 0 ? refill_q : 1u;
  for (;;) {
    // (1) idle lanes take the wave's next entries
    if (pos < end) {
      const bool want = !walking;
      const uint64_t m = __ballot(want);
      const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (want && pos + rank < end) {
        const uint32_t e = pos + rank;
        idx = a.in_list ? list[e] : e;
        const uint32_t slot = pool.slot[idx];
        if (slot != 0u) {
          const float4 o = pool.ray_o[idx], d = pool.ray_d[idx];
          ray.o = Vec3<R>(o.x, o.y, o.z);
          ray.time = o.w;
          ray.d = Vec3<R>(d.x, d.y, d.z);
          if (a.has_media) {  // a constant medium's free-path draw is keyed by (pixel, sample, slot, object)
            const float4 ids = pool.l_ids[idx];
            rng.pixel = __float_as_uint(ids.z);
            rng.sample = __float_as_uint(ids.w);
            rng.set_slot(slot);
          }
          cnt.template inc<ST_RAYS>();
          walking = walk.start(sc, ray);
          if (!walking) {  // empty world
            pool.ray_d[idx].w = walk.closest;
            pool.hit[idx] = walk.best;
          }
        }
      }
      const uint32_t n_want = (uint32_t)__popcll(m);
      pos = pos + n_want < end ? pos + n_want : end;
    }
    const bool drained = pos >= end;  // wave-uniform
    if (__ballot(walking) == 0) {
      if (drained) break;
      continue;
    }
    // (2) step the walking lanes until enough of them have ended (or, once the range is spent, until all have)
    for (;;) {
      const uint32_t n_walking = (uint32_t)__popcll(__ballot(walking));
      if (n_walking == 0 || (!drained && n_walking + refill_at <= 64u)) break;
      const uint32_t leaves_at = (n_walking >> a.leaf_shift) > 0 ? (n_walking >> a.leaf_shift) : 1u;
      for (;;) {
        const bool at_node = walking && !(walk.cur & REF_LEAF);
        const uint32_t n_node = (uint32_t)__popcll(__ballot(at_node));
        const uint32_t n_walk_now = (uint32_t)__popcll(__ballot(walking));
        if (n_node == 0 || n_walk_now - n_node >= leaves_at || (!drained && n_walk_now + refill_at <= 64u)) break;
        if (at_node) {
          walking = walk.node_step(sc, ray, R(0.001), stack, stack_t, (uint32_t)BLOCK, cnt);
          if (!walking) {
            pool.ray_d[idx].w = walk.closest;
            pool.hit[idx] = walk.best;
          }
        }
      }
      if (walking && (walk.cur & REF_LEAF)) {
        walking = walk.leaf_step(sc, ray, R(0.001), rng, stack, stack_t, (uint32_t)BLOCK, cnt);
        if (!walking) {
          pool.ray_d[idx].w = walk.closest;
          pool.hit[idx] = walk.best;
        }
      }
    }
  }
  flush_stats(cnt, stats);
}

// SHADE: one thread per slot — one vertex of ray_color (main.rs:77-104) for the hit the trace kernel found, then either
// the scattered ray, or (path over) the sample's radiance into the accumulator and the next camera sample in place.
template <bool STATS>
__global__ void __launch_bounds__(BLOCK, RT_WF_SHADE_MIN_WAVES) wf_shade_kernel(SceneView<float> sc, CameraView<float> cam, WfArgs a, WfPool pool,
                                                                                WfCounters* ctr, uint32_t parity, float* __restrict__ accum,
                                                                                DeviceStats* stats) {
  using R = float;
  const uint32_t n_entries = a.in_list ? ctr->count[parity] : pool.n_slots;
  if (blockIdx.x * BLOCK >= n_entries) return;
  __shared__ Light<R> s_lights[LDS_MAX_LIGHTS];
  {
    const uint32_t n_l = sc.n_lights < LDS_MAX_LIGHTS ? sc.n_lights : LDS_MAX_LIGHTS;
    constexpr uint32_t LW = sizeof(Light<R>) / 4;
    for (uint32_t i = threadIdx.x; i < n_l * LW; i += BLOCK)
      reinterpret_cast<uint32_t*>(s_lights)[i] = reinterpret_cast<const uint32_t*>(sc.lights)[i];
    __syncthreads();
  }
  const Light<R>* lights = sc.n_lights <= LDS_MAX_LIGHTS ? s_lights : sc.lights;
  const uint32_t tid = blockIdx.x * BLOCK + threadIdx.x;
  const uint32_t i = tid < n_entries ? (a.in_list ? pool.list[parity][tid] : tid) : 0u;
  Cnt<STATS> cnt;
  cnt.clear();
  uint32_t nonfinite = 0;
  uint32_t slot = tid < n_entries ? pool.slot[i] : 0u;
  const bool live = slot != 0u;
  Rng rng;
  rng.init(a.seed, 0, 0);
  RayT<R> ray;
  Vec3<R> thr, L;
  bool ended = false;
  if (live) {
    const float4 o = pool.ray_o[i], d = pool.ray_d[i], t0 = pool.thr_l[i], t1 = pool.l_ids[i];
    const uint32_t best = pool.hit[i];
    ray.o = Vec3<R>(o.x, o.y, o.z);
    ray.time = o.w;
    ray.d = Vec3<R>(d.x, d.y, d.z);
    thr = Vec3<R>(t0.x, t0.y, t0.z);
    L = Vec3<R>(t0.w, t1.x, t1.y);
    rng.pixel = __float_as_uint(t1.z);
    rng.sample = __float_as_uint(t1.w);
    rng.begin(slot);
    const Vec3<R> background(sc.background[0], sc.background[1], sc.background[2]);
    bool cont;
    if (best == REF_NONE) {  // main.rs:104
      cnt.template inc<ST_MISSES>();
      L = madd(L, thr, background);
      cont = false;
    } else {
      cnt.template inc<ST_CLOSEST>();
      HitRec<R> h;
      make_hit_record(sc, best, ray, d.w, false, h);
      const Material<R> mat = sc.materials[h.material];
      cont = shade(sc, lights, mat, h, ray, thr, L, rng, cnt);
    }
    if (slot == a.max_depth && cont) {  // the next ray_color call would return 0 (slot b = the b-th ray of the path)
      depth_exhausted(L, thr);
      cont = false;
    }
    if (cont && thr.x == R(0) && thr.y == R(0) && thr.z == R(0)) cont = false;
    ++slot;
    if (!cont) {
      const bool fin = rfinite(L.x) && rfinite(L.y) && rfinite(L.z);
      if (!fin) ++nonfinite;
      if (fin) {
        float* out = accum + 3ull * rng.pixel;
        atomicAdd(out + 0, L.x);
        atomicAdd(out + 1, L.y);
        atomicAdd(out + 2, L.z);
      }
      ended = true;
    }
  }
  // paths that ended: the next camera sample in place (wave-wide: every lane of the wave runs the loop's ballots)
  const bool got = wf_take_job(cam, a, ctr, ended, rng, ray);
  if (ended) {
    if (got) {
      thr = Vec3<R>(1, 1, 1);
      L = Vec3<R>(0, 0, 0);
      slot = 1u;
    } else {
      slot = 0u;
    }
  }
  const uint64_t rm = __ballot(ended && !got);
  if (rm && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(rm)) atomicAdd(&ctr->retired, (unsigned int)__popcll(rm));
  if (live) {
    if (slot != 0u)
      wf_store_path(pool, i, ray, thr, L, rng.pixel, rng.sample, slot);
    else
      pool.slot[i] = 0u;
  }
  if (a.out_list) wf_list_append(pool.list[parity ^ 1u], &ctr->count[parity ^ 1u], live && slot != 0u, i);  // tail: next iteration's list
  flush_stats(cnt, stats);
  const uint32_t nf = wave_sum(nonfinite);
  if ((threadIdx.x & 63) == 0 && nf) atomicAdd(&stats->v[ST_NONFINITE], (unsigned long long)nf);
}

}  // namespace rt
