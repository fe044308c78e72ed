// Plain data shared by the wavefront kernels (device/wavefront.hpp) and their host orchestration (rt_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

struct WfPool {  // P slots, structure of arrays
  float4* ray_o;      // origin xyz, time
  float4* ray_d;      // direction xyz, t of the closest hit (written by the trace kernel)
  uint32_t* hit;      // leaf reference of the closest hit, REF_NONE = miss (written by the trace kernel)
  float4* thr_l;      // throughput xyz, radiance x
  float4* l_ids;      // radiance y z, pixel (bits), sample (bits)
  uint32_t* slot;     // Philox slot of the ray in flight = 1 + bounce index; 0 = the slot holds no path
  uint32_t* list[2];  // TAIL mode only: the slots that still hold a path, for iteration k in list[k & 1]
  uint32_t n_slots;
};

// The (pixel, sample) jobs are dealt to WF_SHARDS cursors — job block b (64 jobs: one 8x8 tile, one sample) belongs to
// shard b % WF_SHARDS — because a single word takes only ~88 atomics per microsecond and the pool ends ~10^6 paths per
// iteration, one wave-wide request each ~25 of them.  A wave asks its own shard first (its wave index % WF_SHARDS:
// neighbouring waves then render neighbouring tiles), then the others in turn.
constexpr uint32_t WF_SHARDS = 64;
constexpr uint32_t WF_SHARD_STRIDE = 32;  // uints: each cursor on a 128-byte line of its own
constexpr uint32_t WF_TRACE_ENTRIES = 1024;  // entries (slots, or in tail mode list entries) per wave of the trace kernel

struct WfCounters {  // device memory, zeroed per render call
  unsigned int job_cursor[WF_SHARDS * WF_SHARD_STRIDE];  // jobs of each shard handed out so far
  unsigned int retired;     // slots that found no job left: the call is over when this reaches n_slots
  unsigned int count[2];    // TAIL mode: entries of list[0], list[1]
  unsigned int pad;
  unsigned long long spent;  // bit s: shard s has no jobs left (set by whoever finds out; read once per wave and request)
};

struct WfArgs {
  uint32_t width, height, sample_begin, n_samples, max_depth;
  uint32_t tiles_x, n_tiles;  // 8x8 pixel tiles
  uint64_t total_jobs;        // 64 * n_tiles * n_samples: job j = (sample j / (64 n_tiles), tile (j / 64) % n_tiles, pixel j % 64)
  uint64_t n_blocks;          // job blocks of 64: n_tiles * n_samples
  uint32_t stack_depth, refill_eighths, leaf_shift;
  uint32_t has_media;
  uint32_t in_list, out_list; // this iteration reads / writes the live-slot list (tail) instead of walking all slots
  uint64_t seed;
};

}  // namespace rt
