// Philox4x32 (Salmon et al., SC'11; 7 rounds for the path streams, see RT_PATH_PHILOX_ROUNDS) and the per-path draw streams that replace the
// reference's `rand` thread_rng (main.rs:161-162, cam.rs:69, alias.rs:41-91,
// hittable.rs:342,511, pdf.rs:36, material.rs:77).
//
// Stream layout: block j of stream (pixel, sample, slot) under key (seed_lo, seed_hi) is
// philox(ctr = (pixel, sample, slot, j)).  slot 0 = camera ray (jitter u, jitter v, lens
// disk, shutter time); slot 1+b = bounce b (medium draws during traversal first, then the
// material / PDF draws).  A slot's first block is generated unconditionally by every lane
// at the top of a bounce, so the 10 rounds run wave-uniform; further blocks (rejection
// loops, long medium chains) are drawn on demand.
#pragma once
#include "vec.hpp"

namespace rt {

struct U4 {
  uint32_t x, y, z, w;
};

// Path streams run Philox4x32-7: the fewest rounds Salmon et al. report as passing BigCrush (10 is their
// conservative default).  Measured on the Cornell kernel: 54.8 -> 57.7 Grays/s.  The oracle uses the same
// constant (oracle/src/philox.h); the host's scene and BVH-axis streams keep 10 rounds.
#ifndef RT_PATH_PHILOX_ROUNDS
#define RT_PATH_PHILOX_ROUNDS 7
#endif

template <int ROUNDS>
RT_DEV U4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    // one 32x32->64 multiply each (v_mad_u64_u32) instead of a mul_hi + mul_lo pair
    const uint64_t p0 = (uint64_t)M0 * c0;
    const uint64_t p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c0 = n0;
    c1 = (uint32_t)p1;
    c2 = n2;
    c3 = (uint32_t)p0;
    k0 += W0;
    k1 += W1;
  }
  return U4{c0, c1, c2, c3};
}
RT_DEV U4 philox_path(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return philox4x32<RT_PATH_PHILOX_ROUNDS>(c0, c1, c2, c3, k0, k1);
}

template <typename R> RT_DEV R u01(uint32_t x);
// one 32-bit word per draw in both precisions, so f32 and f64 runs consume the same words
template <> RT_DEV double u01<double>(uint32_t x) { return (double)x * (1.0 / 4294967296.0); }
template <> RT_DEV float u01<float>(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

struct Rng {
  uint32_t k0, k1, pixel, sample, slot, block;
  uint32_t w0, w1, w2, w3;
  uint32_t have;

  RT_DEV void init(uint64_t seed, uint32_t pixel_, uint32_t sample_) {
    k0 = (uint32_t)seed;
    k1 = (uint32_t)(seed >> 32);
    pixel = pixel_;
    sample = sample_;
    slot = 0;
    block = 0;
    have = 0;
    w0 = w1 = w2 = w3 = 0;
  }
  RT_DEV void refill() {
    U4 r = philox_path(pixel, sample, slot, block, k0, k1);
    ++block;
    w0 = r.x;
    w1 = r.y;
    w2 = r.z;
    w3 = r.w;
    have = 4;
  }
  RT_DEV void begin(uint32_t slot_) {
    slot = slot_;
    block = 0;
    refill();
  }
  // select a slot without generating its first block (keyed draws only need the counter)
  RT_DEV void set_slot(uint32_t slot_) {
    slot = slot_;
    block = 0;
    have = 0;
  }
  RT_DEV uint32_t next_u32() {
    if (have == 0) refill();
    uint32_t r = w0;
    w0 = w1;
    w1 = w2;
    w2 = w3;
    --have;
    return r;
  }
  // The draw of one constant medium in the current slot: word 0 of block 0x80000000 | object id.
  // Keyed instead of sequential, so any traversal order gives every medium the same draw.
  template <typename R> RT_DEV R keyed(uint32_t object_id) const {
    return u01<R>(philox_path(pixel, sample, slot, 0x80000000u | object_id, k0, k1).x);
  }
  // ln of that draw (hittable.rs:342: `ln(rand::random())`), from all 32 bits of the word in BOTH precisions.  u01<float>
  // keeps 24 bits, and near u = 1 (a short free path) ln(u) ~ -(1 - u) then carries a relative error of 2^-24 / (1 - u):
  // seen as 5e-4 on the primary t of final_scene's fog.  Here u < 1/2 goes through ln(x 2^-32) and u >= 1/2 through
  // log1p(-(2^32 - x) 2^-32): in either branch the argument has full relative precision, so ln(u) agrees with the f64
  // kernel's (and the oracle's) to f32 rounding.  x = 0 gives -inf (no scatter) in both.
  template <typename R> RT_DEV R keyed_log(uint32_t object_id) const {
    const uint32_t x = philox_path(pixel, sample, slot, 0x80000000u | object_id, k0, k1).x;
    if constexpr (sizeof(R) == 8) {
      return log(u01<double>(x));
    } else {
      const float lo = logf((float)x * (1.0f / 4294967296.0f));
      const float hi = log1pf(-((float)(0u - x) * (1.0f / 4294967296.0f)));
      return x < 0x80000000u ? lo : hi;
    }
  }
  template <typename R> RT_DEV R next() { return u01<R>(next_u32()); }
  template <typename R> RT_DEV R range(R lo, R hi) { return lo + (hi - lo) * next<R>(); }
  RT_DEV uint32_t below(uint32_t n) { return __umulhi(next_u32(), n); }
};

}  // namespace rt
