// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// Philox4x32 counter RNG (Salmon et al., "Parallel Random Numbers: As Easy as 1, 2, 3",
// SC'11) and the keyed draw streams that stand in for the reference's `rand` 0.8.3
// thread_rng (main.rs:161-162, cam.rs:69, alias.rs:41-91, hittable.rs:161,342,511,
// pdf.rs:36, material.rs:77, scenes.rs:64-94).  `rand`'s ChaCha12 stream is deliberately
// NOT restated (BASELINE.json north_star: "counter-based Philox RNG in place of the rand
// crate"); only the distributions are: uniform [0,1), uniform [a,b), uniform index.
#pragma once
#include <cstdint>

namespace orc {

struct Philox4 {
  uint32_t v[4];
};

// Rounds: the per-path draw streams use Philox4x32-7, the fewest rounds the paper reports as passing
// BigCrush ("Crush-resistant", SC'11 Table 2: 10 is its conservative default, 7 its validated minimum) —
// on the GPU the path streams are ~13 % of the kernel and every round costs two 32x32->64 multiplies.
// The host-side scene and BVH-axis streams (drawn once per scene) keep 10 rounds.
constexpr int PHILOX_PATH_ROUNDS = 7;
constexpr int PHILOX_SCENE_ROUNDS = 10;

inline Philox4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          int rounds) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int round = 0; round < rounds; ++round) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += W0;
    k1 += W1;
  }
  return Philox4{{c0, c1, c2, c3}};
}
inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                             uint32_t k1) {
  return philox4x32(c0, c1, c2, c3, k0, k1, 10);
}

// A draw stream: block j of stream (a, b, c) under key (k0, k1) is
// philox(ctr = (a, b, c, j)); draws are the words of block 0, 1, ... in order.
//   path streams : a = pixel index (y*w + x), b = global sample index,
//                  c = slot (0 = camera ray, 1 + bounce = one ray_color invocation)
//   scene streams: see scenes.h (a different key word k1).
struct Rng {
  uint32_t k0 = 0, k1 = 0;
  uint32_t a = 0, b = 0, c = 0, block = 0;
  uint32_t w[4] = {0, 0, 0, 0};
  int have = 0;
  int rounds = PHILOX_SCENE_ROUNDS;  // scene streams are default-constructed (scenes.h); path streams below
  uint64_t draws = 0;

  Rng() = default;
  Rng(uint64_t seed, uint32_t a_, uint32_t b_, uint32_t c_)
      : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), a(a_), b(b_), c(c_), rounds(PHILOX_PATH_ROUNDS) {}

  void begin(uint32_t slot) {
    c = slot;
    block = 0;
    have = 0;
  }
  uint32_t next_u32() {
    if (have == 0) {
      Philox4 r = philox4x32(a, b, c, block++, k0, k1, rounds);
      w[0] = r.v[0];
      w[1] = r.v[1];
      w[2] = r.v[2];
      w[3] = r.v[3];
      have = 4;
    }
    uint32_t r = w[4 - have];
    --have;
    ++draws;
    return r;
  }
  // A draw that belongs to one constant medium of the current slot: word 0 of block
  // 0x80000000 | object id.  Keyed, not sequential, so it does not depend on the order in which a
  // traversal reaches the media (stands in for rand::random() at hittable.rs:342).
  double keyed_f64(uint32_t object_id) {
    Philox4 r = philox4x32(a, b, c, 0x80000000u | object_id, k0, k1, rounds);
    ++draws;
    return (double)r.v[0] * (1.0 / 4294967296.0);
  }
  // uniform [0,1): one 32-bit word, exact in f64 (stands in for rand's gen::<f64>()).
  double next_f64() { return (double)next_u32() * (1.0 / 4294967296.0); }
  // uniform [lo,hi) (stands in for gen_range(lo..hi) on f64).
  double range(double lo, double hi) { return lo + (hi - lo) * next_f64(); }
  // uniform index in [0,n) (stands in for gen_range(0..n) / SliceRandom::choose).
  uint32_t below(uint32_t n) { return (uint32_t)(((uint64_t)next_u32() * n) >> 32); }
};

}  // namespace orc
