// oracle/src/stdrng.h — TEST INFRASTRUCTURE (see oracle/README.md).
//
// `StdRng::seed_from_u64(5)` of scenes.rs:64, restated from the published algorithms of the crates the reference pins
// (Cargo.lock: rand 0.8.3 -> rand_core 0.6, rand_chacha 0.3); the crates' sources are not in /root/reference:
//   * StdRng = ChaCha12Rng (rand 0.8 rngs/std.rs): the ChaCha block function with 12 rounds, a 64-bit block counter
//     in state words 12-13 starting at 0, a 64-bit stream id in words 14-15 = 0; output = the words of block 0,
//     then block 1, ...;
//   * SeedableRng::seed_from_u64 (rand_core 0.6): the 32-byte key is 8 outputs of a PCG32 (XSH-RR) started at the
//     u64, state advanced BEFORE each output, little-endian;
//   * next_u64 = two consecutive words, low word first (rand_core BlockRng on a u32 buffer, even index);
//   * gen::<f64>() (Standard): (next_u64() >> 11) * 2^-53;
//   * gen_range(low..high) for f64 (UniformFloat::sample_single): value1_2 = bits (next_u64() >> 12) with exponent 0,
//     res = (value1_2 - 1.0) * (high - low) + low, accepted if res < high (always, for the ranges of scenes.rs).
// PARITY: the ChaCha core is pinned by RFC 7539's vectors (quarter round 2.1.1, 20-round block 2.3.2;
// tests/test_stdrng.py); the seeding and the float conversions have no vector available offline — "parity unpinned".
#pragma once
#include <cstdint>
#include <cstring>

namespace orc {

inline uint32_t rotl32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
inline void chacha_quarter(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
  a += b; d ^= a; d = rotl32(d, 16);
  c += d; b ^= c; b = rotl32(b, 12);
  a += b; d ^= a; d = rotl32(d, 8);
  c += d; b ^= c; b = rotl32(b, 7);
}
// in: the 16 state words; out = in + rounds(in); `rounds` even
inline void chacha_block(const uint32_t in[16], int rounds, uint32_t out[16]) {
  uint32_t x[16];
  std::memcpy(x, in, sizeof(x));
  for (int r = 0; r < rounds; r += 2) {
    chacha_quarter(x[0], x[4], x[8], x[12]);
    chacha_quarter(x[1], x[5], x[9], x[13]);
    chacha_quarter(x[2], x[6], x[10], x[14]);
    chacha_quarter(x[3], x[7], x[11], x[15]);
    chacha_quarter(x[0], x[5], x[10], x[15]);
    chacha_quarter(x[1], x[6], x[11], x[12]);
    chacha_quarter(x[2], x[7], x[8], x[13]);
    chacha_quarter(x[3], x[4], x[9], x[14]);
  }
  for (int i = 0; i < 16; ++i) out[i] = x[i] + in[i];
}

struct StdRng {
  uint32_t state[16];
  uint32_t buf[16];
  int pos = 16;

  static StdRng seed_from_u64(uint64_t s) {
    StdRng g;
    g.state[0] = 0x61707865u;  // "expand 32-byte k"
    g.state[1] = 0x3320646eu;
    g.state[2] = 0x79622d32u;
    g.state[3] = 0x6b206574u;
    for (int i = 0; i < 8; ++i) {
      s = s * 6364136223846793005ull + 11634580027462260723ull;
      const uint32_t xorshifted = (uint32_t)(((s >> 18) ^ s) >> 27);
      const uint32_t rot = (uint32_t)(s >> 59);
      g.state[4 + i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    g.state[12] = g.state[13] = g.state[14] = g.state[15] = 0;
    return g;
  }
  uint32_t next_u32() {
    if (pos == 16) {
      chacha_block(state, 12, buf);
      if (++state[12] == 0) ++state[13];
      pos = 0;
    }
    return buf[pos++];
  }
  uint64_t next_u64() {
    const uint64_t lo = next_u32();
    return lo | ((uint64_t)next_u32() << 32);
  }
  double next_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
  double range(double low, double high) {
    const double scale = high - low;
    for (;;) {
      const uint64_t bits = (next_u64() >> 12) | 0x3FF0000000000000ull;
      double value1_2;
      std::memcpy(&value1_2, &bits, 8);
      const double res = (value1_2 - 1.0) * scale + low;
      if (res < high) return res;
    }
  }
};

}  // namespace orc
