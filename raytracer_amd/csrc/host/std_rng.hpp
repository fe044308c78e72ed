// Host layer: the generator behind `StdRng::seed_from_u64(5)` (scenes.rs:64), for scene 14 ("the reference's own
// random_scene geometry").  The crates are not vendored in the reference; this follows their published algorithms
// (Cargo.lock: rand 0.8.3, rand_core 0.6, rand_chacha 0.3):
//   StdRng = ChaCha with 12 rounds, key = 8 words of PCG32-XSH-RR seeded from the u64 (the state steps before every
//   output), 64-bit block counter from 0 in words 12..13, stream id 0 in words 14..15; a u64 is two consecutive
//   words, low first; f64 in [0,1) = top 53 bits / 2^53; gen_range(lo..hi) = (top 52 bits as a [1,2) mantissa - 1)
//   * (hi - lo) + lo, redrawn if that rounds up to hi.
// The block function is pinned by RFC 7539's vectors (rth_kat_chacha, tests/test_stdrng.py); seeding and float
// conversion have no vector available offline: parity unpinned.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>

namespace rth {

class ChaCha {
 public:
  using Words = std::array<uint32_t, 16>;
  // `rounds` column+diagonal passes in total (20 = RFC 7539, 12 = StdRng): out = permuted(in) + in
  static Words block(const Words& in, unsigned rounds) {
    static constexpr unsigned char LANES[8][4] = {{0, 4, 8, 12}, {1, 5, 9, 13}, {2, 6, 10, 14}, {3, 7, 11, 15},
                                                  {0, 5, 10, 15}, {1, 6, 11, 12}, {2, 7, 8, 13}, {3, 4, 9, 14}};
    Words x = in;
    for (unsigned pass = 0; pass < rounds / 2; ++pass)
      for (const auto& q : LANES) quarter(x[q[0]], x[q[1]], x[q[2]], x[q[3]]);
    for (size_t i = 0; i < 16; ++i) x[i] += in[i];
    return x;
  }
  static void quarter(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    static constexpr int ROT[4] = {16, 12, 8, 7};
    for (int step = 0; step < 4; ++step) {
      uint32_t& sum = (step & 1) ? c : a;
      uint32_t& addend = (step & 1) ? d : b;
      uint32_t& mixed = (step & 1) ? b : d;
      sum += addend;
      mixed ^= sum;
      mixed = (mixed << ROT[step]) | (mixed >> (32 - ROT[step]));
    }
  }
};

class StdRng {
 public:
  explicit StdRng(uint64_t seed) {
    static constexpr uint32_t SIGMA[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    std::memcpy(state_.data(), SIGMA, sizeof(SIGMA));
    uint64_t pcg = seed;
    for (size_t w = 4; w < 12; ++w) {
      pcg = pcg * 6364136223846793005ull + 11634580027462260723ull;
      const uint32_t x = (uint32_t)(((pcg >> 18) ^ pcg) >> 27), r = (uint32_t)(pcg >> 59);
      state_[w] = r ? (x >> r) | (x << (32u - r)) : x;
    }
    for (size_t w = 12; w < 16; ++w) state_[w] = 0;
  }
  uint64_t next_u64() {
    const uint64_t low = word();
    return low | ((uint64_t)word() << 32);
  }
  double next_f64() { return (double)(next_u64() >> 11) * 0x1p-53; }
  double range(double lo, double hi) {
    for (;;) {
      const uint64_t mantissa = next_u64() >> 12;
      const uint64_t image = mantissa | (1023ull << 52);  // exponent 0: a value in [1, 2)
      double one_two;
      std::memcpy(&one_two, &image, sizeof(one_two));
      const double v = (one_two - 1.0) * (hi - lo) + lo;
      if (v < hi) return v;
    }
  }

 private:
  uint32_t word() {
    if (at_ == 16) {
      out_ = ChaCha::block(state_, 12);
      if (++state_[12] == 0) ++state_[13];
      at_ = 0;
    }
    return out_[at_++];
  }
  ChaCha::Words state_{}, out_{};
  size_t at_ = 16;
};

}  // namespace rth
