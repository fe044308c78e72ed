// PNG encoder for `image.save(opts.output)` (main.rs:176): 8-bit RGB, per-row filters, a zlib stream of one
// fixed-Huffman deflate block over a greedy LZ77 parse (no compression library needed on either machine).  The decoded
// pixels are what the reference's `image` crate would decode from its own file; the bytes of the file are not (its
// encoder makes other choices).
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace rth {

inline uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
  return ~crc;
}

inline void put_be32(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24));
  v.push_back((uint8_t)(x >> 16));
  v.push_back((uint8_t)(x >> 8));
  v.push_back((uint8_t)x);
}

inline void put_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
  put_be32(out, (uint32_t)data.size());
  const size_t at = out.size();
  out.insert(out.end(), type, type + 4);
  out.insert(out.end(), data.begin(), data.end());
  put_be32(out, crc32_update(0, out.data() + at, out.size() - at));
}

// LSB-first bit stream of RFC 1951; Huffman codes go in most-significant bit first (3.1.1)
struct BitWriter {
  std::vector<uint8_t>& out;
  uint64_t acc = 0;
  int n = 0;
  void put(uint32_t v, int bits) {
    acc |= (uint64_t)v << n;
    n += bits;
    while (n >= 8) {
      out.push_back((uint8_t)(acc & 0xFFu));
      acc >>= 8;
      n -= 8;
    }
  }
  void put_code(uint32_t code, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) r |= ((code >> i) & 1u) << (bits - 1 - i);
    put(r, bits);
  }
  void flush() {
    if (n > 0) out.push_back((uint8_t)(acc & 0xFFu));
    acc = 0;
    n = 0;
  }
};

// One deflate block with the FIXED Huffman codes (RFC 1951 3.2.6) over a greedy LZ77 parse (3-byte hash chains, 32 KiB
// window): a compressor small enough to carry, good for 2-3x on rendered images after PNG's row filters.
inline void deflate_fixed(const std::vector<uint8_t>& in, std::vector<uint8_t>& out) {
  static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  BitWriter bw{out};
  auto symbol = [&](uint32_t s) {  // literal / length alphabet
    if (s < 144) bw.put_code(0x30 + s, 8);
    else if (s < 256) bw.put_code(0x190 + (s - 144), 9);
    else if (s < 280) bw.put_code(s - 256, 7);
    else bw.put_code(0xC0 + (s - 280), 8);
  };
  bw.put(1, 1);  // BFINAL
  bw.put(1, 2);  // BTYPE = 01
  const size_t n = in.size();
  constexpr uint32_t HASH_BITS = 15, WINDOW = 32768, NIL = 0xFFFFFFFFu;
  std::vector<uint32_t> head(1u << HASH_BITS, NIL), prev(n, NIL);
  auto hash = [&](size_t i) { return ((uint32_t)in[i] * 0x9E3779B1u ^ (uint32_t)in[i + 1] * 0x85EBCA6Bu ^ (uint32_t)in[i + 2] * 0xC2B2AE35u) >> (32 - HASH_BITS); };
  auto insert = [&](size_t i) {
    if (i + 2 < n) {
      const uint32_t h = hash(i);
      prev[i] = head[h];
      head[h] = (uint32_t)i;
    }
  };
  size_t i = 0;
  while (i < n) {
    uint32_t best_len = 0, best_dist = 0;
    if (i + 2 < n) {
      uint32_t cand = head[hash(i)];
      const uint32_t max_len = (uint32_t)(n - i < 258 ? n - i : 258);
      for (int chain = 0; chain < 32 && cand != NIL && i - cand <= WINDOW; ++chain, cand = prev[cand]) {
        uint32_t l = 0;
        while (l < max_len && in[cand + l] == in[i + l]) ++l;
        if (l > best_len) {
          best_len = l;
          best_dist = (uint32_t)(i - cand);
          if (l == max_len) break;
        }
      }
    }
    if (best_len >= 3) {
      int lc = 28;
      while (len_base[lc] > best_len) --lc;
      symbol(257 + lc);
      bw.put(best_len - len_base[lc], len_extra[lc]);
      int dc = 29;
      while (dist_base[dc] > best_dist) --dc;
      bw.put_code((uint32_t)dc, 5);
      bw.put(best_dist - dist_base[dc], dist_extra[dc]);
      for (uint32_t k = 0; k < best_len; ++k) insert(i + k);
      i += best_len;
    } else {
      symbol(in[i]);
      insert(i);
      ++i;
    }
  }
  symbol(256);
  bw.flush();
}

// rgb8: [height][width][3].  Returns the encoded file.
inline std::vector<uint8_t> encode_png_rgb8(const uint8_t* rgb8, uint32_t width, uint32_t height) {
  // scanlines, each preceded by its filter type: the one of None / Sub / Up / Paeth with the smallest sum of absolute
  // (signed) residuals (the heuristic of the PNG specification, 12.8)
  const size_t stride = 3 * (size_t)width;
  std::vector<uint8_t> raw;
  raw.reserve((size_t)height * (stride + 1));
  std::vector<uint8_t> cand[4];
  for (auto& c : cand) c.resize(stride);
  const std::vector<uint8_t> zero_row(stride, 0);
  for (uint32_t y = 0; y < height; ++y) {
    const uint8_t* cur = rgb8 + (size_t)y * stride;
    const uint8_t* up = y ? rgb8 + (size_t)(y - 1) * stride : zero_row.data();
    for (size_t x = 0; x < stride; ++x) {
      const int a = x >= 3 ? cur[x - 3] : 0, b = up[x], c = x >= 3 ? up[x - 3] : 0;
      const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
      const int paeth = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
      cand[0][x] = cur[x];
      cand[1][x] = (uint8_t)(cur[x] - a);
      cand[2][x] = (uint8_t)(cur[x] - b);
      cand[3][x] = (uint8_t)(cur[x] - paeth);
    }
    static const uint8_t type_of[4] = {0, 1, 2, 4};
    int best = 0;
    uint64_t best_sum = ~0ull;
    for (int f = 0; f < 4; ++f) {
      uint64_t sum = 0;
      for (size_t x = 0; x < stride; ++x) sum += cand[f][x] < 128 ? cand[f][x] : 256 - cand[f][x];
      if (sum < best_sum) {
        best_sum = sum;
        best = f;
      }
    }
    raw.push_back(type_of[best]);
    raw.insert(raw.end(), cand[best].begin(), cand[best].end());
  }
  std::vector<uint8_t> z;  // zlib: CMF/FLG, one fixed-Huffman deflate block, Adler-32
  z.push_back(0x78);
  z.push_back(0x01);
  deflate_fixed(raw, z);
  if (z.size() > raw.size() + 5 * (raw.size() / 65535 + 1) + 2) {  // (noise: fixed codes spend 9 bits on half the literals)
    z.resize(2);
    size_t pos = 0;
    do {  // stored blocks of <= 65535 bytes
      const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
      z.push_back(pos + n == raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00
      z.push_back((uint8_t)(n & 0xFF));
      z.push_back((uint8_t)(n >> 8));
      z.push_back((uint8_t)(~n & 0xFF));
      z.push_back((uint8_t)((~n >> 8) & 0xFF));
      z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
      pos += n;
    } while (pos < raw.size());
  }
  uint32_t a = 1, b = 0;
  for (size_t i = 0; i < raw.size();) {
    const size_t stop = i + 5552 < raw.size() ? i + 5552 : raw.size();  // largest run before a, b can overflow
    for (; i < stop; ++i) {
      a += raw[i];
      b += a;
    }
    a %= 65521u;
    b %= 65521u;
  }
  put_be32(z, (b << 16) | a);

  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, width);
  put_be32(ihdr, height);
  ihdr.push_back(8);  // bit depth
  ihdr.push_back(2);  // colour type: truecolour
  ihdr.push_back(0);  // deflate
  ihdr.push_back(0);  // adaptive filtering (filter 0 on every line)
  ihdr.push_back(0);  // no interlace
  put_chunk(out, "IHDR", ihdr);
  put_chunk(out, "IDAT", z);
  put_chunk(out, "IEND", {});
  return out;
}

inline bool write_file(const std::string& path, const std::vector<uint8_t>& bytes) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
  return std::fclose(f) == 0 && ok;
}

}  // namespace rth
