// Minimal PNG encoder for `image.save(opts.output)` (main.rs:176): 8-bit RGB, filter 0, zlib stream of
// STORED deflate blocks (no compression library needed on either machine).  The decoded pixels are what
// the reference's `image` crate would decode from its own file; the bytes of the file are not (its
// encoder compresses).
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

// rgb8: [height][width][3].  Returns the encoded file.
inline std::vector<uint8_t> encode_png_rgb8(const uint8_t* rgb8, uint32_t width, uint32_t height) {
  std::vector<uint8_t> raw;  // scanlines, each preceded by filter type 0
  raw.reserve((size_t)height * (3 * (size_t)width + 1));
  for (uint32_t y = 0; y < height; ++y) {
    raw.push_back(0);
    raw.insert(raw.end(), rgb8 + (size_t)y * width * 3, rgb8 + (size_t)(y + 1) * width * 3);
  }
  std::vector<uint8_t> z;  // zlib: CMF/FLG, stored blocks of <= 65535 bytes, Adler-32
  z.push_back(0x78);
  z.push_back(0x01);
  size_t pos = 0;
  do {
    const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
    z.push_back(pos + n == raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00
    z.push_back((uint8_t)(n & 0xFF));
    z.push_back((uint8_t)(n >> 8));
    z.push_back((uint8_t)(~n & 0xFF));
    z.push_back((uint8_t)((~n >> 8) & 0xFF));
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    pos += n;
  } while (pos < raw.size());
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
